"""Harness code that DRIVES the hot path the way the reference's caller does (bench.py workloads, GPU tests, fixture generation):
the reference's real-view / virtual-view training steps restated around `HotPathRenderer.render_rays`, the reference-glue form of
the caller's loss code, the stand-in for the SDS guidance, HIP-graph capture of a whole step, the occupancy warm-up.  None of it
is part of the drop-in: a maintainer's three edits import `morpheus_amd` only, and morpheus.py stays the caller."""
