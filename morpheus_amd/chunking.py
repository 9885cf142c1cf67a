"""A bound on parked memory: per-sample field / warp queries in row chunks whose activations are re-made in backward.

The MLP kernels park what their backward needs in HBM (warp nets: 10.9 KB per sample; every field point query, incl. the six
finite-difference taps of a normal: ~2 KB).  A whole-view training step (180 x 180 rays of a novel view, finite-difference
normals, the perturbed-normal regulariser) parks ~45 KB per sample -- 104 GB at its peak on a 288 GB part, and nothing bounded it.
`chunked_query` splits such a query by rows when the call's estimate exceeds the cap (MORPHEUS_MAX_PARK_GB; unset: 0.4 of the
device's memory, so that a step which fits comfortably -- the 180 x 180 view: ~106 GB by this estimate -- pays nothing): every
chunk but the last runs through torch.utils.checkpoint (forward without autograd state, i.e. nothing parked; re-run with parking
when backward reaches it, one chunk at a time), the last chunk runs as usual -- backward reaches it first, so its parked tiles are
gone before the first re-run.  The outputs are the concatenation; values and gradients are those of the unchunked call
(tests/test_gpu_render.py::test_parked_memory_cap_chunks_the_queries, backward outside and INSIDE an open operand_scope).
Reentrant checkpointing means: a chunked call's gradients are reached by `loss.backward()` only -- `torch.autograd.grad(...)` and
`backward(inputs=...)` raise torch's own "Checkpointing is not compatible with .grad()" error; and a
chunk's re-run prepares weight operands of its own (model.fresh_operand_scope), never the enclosing step's.
Cost: one extra forward of the re-run rows, paid only by calls over the cap.  Measured on the 180 x 180 virtual-view step (2.2 M
samples; profiles/r05_park_cap_180.txt): no bound 47.8 ms with 243 GB reserved by the caching allocator; cap 64 GB: 58.0 ms, 64 GB
reserved; cap 32 GB: 61.3 ms, 38 GB reserved."""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import torch
import torch.utils.checkpoint

WARP_PARK_BYTES = (64 + 2 * 640 + 40 + 2 * 672) * 4      # csrc/mlp_dev.h: WARP_ACT_ROWS + WARP_DPRE_ROWS floats per sample
FIELD_PARK_BYTES = (96 + 64 * 5 + 8) * 4 + 300           # FIELD_ACT_ROWS floats + hash features / per-point temporaries
STATS = dict(calls=0, chunked_calls=0, chunks=0, rerun_rows=0)


DEFAULT_FRACTION = 0.4       # of the device's memory, when MORPHEUS_MAX_PARK_GB is not set (115 GB on a 288 GB MI355X)
_total = {}


def park_cap_bytes(device=None) -> float:
    """MORPHEUS_MAX_PARK_GB in GB (<= 0: no bound); unset: DEFAULT_FRACTION of the device's memory.  Read at every call."""
    env = os.environ.get("MORPHEUS_MAX_PARK_GB")
    if env is not None and env != "":
        gb = float(env)
        return float("inf") if gb <= 0 else gb * 1e9
    if not torch.cuda.is_available():
        return float("inf")
    idx = torch.cuda.current_device() if device is None or getattr(device, "index", None) is None else device.index
    if idx not in _total:
        _total[idx] = float(torch.cuda.get_device_properties(idx).total_memory)
    return DEFAULT_FRACTION * _total[idx]


# What the caching allocator keeps RESERVED is a second quantity, and not this module's to bound (round 6, profiles/r06_park_alloc.txt).
# The 180 x 180 virtual-view step allocates 104-108 GB at its peak; after a dozen steps the allocator holds 245-250 GB of the
# device's 288 -- the packed sample count changes from view to view, a cached block can be split but never grown, and every step
# opens new segments beside the chopped ones.  Measured: size classes for the large buffers, the allocator's own
# roundup_power2_divisions and garbage_collection_threshold change nothing; handing the unused blocks back before a step
# (empty_cache) costs 0.4 s per driver allocation on this platform (480-720 ms per step instead of 48) -- the sporadic 350-600 ms
# runs of the capped configuration were exactly that; `torch.cuda.memory.set_per_process_memory_fraction(0.6)` in the CALLER bounds
# it for nothing: 112 GB reserved, 48.5 ms (its frees happen once, in the first steps).  INTEGRATION.md section 5 recommends that
# line for whole-view steps; bench.py's train_virtual workload sets it.  What this module does about the rest of the device: the
# default cap also looks at what is actually free when the call is made (`available_bytes`).
BIG_CALL = 8e9


def available_bytes(device=None) -> float:
    """device memory a call could still get: free on the device + cached by this process's allocator and not in use"""
    idx = torch.cuda.current_device() if device is None or getattr(device, "index", None) is None else device.index
    free, _ = torch.cuda.mem_get_info(idx)
    return float(free) + float(torch.cuda.memory_reserved(idx) - torch.cuda.memory_allocated(idx))


def query_bytes_per_row(warp: bool, field_points: int) -> int:
    """estimate of what one row of a query parks: the warp nets (if evaluated) + `field_points` field point queries"""
    return (WARP_PARK_BYTES if warp else 0) + FIELD_PARK_BYTES * int(field_points)


def rows_under_cap(bytes_per_row_all_queries: int, cap: Optional[float] = None, device=None, rows: Optional[int] = None) -> int:
    """rows per chunk under the cap.  rows: the call's row count -- a call whose estimate is large (>= 8 GB) is also held to 0.85 of
    the memory that is free when it is made (another tenant of the device -- the reference's guidance UNet, a second process --
    is not in DEFAULT_FRACTION); an explicit MORPHEUS_MAX_PARK_GB is taken as given."""
    explicit = cap is not None or bool(os.environ.get("MORPHEUS_MAX_PARK_GB"))
    cap = park_cap_bytes(device) if cap is None else cap
    if not explicit and rows is not None and torch.cuda.is_available() and float(bytes_per_row_all_queries) * rows >= BIG_CALL \
            and not torch.cuda.is_current_stream_capturing():
        cap = min(cap, 0.85 * available_bytes(device))
    if cap == float("inf"):
        return 1 << 62
    return max(int(cap // max(bytes_per_row_all_queries, 1)) // 8192 * 8192, 8192)


def chunked_query(fn: Callable, sliced: Sequence[Optional[torch.Tensor]], rows: int, model=None):
    """fn(*sliced) -> tuple of per-row tensors (or None entries); `sliced`: tensors with the same leading length (or None),
    cut by rows.  rows >= the length, or autograd off: one plain call.  model: the scene_representation fn queries -- a chunk's
    re-run in backward gets an operand scope of its own (see model.fresh_operand_scope for why it must)."""
    M = next(t for t in sliced if t is not None).shape[0]
    STATS["calls"] += 1
    if rows >= M or not torch.is_grad_enabled():
        return fn(*sliced)
    bounds = list(range(0, M, rows))
    STATS["chunked_calls"] += 1
    STATS["chunks"] += len(bounds)
    # reentrant checkpointing recomputes only if some INPUT requires a gradient (the parameters are reached through the module,
    # not through the argument list): a one-element carrier that does
    carrier = torch.ones(1, device=next(t for t in sliced if t is not None).device, requires_grad=True)

    def run(carrier_, *args):
        if model is not None and torch.is_grad_enabled():      # the re-run inside backward (the first pass runs under no_grad)
            with model.fresh_operand_scope():
                return fn(*args)
        return fn(*args)

    outs = []
    for i, a in enumerate(bounds):
        b = min(a + rows, M)
        args = [None if t is None else t[a:b] for t in sliced]
        if i == len(bounds) - 1:
            out = fn(*args)
        else:
            STATS["rerun_rows"] += b - a
            out = torch.utils.checkpoint.checkpoint(run, carrier, *args, use_reentrant=True, preserve_rng_state=False)
        outs.append(tuple(out))
    return tuple(None if outs[0][j] is None else torch.cat([o[j] for o in outs], 0) for j in range(len(outs[0])))
