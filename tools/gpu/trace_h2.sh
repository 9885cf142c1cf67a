#!/bin/bash
# phase trace of warp_fwd_h2_kernel: builds the stamped library on the box, then traces both workgroup shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared -DMH_PHASE_TRACE -o morpheus_amd/_build/libmorpheus_trace.so morpheus_amd/csrc/*.hip || exit 1
for w in default; do
  python tools/phase_trace_h2.py 2>&1 | tail -12
done
MH_TRACE_NOPARK=1 python tools/phase_trace_h2.py 2>&1 | tail -12
