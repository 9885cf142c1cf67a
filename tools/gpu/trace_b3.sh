#!/bin/bash
# trace build of the library (s_memtime stamps in warp_fwd_b3_kernel) + the phase tables; never the product library
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
python tools/phase_trace_b3_pair.py 2>&1 | tail -22 | tee gpurun_out/phase_trace_b3_pair.txt
MH_TRACE_NOPARK=1 python tools/phase_trace_b3_pair.py 2>&1 | tail -22 | tee gpurun_out/phase_trace_b3_pair_nopark.txt
