#!/bin/bash
# GPU-box script: hash-grid operator tests + micro-benchmark only
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "grid" 2>&1 | tail -2
timeout 300 python tools/bench_grid.py 2>&1 | grep -v amdgpu | tee gpurun_out/bench_grid.log | grep "binned\|fwd"
