#!/bin/bash
# GPU-box script: the full GPU suite + the two-rank bench path on one device (gloo)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/precision_report.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 ) > gpurun_out/gpu_tests_full.log 2>&1
grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert|^real|s call" gpurun_out/gpu_tests_full.log | head -60 > gpurun_out/gpu_tests.log
tail -22 gpurun_out/gpu_tests.log
timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-kernel-timers > gpurun_out/bench_n2.log 2>&1
tail -c 1500 gpurun_out/bench_n2.log
