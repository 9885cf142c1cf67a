#!/bin/bash
# GPU-box script: tests, smoke, parity report, bench, rocprof kernel trace.  Outputs under gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "^(E  |FAILED|PASSED|tests/|[0-9]+ (passed|failed))|Error|passed|failed" | head -120 > gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 600 python tests/parity_report.py > gpurun_out/parity.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --workload cfg2 --no-cpu-baseline > gpurun_out/bench_cfg2.log 2>&1
REPO=$(pwd)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers > "$REPO/gpurun_out/prof_bench.log" 2>&1
cd "$REPO"
find gpurun_out/prof -name "*stats*" | head -5
tail -4 gpurun_out/gpu_tests.log; tail -2 gpurun_out/smoke.log; tail -1 gpurun_out/bench.log | cut -c1-600
