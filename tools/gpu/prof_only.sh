#!/bin/bash
# GPU-box script: rocprofv3 kernel trace of the default bench only -> gpurun_out/prof/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers ${1:+--workload $1} > "$REPO/gpurun_out/prof_bench.log" 2>&1
tail -1 "$REPO/gpurun_out/prof_bench.log" | cut -c1-200
