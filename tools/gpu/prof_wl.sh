#!/bin/bash
# GPU-box script: rocprofv3 kernel trace of one bench workload ($1, default cfg3b) -> gpurun_out/prof_$1/
WL=${1:-cfg3b}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python bench.py --steps 10 --warmup 3 --workload $WL --no-cpu-baseline > gpurun_out/bench_$WL.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_$WL" -- python "$REPO/bench.py" --steps 5 --warmup 2 --workload $WL --no-cpu-baseline --no-kernel-timers > "$REPO/gpurun_out/prof_bench_$WL.log" 2>&1
cd "$REPO"
tail -1 gpurun_out/bench_$WL.log | cut -c1-400
f=$(find gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -d, -f1-4 | cut -c1-150
