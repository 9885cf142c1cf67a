#!/bin/bash
# GPU-box script: rocprofv3 kernel stats of one bench workload; env vars pass through.  usage: prof_one.sh <workload> <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$(pwd); export TMPDIR=/tmp
rm -rf "$REPO/gpurun_out/prof_$2"; mkdir -p "$REPO/gpurun_out"
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_$2" -- python "$REPO/bench.py" --workload $1 --steps 10 --warmup 3 --no-kernel-timers --no-cpu-baseline > "$REPO/gpurun_out/prof_$2.log" 2>&1
tail -1 "$REPO/gpurun_out/prof_$2.log" | cut -c1-150
