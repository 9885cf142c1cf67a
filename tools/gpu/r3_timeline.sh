#!/bin/bash
# GPU-box script: kernel trace of the replayed real-view step, then the per-phase timeline of one step (tools/step_timeline_real.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_train_real_graph" -- python "$REPO/bench.py" --workload train_real --graph --steps 8 --warmup 2 --no-kernel-timers > "$REPO/gpurun_out/prof_train_real_graph.log" 2>&1
cd "$REPO"
python tools/step_timeline_real.py gpurun_out/prof_train_real_graph > gpurun_out/timeline_train_real_graph.txt 2>&1
python tools/step_aggregate.py gpurun_out/prof_train_real_graph > gpurun_out/aggregate_train_real_graph.txt 2>&1
tail -150 gpurun_out/timeline_train_real_graph.txt
