#!/bin/bash
# GPU-box script: fixed-capacity sampling + graphed real-view step (tests, then bench eager vs graphed), cfg3 with --graph
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_render.py -q -m gpu -k "fixed_capacity or graphed or real_view_step or marcher or regularisers" 2>&1 | grep -v "Warning\|warn\|Consider\|run_backward" | tail -70 > gpurun_out/graph_tests.log
tail -30 gpurun_out/graph_tests.log

timeout 300 python bench.py --workload train_real --no-kernel-timers > gpurun_out/bench_train_real_eager.log 2>&1
timeout 300 python bench.py --workload train_real --graph > gpurun_out/bench_train_real_graph.log 2>&1


python - <<'PY'
import json
for f in ["bench_train_real_eager", "bench_train_real_graph"]:
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("hip_graph"), d["config"]["sample_points_per_step_per_gpu"], d["config"]["loss"], d["config"].get("loss_mean_of_timed_steps"))
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.log").read()[-1800:])
PY
