#!/bin/bash
# GPU-box script: full GPU suite + train_real bench (kernel table) -- quick regression check
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/precision_report.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu --durations=5 ) > gpurun_out/gpu_tests_full.log 2>&1
grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert|^real|s call" gpurun_out/gpu_tests_full.log | head -60 > gpurun_out/gpu_tests.log
tail -14 gpurun_out/gpu_tests.log
timeout 300 python bench.py --workload train_real > gpurun_out/bench_train_real.log 2>&1
python - <<'PY'
import json
for f in ["bench_train_real"]:
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], "kernel sum", d.get("kernel_sum_ms_per_step"), d["config"].get("loss_mean_of_timed_steps"))
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.log").read()[-1500:])
PY
