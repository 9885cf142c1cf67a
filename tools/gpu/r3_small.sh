#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "warp" 2>&1 | tail -5
timeout 300 python bench.py --workload train_real > gpurun_out/bench_train_real.log 2>&1
timeout 300 python bench.py --mode b3 --no-cpu-baseline --steps 10 > gpurun_out/bench_b3.log 2>&1
python - <<'PY'
import json
for f in ["bench_train_real", "bench_b3"]:
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], "kernel sum", d.get("kernel_sum_ms_per_step"))
        for k, v in list((d.get("kernels") or {}).items())[:9]: print("    ", k, v["calls_per_step"], v["avg_ms"], v["ms_per_step"])
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.log").read()[-1500:])
PY
