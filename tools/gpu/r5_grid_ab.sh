#!/bin/bash
# round 5 hash-grid A/B: the ops + render parity tests, then cfg3 (b3) and the real-view training step with per-kernel HIP-event times
O=gpurun_out/r5g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_render.py -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
for rep in 1 2; do
  timeout 300 python bench.py --mode b3 --no-cpu-baseline --no-extras --detail-out $O/bench_b3_$rep.json > $O/bench_b3_$rep.log 2>&1
  timeout 300 python bench.py --workload train_real --mode b3 --no-cpu-baseline --no-extras --detail-out $O/bench_tr_$rep.json > $O/bench_tr_$rep.log 2>&1
  python - <<PY
import json
for n in ("b3", "tr"):
    d=json.load(open("$O/bench_%s_$rep.json" % n))
    print(n, "ms/step", d["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items() if "grid" in k or "bin" in k})
PY
done
