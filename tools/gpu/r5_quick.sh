#!/bin/bash
# round 5 quick A/B: the ops parity tests + the cfg3 step in the b3 mode (per-kernel HIP-event times) + the two-wave phase trace
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q > $O/gpu_ops.txt 2>&1; tail -3 $O/gpu_ops.txt
for rep in 1 2; do
  timeout 300 python bench.py --mode b3 --no-cpu-baseline --detail-out $O/bench_b3_$rep.json > $O/bench_b3_$rep.log 2>&1
  python - <<PY
import json
d=json.load(open("$O/bench_b3_$rep.json"))
print("ms/step", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if v["ms_per_step"]>0.3})
PY
done
if [ -f morpheus_amd/_build/libmorpheus_trace.so ]; then bash tools/gpu/trace_b3.sh 2>&1 | grep -v amdgpu.ids; fi
