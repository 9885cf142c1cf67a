#!/bin/bash
# GPU-box: same-box A/B of every morpheus_amd/_build/ab_*.so (tools/build_grid_variants.sh) on cfg3 and the real-view training step;
# per-kernel HIP-event times of the grid kernels.  Interleaved, two repetitions.
O=gpurun_out/r5ab; mkdir -p $O
for rep in 1 2; do
for so in morpheus_amd/_build/ab_*.so; do
  n=$(basename $so .so); export MORPHEUS_HIP_LIB=$PWD/$so
  timeout 300 python bench.py --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${n}_b3_$rep.json > $O/${n}_b3_$rep.log 2>&1
  timeout 300 python bench.py --workload train_real --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${n}_tr_$rep.json > $O/${n}_tr_$rep.log 2>&1
  timeout 300 python bench.py --workload train_virtual --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${n}_tv_$rep.json > $O/${n}_tv_$rep.log 2>&1
  python - <<PY
import json
for w in ("b3", "tr", "tv"):
    try:
        d=json.load(open("$O/${n}_%s_$rep.json" % w))
        print("$n", w, "ms/step", d["ms_per_step"], {k.replace("mh_grid_",""):round(v["ms_per_step"],3) for k,v in d["kernels"].items() if "grid" in k or "bin" in k})
    except Exception as e:
        print("$n", w, "FAILED", e)
PY
done
done
