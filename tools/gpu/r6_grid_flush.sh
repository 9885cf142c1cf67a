#!/bin/bash
# GPU-box (round 6): what the brick backward's flush is made of -- the shipped kernel, the flush loop without its global atomics, no flush
# at all (wrong-result variants: tools/micro/hashgrid_brk_exp.patch via tools/build_grid_variants.sh).  cfg3 (two 2.1 M-point calls) and
# the real-view training step (calls of 0.14 - 0.83 M points: small work items, where the flush weighs most); 2 repetitions, one box
O=gpurun_out/r6flush; mkdir -p $O; : > $O/summary.txt
for rep in 1 2; do
for so in morpheus_amd/_build/ab_*.so; do
  n=$(basename $so .so); export MORPHEUS_HIP_LIB=$PWD/$so
  timeout 300 python bench.py --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${n}_b3_$rep.json > $O/${n}_b3_$rep.log 2>&1
  timeout 300 python bench.py --workload train_real --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${n}_tr_$rep.json > $O/${n}_tr_$rep.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for w in ("b3", "tr"):
    try:
        d = json.load(open("$O/${n}_%s_$rep.json" % w))
        print("$n", w, "ms/step", d["ms_per_step"], {k.replace("mh_grid_", ""): round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if "grid" in k})
    except Exception as e:
        print("$n", w, "FAILED", e)
PY
done
done
