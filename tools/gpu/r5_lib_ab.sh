#!/bin/bash
# GPU-box: same-box A/B of every morpheus_amd/_build/ab_*.so (tools/build_grid_variants.sh) on cfg3 (b3), the real-view (tr) and
# the virtual-view (tv) training step; per-kernel HIP-event times of the grid kernels.  Interleaved, two repetitions.
# WLS="b3" restricts the workloads.  The printed lines also go to gpurun_out/r5ab/summary.txt.
O=gpurun_out/r5ab; mkdir -p $O; : > $O/summary.txt
WLS=${WLS:-"b3 tr tv"}
for rep in 1 2; do
for so in morpheus_amd/_build/ab_*.so; do
  n=$(basename $so .so); export MORPHEUS_HIP_LIB=$PWD/$so
  rm -f $O/${n}_*_$rep.json
  [[ " $WLS " == *" b3 "* ]] && timeout 300 python bench.py --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${n}_b3_$rep.json > $O/${n}_b3_$rep.log 2>&1
  [[ " $WLS " == *" tr "* ]] && timeout 300 python bench.py --workload train_real --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${n}_tr_$rep.json > $O/${n}_tr_$rep.log 2>&1
  [[ " $WLS " == *" tv "* ]] && timeout 300 python bench.py --workload train_virtual --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${n}_tv_$rep.json > $O/${n}_tv_$rep.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for w in "$WLS".split():
    try:
        d=json.load(open("$O/${n}_%s_$rep.json" % w))
        print("$n", w, "ms/step", d["ms_per_step"], {k.replace("mh_grid_",""):round(v["ms_per_step"],3) for k,v in d["kernels"].items() if "grid" in k or "bin" in k})
    except Exception as e:
        print("$n", w, "FAILED", e)
PY
done
done
