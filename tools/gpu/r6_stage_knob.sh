#!/bin/bash
# GPU-box (round 6, after the flat flush): the brick-staging threshold (MORPHEUS_GRID_STAGE_MIN_POINTS) on the training-step workloads, same box
O=gpurun_out/r6k; mkdir -p $O
for rep in 1 2; do
for thr in 1048576 524288 262144 65536; do
  export MORPHEUS_GRID_STAGE_MIN_POINTS=$thr
  for wl in train_real train_virtual; do
    timeout 300 python bench.py --workload $wl --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${wl}_$thr_$rep.json > $O/${wl}_${thr}_$rep.log 2>&1
    python - <<PY
import json
d=json.load(open("$O/${wl}_$thr_$rep.json"))
k=d["kernels"]
print("$wl", $thr, "ms/step", d["ms_per_step"], "kernel sum", round(sum(v["ms_per_step"] for v in k.values()),3), {n.replace("mh_grid_",""):round(v["ms_per_step"],3) for n,v in k.items() if "grid" in n})
PY
  done
done
done
