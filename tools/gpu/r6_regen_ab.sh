#!/bin/bash
# round 6 same-box A/B: dPre4 of the warp nets regenerated in the layer-4 weight-gradient launches
# (MORPHEUS_REGEN_DPRE4=1, shipped) against parked by backward-data and read back (=0); cfg3, b3, per-kernel HIP-event times
O=gpurun_out/r6regen; mkdir -p $O
for rep in 1 2 3; do
for v in 0 1; do
  MORPHEUS_REGEN_DPRE4=$v timeout 300 python bench.py --mode b3 --no-cpu-baseline --no-extras --detail-out $O/regen${v}_$rep.json > $O/regen${v}_$rep.log 2>&1
  python - "$O/regen${v}_$rep.json" "regen=$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernels"]
print(sys.argv[2], d["ms_per_step"], {n.replace("mh_", ""): round(v["ms_per_step"], 3) for n, v in k.items() if v["ms_per_step"] > 0.3}, "loss", d["config"].get("loss"), flush=True)
PY
done
done 2>&1 | tee $O/ab.txt
