#!/bin/bash
# GPU-box (round 6): same-box A/B of the field forward's parking stores (tools/build_field_park_variants.sh): cfg3 and the 180^2
# virtual-view step, per-kernel HIP-event times; head = the in-tree library
O=gpurun_out/r6fpark; mkdir -p $O; : > $O/summary.txt
for rep in 1 2; do
for v in ${FPARK_RUN:-head nopad nohash}; do
  unset MORPHEUS_HIP_LIB
  [ $v != head ] && export MORPHEUS_HIP_LIB=$PWD/morpheus_amd/_build/libmorpheus_fpark_$v.so
  timeout 300 python bench.py --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${v}_b3_$rep.json > $O/${v}_b3_$rep.log 2>&1
  timeout 300 python bench.py --workload train_virtual --virtual-res 180 --steps 8 --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${v}_tv_$rep.json > $O/${v}_tv_$rep.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for w in ("b3", "tv"):
    try:
        d = json.load(open("$O/${v}_%s_$rep.json" % w))
        print("$v", w, "ms/step", d["ms_per_step"], {k.replace("mh_", ""): round(x["ms_per_step"], 3) for k, x in d["kernels"].items() if "field" in k})
    except Exception as e:
        print("$v", w, "FAILED", e)
PY
done
done
