#!/bin/bash
# same-box A/B of the scheduling-group fill counts: library variants built in the container under morpheus_amd/_build/ab/
# (-DWG_FILL=k: weight gradients; -DB3_Q_FILL=k -DB3_Q4_FILL=k: warp forward / backward-data quarters that carry an epilogue):
#   for v in "wg0:-DWG_FILL=0" "wg3:-DWG_FILL=3" "wg5:-DWG_FILL=5" "q4:-DB3_Q_FILL=4 -DB3_Q4_FILL=4" "q5:-DB3_Q_FILL=5 -DB3_Q4_FILL=5" "q8:-DB3_Q_FILL=8 -DB3_Q4_FILL=8"; do
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared -Iinclude ${v#*:} -o morpheus_amd/_build/ab/lib_${v%%:*}.so morpheus_amd/csrc/*.hip; done
O=gpurun_out/r5ab; mkdir -p $O
for name in head wg0 wg3 wg5 q4 q5 q8 head; do
  lib=morpheus_amd/_build/ab/lib_$name.so; [ $name = head ] && lib=morpheus_amd/_build/libmorpheus_hip.so
  MORPHEUS_HIP_LIB=$PWD/$lib timeout 300 python bench.py --mode b3 --no-cpu-baseline --detail-out $O/$name.json > $O/$name.log 2>&1
  python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("%-5s ms/step %.3f " % ("$name", d["ms_per_step"]), {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith(("mh_mlp_wgrad","mh_warp"))})
PY
done
