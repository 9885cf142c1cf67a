#!/bin/bash
# the 180 x 180 virtual-view step under different bounds on parked memory (MORPHEUS_MAX_PARK_GB)
O=gpurun_out/r5park; mkdir -p $O
for cap in 0 64 32; do
  MORPHEUS_MAX_PARK_GB=$cap timeout 600 python bench.py --workload train_virtual --virtual-res 180 --no-kernel-timers --no-cpu-baseline --detail-out $O/tv180_cap$cap.json > $O/tv180_cap$cap.log 2>&1
  python - <<PY
import json
d=json.load(open("$O/tv180_cap$cap.json"))
c=d["config"]
print("cap $cap: ms/step", d["ms_per_step"], "samples", c["sample_points_per_step_per_gpu"], c["allocator_in_timed_region"], {k:v for k,v in c["parked_memory_bound"].items() if k!="note"})
PY
done
