#!/bin/bash
# round 6: what the 180^2 virtual-view step allocates (peak inside the timed steps) against what the caching allocator reserves,
# per allocator configuration and parked-memory cap
O=gpurun_out/r6p; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --workload train_virtual --virtual-res 180 --steps 16 --no-cpu-baseline --no-kernel-timers --detail-out $O/$name.json > $O/$name.log 2>&1
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d.get("ms_per_step"), d["config"].get("allocator_in_timed_region"), flush=True)
PY
}
run trim50 X=1
run trim40 MORPHEUS_MAX_RESERVED_FRACTION=0.4
run notrim MORPHEUS_MAX_RESERVED_FRACTION=0
run cap64 MORPHEUS_MAX_PARK_GB=64
