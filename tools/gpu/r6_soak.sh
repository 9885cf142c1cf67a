#!/bin/bash
# round 6: longer runs of the step workloads (finite losses, steady timings, no allocator growth): a soak, not a measurement
O=gpurun_out/r6soak; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-cpu-baseline --detail-out $O/$name.json > $O/$name.log 2>&1; python - <<PY
import json
d=json.load(open("$O/$name.json")); c=d["config"]
print("%-22s steps %4d  ms/step %8.3f  value %12.1f %s  loss %.6g  mean loss %.6g  alloc %s" % ("$name", d["steps"], d["ms_per_step"], d["value"], d["unit"], c["loss"], c.get("loss_mean_of_timed_steps") or float("nan"), c.get("allocator_in_timed_region")))
PY
}
run cfg3_200 --mode b3 --steps 200 --warmup 5 --no-kernel-timers
run train_loop_100 --workload train_loop --steps 100 --warmup 2 --no-kernel-timers
run train_loop_graph_100 --workload train_loop --graph --steps 100 --warmup 2
run train_virtual_72_200 --workload train_virtual --steps 200 --warmup 3 --no-kernel-timers
run train_real_400 --workload train_real --steps 400 --warmup 5 --no-kernel-timers
run train_real_graph_400 --workload train_real --graph --steps 400 --warmup 5
MORPHEUS_MAX_PARK_GB=64 run train_virtual_180_cap64_40 --workload train_virtual --virtual-res 180 --steps 40 --warmup 3 --no-kernel-timers
