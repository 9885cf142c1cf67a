#!/bin/bash
# GPU-box script: is the slow 180 x 180 virtual-view step the allocator configuration or the first large process on a fresh box?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
run() {  # $1 = label, $2 = allocator conf ("" = default)
  if [ -n "$2" ]; then export PYTORCH_HIP_ALLOC_CONF="$2"; else export PYTORCH_HIP_ALLOC_CONF="backend:native"; fi
  python bench.py --workload train_virtual --virtual-res 180 --steps 12 --warmup 4 --no-kernel-timers --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); a=d['config']['allocator_in_timed_region']
print('$1', d['ms_per_step'], 'ms', a['conf'], 'reserved', a['reserved_GB'], 'peak', a['peak_allocated_GB'], 'dev_alloc', a['num_device_alloc'])"
}
run E1 expandable_segments:True
run E2 expandable_segments:True
run D1 ""
run D2 ""
run E3 expandable_segments:True
