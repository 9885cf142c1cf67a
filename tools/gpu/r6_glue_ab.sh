#!/bin/bash
# round 6: same-box A/B of the model's step cache (MORPHEUS_IMPLICIT_OPERANDS) on the reference-glue real-view step, alternating runs
O=gpurun_out/r6ab; mkdir -p $O
run() { MORPHEUS_IMPLICIT_OPERANDS=$1 timeout 300 python bench.py --workload train_real --glue $2 --steps 64 --no-cpu-baseline --no-kernel-timers --detail-out /tmp/x.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('implicit=$1', '$2', d['ms_per_step'])"; }
for i in 1 2 3; do run 1 reference; run 0 reference; done | tee $O/ab.txt
for i in 1 2; do run 1 fused; run 1 reference_scoped; done | tee -a $O/ab.txt
