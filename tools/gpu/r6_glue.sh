#!/bin/bash
# round 6: the drop-in step's glue -- parity tests that cover it, the launch censuses, the train_real / train_loop numbers
O=gpurun_out/r6g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 200 python tools/gpu/launch_census.py --top 200 2>&1 | grep -v "amdgpu\|Anomaly\|detect_anomaly" > $O/census_fused.log
timeout 200 python tools/gpu/launch_census.py --glue reference --top 200 2>&1 | grep -v "amdgpu\|Anomaly\|detect_anomaly" > $O/census_ref.log
head -1 $O/census_fused.log $O/census_ref.log
for g in fused reference reference_scoped; do
  timeout 300 python bench.py --workload train_real --glue $g --no-cpu-baseline --no-kernel-timers --detail-out $O/detail_$g.json 2> $O/bench_$g.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$g', d['ms_per_step'], d['value'])"
done
