#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_render.py -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 3 --workload cfg3b --no-cpu-baseline > gpurun_out/bench_cfg3b.log 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/bench_cfg3b.log'):
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'])
        for k,v in list(d['kernels'].items())[:10]: print('   ',k,v['calls_per_step'],v['ms_per_step'],v.get('tflops'))
PY
