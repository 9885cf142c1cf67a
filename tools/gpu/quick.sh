#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed" | head -40 > gpurun_out/gpu_tests.log
timeout 300 python tools/bench_grid.py 2>&1 | grep -v amdgpu > gpurun_out/bench_grid.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1
tail -5 gpurun_out/gpu_tests.log; grep "binned\|naive" gpurun_out/bench_grid.log
python - <<'PY'
import json
for line in open('gpurun_out/bench.log'):
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'])
        for k,v in d['kernels'].items(): print('   ',k,v)
PY
