#!/bin/bash
# GPU-box script: GPU tests, headline bench with kernel table (x2), hash-grid micro-benchmark
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert" | head -40 > gpurun_out/gpu_tests.log
tail -6 gpurun_out/gpu_tests.log
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench.log 2>&1
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], ' '.join(f"{k.replace('mh_','')}={v['ms_per_step']}" for k, v in list(d['kernels'].items())[:9]))
PY
done
timeout 300 python tools/bench_grid.py 2>&1 | grep -v amdgpu | tee gpurun_out/bench_grid.log | grep "fwd\|binned"
