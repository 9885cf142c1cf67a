#!/bin/bash
# GPU-box script: longer runs to catch drift / leaks / flakiness
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for i in 1 2; do timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -1; done
timeout 900 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-kernel-timers 2>&1 | tail -1 | cut -c1-170
timeout 900 python - <<'PY'
import torch, json, sys
sys.argv=['bench.py','--steps','20','--warmup','3','--no-cpu-baseline','--no-kernel-timers']
import runpy
m0=torch.cuda.memory_allocated()
runpy.run_path('bench.py', run_name='__main__')
print('peak GB', torch.cuda.max_memory_allocated()/2**30, 'reserved GB', torch.cuda.memory_reserved()/2**30)
PY
