#!/bin/bash
# GPU-box script: A/B of the fused field backward (default) against the split form (MORPHEUS_FIELD_BWD=split), same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert" | head -40 > gpurun_out/gpu_tests.log
tail -12 gpurun_out/gpu_tests.log
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], ' '.join(f"{k.replace('mh_','')}={v['ms_per_step']}" for k, v in list(d['kernels'].items())[:10]))
PY
}
for wl in cfg3 cfg3b train_real; do
  for mode in fused split; do
    if [ $mode = split ]; then export MORPHEUS_FIELD_BWD=split; else unset MORPHEUS_FIELD_BWD; fi
    timeout 300 python bench.py --workload $wl --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_${wl}_${mode}.log 2>&1
    show gpurun_out/bench_${wl}_${mode}.log
  done
done
