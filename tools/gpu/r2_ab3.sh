#!/bin/bash
# GPU-box script: GPU tests, then A/B of one env switch ($1=VAR $2=value for the "old" arm) on the workloads in $WLS
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
if [ "$3" != "notest" ]; then
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "^(FAILED|[0-9]+ (passed|failed))|passed|failed" | head -30 > gpurun_out/gpu_tests.log
tail -6 gpurun_out/gpu_tests.log
fi
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], ' '.join(f"{k.replace('mh_','')}={v['ms_per_step']}" for k, v in list(d['kernels'].items())[:9]))
PY
}
for wl in ${WLS:-cfg3}; do
 for i in 1 2; do
  for mode in new old; do
    if [ $mode = old ]; then export $1=$2; else unset $1; fi
    timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_ab_${wl}_${mode}.log 2>&1
    show gpurun_out/bench_ab_${wl}_${mode}.log
  done
 done
done
