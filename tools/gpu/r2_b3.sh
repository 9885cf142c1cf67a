#!/bin/bash
# GPU-box script: GPU tests with the bf16x3 warp kernels on, then cfg3 A/B (b3 vs f32) on the same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
if [ "$1" != "notest" ]; then
MORPHEUS_MLP=b3 timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert" | head -40 > gpurun_out/gpu_tests_b3.log
tail -8 gpurun_out/gpu_tests_b3.log
fi
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], ' '.join(f"{k.replace('mh_','')}={v['ms_per_step']}" for k, v in list(d['kernels'].items())[:8]))
PY
}
for wl in cfg3 ${B3_WORKLOADS}; do
 for i in 1 2; do
  for mode in b3 f32; do
    MORPHEUS_MLP=$mode timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_b3_${wl}_${mode}.log 2>&1
    show gpurun_out/bench_b3_${wl}_${mode}.log
  done
 done
done
