#!/bin/bash
# GPU-box script (round 2, call 2): new parity tests, MFMA power/clock micro-benchmark, parity report at both floors
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert" | head -60 > gpurun_out/gpu_tests.log
tail -25 gpurun_out/gpu_tests.log
timeout 120 tools/micro/mfma_power | tee gpurun_out/mfma_power.log
timeout 600 python tests/parity_report.py > gpurun_out/parity.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/parity.log'):
    if not l.startswith('{'): continue
    d = json.loads(l)
    if 'detail' in d:
        print(d['case'])
        for k, v in d['detail'].items():
            print('   ', k, 'floor1e-3: %.2e' % v['rel_floor_1e3'], 'relaxed(%g): %.2e' % (v['relaxed_floor'], v['rel_floor_relaxed']),
                  'max_abs %.2e' % v['max_abs'], 'worst_ref %.3e' % v['worst_ref'], 'over', v['n_over_1e4_at_1e3'], '/', v['n'])
PY
