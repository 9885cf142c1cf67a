#!/bin/bash
# GPU-box script: GPU suite + a list of bench invocations given as arguments ("name|flags" each); outputs under gpurun_out/r4/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
O=gpurun_out/r4
if [ "$1" != "--no-tests" ]; then
  ( time timeout 1500 python -m pytest tests -q -m gpu --durations=5 ) > $O/gpu_tests_full.log 2>&1
  grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert|^real" $O/gpu_tests_full.log | head -60 > $O/gpu_tests.log
  tail -25 $O/gpu_tests.log
else
  shift
fi
for spec in "$@"; do
  name="${spec%%|*}"; flags="${spec#*|}"
  ( time timeout 600 python bench.py $flags --no-cpu-baseline ) > $O/bench_$name.log 2> $O/bench_$name.err
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/r4/bench_{name}.log") if l.startswith("{")][-1])
    print(name, d["value"], d["unit"], d["ms_per_step"], "ms; kernel sum", d.get("kernel_sum_ms_per_step"), "launches", d.get("timed_launches_per_step"),
          "samples", d["config"].get("sample_points_per_step_per_gpu"), d["config"].get("hip_graph", {}).get("graphs_captured"))
    for k, v in list(d.get("kernels", {}).items())[:14]:
        print("   ", k, v)
except Exception as e:
    print(name, "FAILED", e); print(open(f"gpurun_out/r4/bench_{name}.err").read()[-1500:])
PY
  grep real $O/bench_$name.err | tail -1
done
