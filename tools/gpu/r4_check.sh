#!/bin/bash
# GPU-box script (round 4, first contact): full GPU suite, float64 parity table, the default bench line (cfg3 x 3 modes + graph replay
# + train_real x 3 + train_virtual x 2 + CPU baseline).  Outputs under gpurun_out/r4/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
O=gpurun_out/r4
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 ) > $O/gpu_tests_full.log 2>&1
grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert|^real|s call" $O/gpu_tests_full.log | head -80 > $O/gpu_tests.log
tail -30 $O/gpu_tests.log
timeout 600 python tests/parity_report.py --f64 > $O/parity_f64.jsonl 2> $O/parity_f64.err
python - <<'PY'
import json
for l in open("gpurun_out/r4/parity_f64.jsonl"):
    r = json.loads(l)
    print(r["mode"], r["case"][11:], " | ".join(f"{k}: hip {v['hip_vs_f64_max']:.1e}/{v['hip_vs_f64_n_over_1e4']} ref {v['ref32_vs_f64_max']:.1e}/{v['ref32_vs_f64_n_over_1e4']} h-r {v['hip_vs_ref32_max']:.1e}/{v['hip_vs_ref32_n_over_1e4']}" for k, v in r.items() if isinstance(v, dict)))
PY
tail -3 $O/parity_f64.err
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r4/bench.log") if l.startswith("{")][-1])
    print("cfg3", d["value"], d["ms_per_step"], d["headline_mode"], "frac", d["roofline"]["frac"], {m: r["ms_per_step"] for m, r in d["modes"].items()})
    print("per_kernel", {k: (v["ms_per_step"], v["frac"]) for k, v in d["roofline"]["per_kernel"].items()})
    for k in ("train_real", "train_virtual"):
        for kk, v in d[k].items():
            if isinstance(v, dict):
                print(k, kk, v.get("value"), v.get("ms_per_step"), v.get("sample_points_per_step"), v.get("error", "")[-300:], v.get("shadings_of_timed_steps"))
    print("cpu", d["cpu_baseline"])
except Exception as e:
    print("bench FAILED", e); print(open("gpurun_out/r4/bench.err").read()[-1500:]); print(open("gpurun_out/r4/bench.log").read()[-1500:])
PY
tail -4 $O/bench.err
