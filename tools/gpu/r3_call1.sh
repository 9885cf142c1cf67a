#!/bin/bash
# GPU-box script (round 3, first contact): GPU tests + smoke + the three-mode bench line + train_real (kernel table, rocprof
# launch census with the kernel trace for tools/step_timeline.py, host profile) + rocprof stats of the headline mode.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
rm -f gpurun_out/precision_report.jsonl
( time timeout 1200 python -m pytest tests -q -m gpu -x --durations=15 ) > gpurun_out/gpu_tests_full.log 2>&1
grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert|^real|s call" gpurun_out/gpu_tests_full.log | head -60 > gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/bench.log 2>&1
timeout 300 python bench.py --workload train_real > gpurun_out/bench_train_real.log 2>&1
timeout 200 python tools/gpu/prof_cpu.py train_real > gpurun_out/prof_cpu_train_real.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train_real" -- python "$REPO/bench.py" --workload train_real --steps 16 --warmup 3 --no-kernel-timers > "$REPO/gpurun_out/prof_train_real.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -- python "$REPO/bench.py" --mode b3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers > "$REPO/gpurun_out/prof_bench.log" 2>&1
cd "$REPO"
tail -25 gpurun_out/gpu_tests.log; tail -2 gpurun_out/smoke.log | cut -c1-300
python - <<'PY'
import json, glob, csv, os
for f in ["bench", "bench_train_real"]:
    try:
        line = [l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1]
        d = json.loads(line)
        print(f, d["value"], d["ms_per_step"], d.get("headline_mode"), d["dtype"][:40], "kernel sum", d.get("kernel_sum_ms_per_step"))
        for m, r in (d.get("modes") or {}).items():
            print("   mode", m, r["value"], r["ms_per_step"], "loss", r["loss"], "roof", (r["roofline"] or {}).get("kernel"), (r["roofline"] or {}).get("frac"))
        if d.get("mode_errors"): print("   mode errors", d["mode_errors"])
        r = d.get("roofline") or {}
        print("   roofline", {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "frac", "traffic", "algorithmic_bytes", "parked_bytes")})
        for k, v in list((d.get("kernels") or {}).items())[:14]: print("    ", k, v["calls_per_step"], v["avg_ms"], v["ms_per_step"])
        print("   cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.log").read()[-2500:])
try:
    fs = sorted(glob.glob('gpurun_out/prof_train_real/**/*kernel_stats.csv', recursive=True), key=lambda p: -os.path.getmtime(p))
    rows = list(csv.DictReader(open(fs[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
    print('rocprof train_real: total kernel ms', round(tot / 1e6, 1), 'launches', calls, 'per step (19 steps)', round(calls / 19), 'GPU-busy ms/step', round(tot / 1e6 / 19, 2))
except Exception as e:
    print("prof_train_real FAILED", e)
PY
grep real gpurun_out/bench.log gpurun_out/gpu_tests_full.log | head; head -3 gpurun_out/prof_cpu_train_real.log
python tools/step_timeline_real.py gpurun_out/prof_train_real > gpurun_out/timeline_train_real.txt 2>&1; tail -3 gpurun_out/timeline_train_real.txt
