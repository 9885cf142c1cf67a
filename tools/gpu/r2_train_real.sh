#!/bin/bash
# GPU-box script: GPU tests + train_real bench with kernel table + rocprof launch census of the same
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert" | head -40 > gpurun_out/gpu_tests.log
tail -12 gpurun_out/gpu_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1
timeout 300 python bench.py --workload cfg3b --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg3b.log 2>&1
timeout 300 python bench.py --workload train_real > gpurun_out/bench_train_real.log 2>&1
timeout 300 python bench.py --workload train_real --no-kernel-timers > gpurun_out/bench_train_real_nt.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train_real" -- python "$REPO/bench.py" --workload train_real --steps 16 --warmup 3 --no-kernel-timers > "$REPO/gpurun_out/prof_train_real.log" 2>&1
cd "$REPO"
python - <<'PY'
import json, glob, csv
for f in ["bench", "bench_cfg3b", "bench_train_real", "bench_train_real_nt"]:
    try:
        line = [l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1]
        d = json.loads(line)
        print(f, d["value"], d["ms_per_step"], d["config"].get("sample_points_per_step_per_gpu"))
        ks = d.get("kernels") or {}
        if ks: print("   sum of timed C-ABI calls ms/step:", round(sum(v["ms_per_step"] for v in ks.values()), 3))
        for k, v in list(ks.items())[:12]: print("    ", k, v["calls_per_step"], v["avg_ms"], v["ms_per_step"], v.get("tflops"))
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.log").read()[-1500:])
fs = sorted(glob.glob('gpurun_out/prof_train_real/**/*kernel_stats.csv', recursive=True), key=lambda p: -__import__('os').path.getmtime(p))
rows = list(csv.DictReader(open(fs[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('rocprof: total kernel ms', round(tot / 1e6, 1), 'launches', calls, 'per step (19 steps)', round(calls / 19), 'GPU-busy ms/step', round(tot / 1e6 / 19, 2))
for r in rows[:28]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['AverageNs'])/1e3:8.1f}us")
PY
