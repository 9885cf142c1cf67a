#!/bin/bash
# GPU-box script (round 2, first call): full GPU tests, the new N=2 self-launch, first train_real / density128 numbers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed" | head -40 > gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1
timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-kernel-timers > gpurun_out/bench_n2.log 2>&1
timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-kernel-timers --no-overlap > gpurun_out/bench_n2_noovl.log 2>&1
timeout 300 python bench.py --workload train_real > gpurun_out/bench_train_real.log 2>&1
timeout 300 python bench.py --workload train_real --no-kernel-timers > gpurun_out/bench_train_real_nt.log 2>&1
MORPHEUS_MARCH=two_pass timeout 300 python bench.py --workload train_real > gpurun_out/bench_train_real_2pass.log 2>&1
timeout 300 python bench.py --workload density128 --steps 5 --warmup 2 > gpurun_out/bench_density128.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train_real" -- python "$REPO/bench.py" --workload train_real --steps 16 --warmup 3 --no-kernel-timers > "$REPO/gpurun_out/prof_train_real.log" 2>&1
cd "$REPO"
python - <<'PY'
import json, glob
for f in ["bench", "bench_n2", "bench_n2_noovl", "bench_train_real", "bench_train_real_nt", "bench_train_real_2pass", "bench_density128"]:
    try:
        line = [l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1]
        d = json.loads(line)
        print(f, d["value"], d["ms_per_step"], d["config"].get("sample_points_per_step_per_gpu"), d["config"].get("backend"))
        ks = d.get("kernels") or {}
        tot = sum(v["ms_per_step"] for v in ks.values())
        if ks: print("   sum of timed C-ABI calls ms/step:", round(tot, 3))
        for k, v in list(ks.items())[:14]: print("    ", k, v["calls_per_step"], v["avg_ms"], v["ms_per_step"], v.get("tflops"))
    except Exception as e:
        print(f, "FAILED", e)
        print(open(f"gpurun_out/{f}.log").read()[-1500:])
PY
f=$(find gpurun_out/prof_train_real -name "*kernel_stats.csv" | head -1)
echo "--- rocprof train_real: $f"; head -40 "$f" | cut -d, -f1-5 | cut -c1-160
