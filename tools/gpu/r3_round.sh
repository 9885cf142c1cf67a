#!/bin/bash
# GPU-box script (round 3): everything the committed profiles/r03_* summaries come from.  Outputs under gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
rm -f gpurun_out/precision_report.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu --durations=12 ) > gpurun_out/gpu_tests_full.log 2>&1
grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert|^real|s call" gpurun_out/gpu_tests_full.log | head -60 > gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 600 python tests/parity_report.py > gpurun_out/parity.log 2>&1
# host-sensitive workloads before the CPU-baseline leg of the default bench (it loads 16-64 host threads for ~20 s)
for wl in cfg2 train_real cfg3b density128; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_$wl.log 2>&1
done
( time timeout 900 python bench.py ) > gpurun_out/bench.log 2>&1
timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-kernel-timers > gpurun_out/bench_n2.log 2>&1
timeout 300 python bench.py --workload train_real --graph > gpurun_out/bench_train_real_graph.log 2>&1
cd /tmp
for m in b3 h2 f32; do
  sfx=$([ $m = b3 ] && echo "" || echo "_$m")
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof$sfx" -- python "$REPO/bench.py" --mode $m --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers > "$REPO/gpurun_out/prof_bench$sfx.log" 2>&1
  CMD="python $REPO/bench.py --mode $m --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch$sfx -- $CMD > $REPO/gpurun_out/pmc_fetch$sfx.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_write$sfx -- $CMD > $REPO/gpurun_out/pmc_write$sfx.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_sq$sfx -- $CMD > $REPO/gpurun_out/pmc_sq$sfx.log 2>&1
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train_real" -- python "$REPO/bench.py" --workload train_real --steps 16 --warmup 3 --no-kernel-timers > "$REPO/gpurun_out/prof_train_real.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_train_real_graph" -- python "$REPO/bench.py" --workload train_real --graph --steps 8 --warmup 2 --no-kernel-timers > "$REPO/gpurun_out/prof_train_real_graph.log" 2>&1
cd "$REPO"
python tools/step_timeline_real.py gpurun_out/prof_train_real_graph > gpurun_out/timeline_train_real_graph.txt 2>&1
timeout 200 python tools/gpu/graph_memset_probe.py 2>&1 | grep -v "amdgpu\|Warn\|warn" > gpurun_out/graph_memset_probe.txt
tail -14 gpurun_out/gpu_tests.log; tail -2 gpurun_out/smoke.log | cut -c1-300
python - <<'PY'
import json
for f in ["bench", "bench_cfg2", "bench_cfg3b", "bench_train_real", "bench_density128", "bench_n2"]:
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1])
        print(f, d["value"], d["unit"], d["ms_per_step"], "ms", d.get("headline_mode"), (d.get("roofline") or {}).get("frac"), d["config"].get("backend"),
              {m: r["ms_per_step"] for m, r in (d.get("modes") or {}).items()})
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.log").read()[-1200:])
PY
grep -c . gpurun_out/parity.log; ls gpurun_out/pmc_fetch*/runc/ 2>/dev/null | head
