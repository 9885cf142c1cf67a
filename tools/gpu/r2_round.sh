#!/bin/bash
# GPU-box script (round 2): everything the committed profiles/ summaries come from.  Outputs under gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "^(E  |FAILED|PASSED|[0-9]+ (passed|failed))|Error|passed|failed" | head -60 > gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 600 python tests/parity_report.py > gpurun_out/parity.log 2>&1
# the CPU-baseline leg of the default bench loads 16-64 host threads for ~20 s and the host stays slow for a while after it
# (a 5 ms step like cfg2 then measures 8 ms): run the host-sensitive workloads first
for wl in cfg2 train_real cfg3b density128; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_$wl.log 2>&1
done
MORPHEUS_MLP=f32 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_cfg3_f32.log 2>&1
MORPHEUS_MLP=b3 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_cfg3_b3.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1
timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-kernel-timers > gpurun_out/bench_n2.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers > "$REPO/gpurun_out/prof_bench.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train_real" -- python "$REPO/bench.py" --workload train_real --steps 16 --warmup 3 --no-kernel-timers > "$REPO/gpurun_out/prof_train_real.log" 2>&1
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_sq -- $CMD > $REPO/gpurun_out/pmc_sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch -- $CMD > $REPO/gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_write -- $CMD > $REPO/gpurun_out/pmc_write.log 2>&1
cd "$REPO"
( echo "--- product (2 workgroups per CU, parking), 3 launches"; python tools/phase_trace.py; echo "--- the same after 300 back-to-back launches (sustained clock)"; MH_TRACE_ITERS=300 python tools/phase_trace.py; echo "--- no parking (inference form)"; MH_TRACE_NOPARK=1 python tools/phase_trace.py ) 2>&1 | grep -v amdgpu > gpurun_out/phase_trace.log
( echo "--- bf16x3 forward (MORPHEUS_MLP=b3), parking"; python tools/phase_trace_b3.py; echo "--- no parking"; MH_TRACE_NOPARK=1 python tools/phase_trace_b3.py ) 2>&1 | grep -v amdgpu > gpurun_out/phase_trace_b3.log
( echo "--- fp16x2 forward (the default), parking"; python tools/phase_trace_h2.py; echo "--- no parking"; MH_TRACE_NOPARK=1 python tools/phase_trace_h2.py ) 2>&1 | grep -v amdgpu > gpurun_out/phase_trace_h2.log
timeout 100 python tools/gpu/hbm_rates.py 2>&1 | grep -v amdgpu > gpurun_out/hbm_rates.log
( tools/micro/mfma_power; tools/micro/mfma_power2 ) > gpurun_out/mfma_power.log 2>&1
timeout 300 python tools/bench_grid.py 2>&1 | grep -v amdgpu > gpurun_out/bench_grid.log
tail -3 gpurun_out/gpu_tests.log; tail -2 gpurun_out/smoke.log | cut -c1-200
python - <<'PY'
import json
for f in ["bench", "bench_cfg3_f32", "bench_cfg3_b3", "bench_cfg2", "bench_cfg3b", "bench_train_real", "bench_density128", "bench_n2"]:
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1])
        print(f, d["value"], d["unit"], d["ms_per_step"], "ms", (d.get("roofline") or {}).get("frac"), d["config"].get("backend"))
    except Exception as e:
        print(f, "FAILED", e)
PY
grep "clock\|kernel ms" gpurun_out/phase_trace.log
