#!/bin/bash
# GPU-box script (round 6): everything the committed profiles/r06_* summaries come from.  Outputs under gpurun_out/ (scratch);
# `python tools/collect_profiles.py r06` copies the judged summaries into profiles/.  R6_LIGHT=1 skips the f32 PMC passes, the
# censuses and the micro-benchmarks (a mid-round refresh).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
rm -f gpurun_out/precision_report.jsonl gpurun_out/bench_detail_*.json
( time timeout 1500 python -m pytest tests -q -m gpu --durations=12 ) > gpurun_out/gpu_tests_full.log 2>&1
grep -E "^(E  |FAILED|[0-9]+ (passed|failed))|Error|passed|failed|assert|^real|s call" gpurun_out/gpu_tests_full.log | head -60 > gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 600 python tests/parity_report.py > gpurun_out/parity.log 2>&1
timeout 600 python tests/parity_report.py --f64 > gpurun_out/parity_f64.jsonl 2> /dev/null
# host-sensitive workloads before the CPU-baseline leg of the default bench (it loads 16-64 host threads for ~20 s)
for wl in cfg2 cfg3b density128 train_real train_virtual; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --detail-out gpurun_out/bench_detail_$wl.json > gpurun_out/bench_$wl.log 2>&1
done
timeout 300 python bench.py --workload train_virtual --virtual-res 180 --steps 16 --no-cpu-baseline --detail-out gpurun_out/bench_detail_train_virtual_180.json > gpurun_out/bench_train_virtual_180.log 2>&1
( time timeout 1200 python bench.py --detail-out gpurun_out/bench_detail_cfg3.json ) > gpurun_out/bench.log 2>&1
timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-kernel-timers --detail-out gpurun_out/bench_detail_n2.json > gpurun_out/bench_n2.log 2>&1
timeout 300 python bench.py --workload train_real --graph --detail-out gpurun_out/bench_detail_train_real_graph.json > gpurun_out/bench_train_real_graph.log 2>&1
# the drop-in step: the step cache on / off inside one process (tools/gpu/glue_ab.py), per glue
( for g in reference fused; do timeout 200 python tools/gpu/glue_ab.py --glue $g --switch implicit --blocks 6 2>&1 | grep -v amdgpu | tail -8; done ) > gpurun_out/glue_ab.log 2>&1
cd /tmp
MODES="b3"; [ -z "$R6_LIGHT" ] && MODES="b3 f32"
for m in $MODES; do
  sfx=$([ $m = b3 ] && echo "" || echo "_$m")
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof$sfx" -- python "$REPO/bench.py" --mode $m --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-extras --detail-out /tmp/prof_detail.json > "$REPO/gpurun_out/prof_bench$sfx.log" 2>&1
  CMD="python $REPO/bench.py --mode $m --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras --detail-out /tmp/pmc_detail.json"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch$sfx -- $CMD > $REPO/gpurun_out/pmc_fetch$sfx.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_write$sfx -- $CMD > $REPO/gpurun_out/pmc_write$sfx.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_sq$sfx -- $CMD > $REPO/gpurun_out/pmc_sq$sfx.log 2>&1
done
# the parked form of dPre4 beside it (MORPHEUS_REGEN_DPRE4=0): the bytes the regeneration removes, measured
CMD="python $REPO/bench.py --mode b3 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras --detail-out /tmp/pmc_detail.json"
MORPHEUS_REGEN_DPRE4=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch_parked -- $CMD > $REPO/gpurun_out/pmc_fetch_parked.log 2>&1
MORPHEUS_REGEN_DPRE4=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/pmc_write_parked -- $CMD > $REPO/gpurun_out/pmc_write_parked.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $REPO/gpurun_out/pmc_lds -- $CMD > $REPO/gpurun_out/pmc_lds.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU --output-format csv -d $REPO/gpurun_out/pmc_lds2 -- $CMD > $REPO/gpurun_out/pmc_lds2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train_real" -- python "$REPO/bench.py" --workload train_real --steps 16 --warmup 3 --no-kernel-timers --detail-out /tmp/prof_detail.json > "$REPO/gpurun_out/prof_train_real.log" 2>&1
cd "$REPO"
if [ -z "$R6_LIGHT" ]; then
  timeout 200 python tools/gpu/launch_census.py --top 60 2>&1 | grep -v "amdgpu\|Anomaly\|detect_anomaly" > gpurun_out/census_fused.log
  timeout 200 python tools/gpu/launch_census.py --glue reference --top 60 2>&1 | grep -v "amdgpu\|Anomaly\|detect_anomaly" > gpurun_out/census_ref.log
  timeout 200 python tools/gpu/launch_census.py --cfg3 --top 60 2>&1 | grep -v "amdgpu\|Anomaly\|detect_anomaly" > gpurun_out/census_cfg3.log
  timeout 200 python tools/gpu/host_profile.py --top 25 2>&1 | grep -v amdgpu > gpurun_out/host_profile.log
  timeout 300 python tools/bench_grid.py 2>&1 | grep -v amdgpu > gpurun_out/bench_grid.log
fi
tail -14 gpurun_out/gpu_tests.log; tail -2 gpurun_out/smoke.log | cut -c1-300
python - <<'PY'
import json
for f in ["bench", "bench_cfg2", "bench_cfg3b", "bench_train_real", "bench_train_virtual", "bench_train_virtual_180", "bench_density128", "bench_n2", "bench_train_real_graph"]:
    try:
        lines = [l for l in open(f"gpurun_out/{f}.log").read().splitlines() if l.strip()]
        last = [l for l in lines if l.startswith("{")][-1]
        d = json.loads(last)
        print(f, len(last), "bytes;", d["value"], d["unit"], d["ms_per_step"], "ms", d.get("headline_mode"), (d.get("roofline") or {}).get("frac"),
              d["config"].get("backend"), d.get("modes_ms_per_step"), d.get("train_real_ms"), d.get("train_virtual_ms"), d.get("train_loop_iters_per_s"))
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.log").read()[-1200:])
PY
cat gpurun_out/glue_ab.log | grep "median of medians"
