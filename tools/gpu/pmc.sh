#!/bin/bash
# GPU-box script: PMC counter passes (own runs, --kernel-trace only, as the guide prescribes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers"
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_sq -- $CMD > $REPO/gpurun_out/pmc_sq.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch -- $CMD > $REPO/gpurun_out/pmc_fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_write -- $CMD > $REPO/gpurun_out/pmc_write.log 2>&1
cd $REPO; find gpurun_out/pmc_* -name "*.csv" | head; tail -2 gpurun_out/pmc_sq.log | cut -c1-300
