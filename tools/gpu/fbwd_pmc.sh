#!/bin/bash
# GPU-box: issue / stall counters of the fused field backward kernels (working-tree library and head), three rocprofv3 --pmc passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/fb
cd /tmp
for lib in ${FB_LIBS:-new head}; do
  if [ $lib = head ]; then export MORPHEUS_HIP_LIB=$REPO/morpheus_amd/_build/libmorpheus_head.so; else unset MORPHEUS_HIP_LIB; fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
    i=$((i+1)); rm -rf /tmp/fbpmc_${lib}_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/fbpmc_${lib}_$i -- python $REPO/tools/gpu/fbwd_ab.py --one /tmp/x.pt > /dev/null 2>&1
  done
  python - $lib <<'PY'
import csv, glob, collections, sys, re
lib = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/fbpmc_{lib}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        if "field_fused" in n:
            per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("==", lib, "(per launch, mean over launches; wave-cycle counters are summed over the 1024 waves)")
for k in sorted(per):
    c = {n: sum(v) / len(v) for n, v in per[k].items()}
    wc = c.get("SQ_WAVE_CYCLES", 1)
    line = f"{k:44s} wave_cycles {wc:.3e}"
    for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
        if n in c:
            line += f" {n[3:]} {c[n] / wc:.3f}"
    print(line)
    print(" " * 44, {n: f"{v:.3e}" for n, v in c.items() if n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_SALU")})
PY
done
