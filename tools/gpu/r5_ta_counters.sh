#!/bin/bash
# GPU-box: texture-address-unit counters of the hash-grid kernels (tools/bench_grid.py runs the gathering and the brick-staged
# forms of both directions on 2.1 M points): one rocprofv3 --pmc pass with --kernel-trace only; summary -> gpurun_out/ta_counters.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_ta -- python $REPO/tools/bench_grid.py > $REPO/gpurun_out/pmc_ta.log 2>&1
cd $REPO
python - <<'PY' | tee gpurun_out/ta_counters.txt
import csv, glob, collections, re
f = sorted(glob.glob("gpurun_out/pmc_ta/*/*_counter_collection.csv"))[-1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v) if v else 0.0
print("texture-address unit per kernel (tools/bench_grid.py: 2 097 152 points, ray-ordered and uniform-random sets pooled; rocprofv3 --pmc, one pass)")
print("TA busy = TA_BUSY_avr (mean over the TA instances of busy cycles) / (GRBM_GUI_ACTIVE / 8 XCDs); wavefront-loads = TA_FLAT_READ_WAVEFRONTS_sum")
for k in sorted(per):
    if "grid" not in k:
        continue
    c = per[k]; gui = mean(c["GRBM_GUI_ACTIVE"]) / 8.0
    print(f"{k:42s} launches {len(c['GRBM_GUI_ACTIVE']):3d}  cycles {gui:10.0f}  TA busy avr {mean(c['TA_BUSY_avr']) / gui:5.2f}  max {mean(c['TA_BUSY_max']) / gui:5.2f}  wavefront-loads {mean(c['TA_FLAT_READ_WAVEFRONTS_sum']):12.0f}")
PY
