cd "$GRAFT_REPO_ROOT"; REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_ta_step -- python $REPO/bench.py --mode b3 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --detail-out /tmp/pmc_detail.json > $REPO/gpurun_out/pmc_ta_step.log 2>&1
cd $REPO
python - <<'PY' | tee gpurun_out/ta_counters_step.txt
import csv, glob, collections, re
f = sorted(glob.glob("gpurun_out/pmc_ta_step/*/*_counter_collection.csv"))[-1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v) if v else 0.0
rows = []
for k, c in per.items():
    gui = mean(c["GRBM_GUI_ACTIVE"]) / 8.0
    if gui * len(c["GRBM_GUI_ACTIVE"]) < 3e5 or k.startswith(("at::", "Cijk", "__amd")):
        continue
    rows.append((gui * len(c["GRBM_GUI_ACTIVE"]), f"{k:44s} launches {len(c['GRBM_GUI_ACTIVE']):3d}  cycles/launch {gui:10.0f}  TA busy avr {mean(c['TA_BUSY_avr']) / gui:5.2f}  max {mean(c['TA_BUSY_max']) / gui:5.2f}"))
print("texture-address unit busy per kernel of the cfg3 step (b3; TA_BUSY_avr, _max over GRBM_GUI_ACTIVE / 8)")
for _, l in sorted(rows, reverse=True):
    print(l)
PY
