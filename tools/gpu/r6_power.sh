#!/bin/bash
# GPU-box: power / clock telemetry of the product kernels (tools/gpu/power_clock.py) + one rocprofv3 --pmc pass for the effective
# clock GRBM_GUI_ACTIVE / duration per kernel.  -> gpurun_out/r6/power_clock.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/r6
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bf16_bare tools/micro/mfma_bf16_bare.hip 2>/dev/null
python tools/gpu/power_clock.py --seconds 2.0 --micro /tmp/mfma_bf16_bare > gpurun_out/r6/power_clock.txt 2> gpurun_out/r6/power_clock.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/r6/pmc_gui -- python $REPO/tools/gpu/power_clock.py --once > $REPO/gpurun_out/r6/pmc_gui.log 2>&1
cd $REPO
python - <<'PY' >> gpurun_out/r6/power_clock.txt
import csv, glob, collections, re
cc = sorted(glob.glob("gpurun_out/r6/pmc_gui/*/*_counter_collection.csv"))
kt = sorted(glob.glob("gpurun_out/r6/pmc_gui/*/*_kernel_trace.csv"))
if cc and kt:
    dur = {}
    for r in csv.DictReader(open(kt[-1])):
        dur[r["Dispatch_Id"]] = (re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[-1])):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            name, ns = dur[r["Dispatch_Id"]]
            per[name].append((float(r["Counter_Value"]), ns))
    print("\n# effective clock per kernel = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration; rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace over `power_clock.py --once`")
    print("# (three launches each, the LAST one reported: a cold first launch clocks differently; serialised by the profiler, so durations are longer than in the loop above)")
    for k in sorted(per, key=lambda k: -per[k][-1][1]):
        c, ns = per[k][-1]
        if ns < 50e3:
            continue
        print(f"{k:50s} {ns / 1e6:8.3f} ms   GUI_ACTIVE/8 {c / 8:12.0f} cycles   effective clock {c / 8 / ns * 1e3:7.0f} MHz")
PY
tail -40 gpurun_out/r6/power_clock.txt
