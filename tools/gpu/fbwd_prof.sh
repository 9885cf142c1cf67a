#!/bin/bash
# GPU-box: A/B of the fused field backward (head library vs working tree) + per-kernel times of both from rocprofv3 --kernel-trace --stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/fb
python tools/gpu/fbwd_ab.py 2>&1 | tail -30
cd /tmp
for lib in head new; do
  if [ $lib = head ]; then export MORPHEUS_HIP_LIB=$REPO/morpheus_amd/_build/libmorpheus_head.so; else unset MORPHEUS_HIP_LIB; fi
  rm -rf /tmp/fbprof_$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fbprof_$lib -- python $REPO/tools/gpu/fbwd_ab.py --one /tmp/x_$lib.pt > /dev/null 2>&1
  f=$(ls /tmp/fbprof_$lib/*/*_kernel_stats.csv 2>/dev/null | tail -1)
  echo "== $lib"; [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "field_fused" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"]) / 1e6:7.3f} ms')
PY
done
