#!/bin/bash
# GPU-box script: A/B of two builds of the library on one box: the in-tree build against morpheus_amd/_build/libmorpheus_head.so
# (another source state compiled in the container).  Arguments: "name|bench flags" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/libab
for rep in 1 2; do
for lib in head new; do
  if [ $lib = head ]; then export MORPHEUS_HIP_LIB=$PWD/morpheus_amd/_build/libmorpheus_head.so; else unset MORPHEUS_HIP_LIB; fi
  for spec in "$@"; do
    name="${spec%%|*}"; flags="${spec#*|}"
    timeout 600 python bench.py $flags --no-cpu-baseline > gpurun_out/libab/${name}_${lib}_$rep.log 2> gpurun_out/libab/${name}_${lib}_$rep.err
    python - "gpurun_out/libab/${name}_${lib}_$rep.log" "$name $lib" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d.get("kernels", {})
    print(sys.argv[2], d["ms_per_step"], "ms;", {n.replace("mh_", ""): round(v["ms_per_step"], 3) for n, v in list(k.items())[:7]})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
done
