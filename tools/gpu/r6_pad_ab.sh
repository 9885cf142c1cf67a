#!/bin/bash
# round 6 same-box A/B: pad rows of H0 / dPre5 written and read (library built with -DMH_PARK_PAD_ROWS +
# MORPHEUS_WGRAD_LIVE=0 = the round-5 traffic) against the shipped form; cfg3, b3, per-kernel HIP-event times
O=gpurun_out/r6pad; mkdir -p $O
for rep in 1 2 3; do
for v in ${VARIANTS:-pad new}; do
  unset MORPHEUS_HIP_LIB MORPHEUS_WGRAD_LIVE
  if [ $v = pad ]; then export MORPHEUS_HIP_LIB=$PWD/morpheus_amd/_build/libmorpheus_padrows.so MORPHEUS_WGRAD_LIVE=0; fi
  if [ $v = h0pad ]; then export MORPHEUS_HIP_LIB=$PWD/morpheus_amd/_build/libmorpheus_h0pad.so; fi     # only the forward's H0 pad rows written (nobody reads them)
  timeout 300 python bench.py --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${v}_$rep.json > $O/${v}_$rep.log 2>&1
  python - "$O/${v}_$rep.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernels"]
print(sys.argv[2], d["ms_per_step"], {n.replace("mh_", ""): round(v["ms_per_step"], 3) for n, v in k.items() if v["ms_per_step"] > 0.3}, flush=True)
PY
done
done 2>&1 | tee $O/ab.txt
