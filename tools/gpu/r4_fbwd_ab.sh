#!/bin/bash
# GPU-box script: A/B of the fused field backward's arithmetic (MORPHEUS_FIELD_BWD = f32 | sdf | b3; b3 = the default) on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/fbwd; export TMPDIR=/tmp
O=gpurun_out/fbwd
timeout 900 python -m pytest tests -q -m gpu -x > $O/tests_b3.log 2>&1
tail -5 $O/tests_b3.log
for rep in 1 2; do
for fb in f32 sdf b3; do
  for spec in "cfg3|--workload cfg3 --mode b3" "tr|--workload train_real --mode b3" "v72|--workload train_virtual --virtual-res 72 --mode b3"; do
    name="${spec%%|*}"; flags="${spec#*|}"
    MORPHEUS_FIELD_BWD=$fb timeout 600 python bench.py $flags --no-cpu-baseline > $O/${name}_${fb}_$rep.log 2> $O/${name}_${fb}_$rep.err
    python - "$O/${name}_${fb}_$rep.log" "$name $fb" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d.get("kernels", {})
    print(sys.argv[2], d["ms_per_step"], "ms; field_bwd", k.get("mh_field_bwd_fused", {}).get("ms_per_step"), "field_fwd", k.get("mh_field_fwd", {}).get("ms_per_step"), "loss", d["config"].get("loss_mean_of_timed_steps"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
done
