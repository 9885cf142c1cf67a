O=gpurun_out/r6ww; mkdir -p $O
for rep in 1 2; do
for v in ${WW_RUN:-head warpwrap}; do
  unset MORPHEUS_HIP_LIB
  [ $v != head ] && export MORPHEUS_HIP_LIB=$PWD/morpheus_amd/_build/libmorpheus_fpark_$v.so
  timeout 300 python bench.py --mode b3 --no-cpu-baseline --no-extras --detail-out $O/${v}_$rep.json > $O/${v}_$rep.log 2>&1
  python - <<PY
import json
d = json.load(open("$O/${v}_$rep.json"))
print("$v", d["ms_per_step"], {k.replace("mh_", ""): round(x["ms_per_step"], 3) for k, x in d["kernels"].items() if x["ms_per_step"] > 0.5})
PY
done
done
