#!/bin/bash
# container-side: build the experiment's library variant, morpheus_amd/_build/ab_fbwd.so = the product library with the generated,
# software-pipelined fused field backward (bf16 x 3 form) in place of field_fused_*_kernel<.., true>.
#   tools/experiments/field_bwd_pipeline/build_variant.sh [--fill 6] [-DFB_ASM_MEM=1]
# Run the A/B on the box with tools/gpu/fbwd_ab.py (MORPHEUS_HIP_LIB=.../ab_fbwd.so for the "new" side).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$HERE/../../.."
FILL=6; EXTRA=""
while [ $# -gt 0 ]; do case "$1" in --fill) FILL=$2; shift 2;; *) EXTRA="$EXTRA $1"; shift;; esac; done
T=$(mktemp -d)
cp "$ROOT"/morpheus_amd/csrc/* $T/
sed -i "s#../../include/morpheus_hip.h#$ROOT/include/morpheus_hip.h#" $T/common.h
cp "$HERE/field_bwd_b3.h" $T/
python "$HERE/gen_field_bwd.py" --fill $FILL > $T/field_bwd_b3_gen.h
(cd $T && patch -p1 -s < "$HERE/mlp_launch.patch")
python -c "import sys; sys.path.insert(0, '$ROOT'); from morpheus_amd import build; build.build()" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $EXTRA -c $T/mlp.hip -o $T/mlp.o
OBJS=$(ls "$ROOT"/morpheus_amd/_build/*.o | grep -v "/mlp.o\|ab_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT"/morpheus_amd/_build/ab_fbwd.so $OBJS $T/mlp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $EXTRA --cuda-device-only -S -o "$ROOT"/morpheus_amd/_build/ab_fbwd.s $T/mlp.hip 2>/dev/null
echo "built morpheus_amd/_build/ab_fbwd.so (+ ab_fbwd.s); spilled registers per generated kernel:"
awk '/\.name:.*field_fused.*b3/{n=$2} n&&/vgpr_spill_count/{print "  " n, $2; n=""}' "$ROOT"/morpheus_amd/_build/ab_fbwd.s
rm -rf $T
