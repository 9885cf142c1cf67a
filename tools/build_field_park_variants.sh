#!/bin/bash
# container-side (round 6): variants of the b3 field FORWARD that park fewer rows, for a same-box timing A/B of "is the field forward
# bound by its parking stores?".  Edits are made on a TEMPORARY copy of mlp_b3.hip (never in the product source):
#   nopad   rows 80..95 of the sdf net's input block (zeros nobody needs) not written               -> results unchanged
#   nohash  ... and the hash-feature rows (40..71, 224..255) not written: the backward then reads stale rows -- WRONG gradients on
#           purpose; only the forward's time is of interest (what reading the features from the encoder's output instead would save)
# -> morpheus_amd/_build/libmorpheus_fpark_<name>.so (MORPHEUS_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
B=morpheus_amd/_build
OBJS=$(ls $B/*.o | grep -v "/mlp_b3.o\|ab_\|fpark_")
for name in nopad nohash; do
  src=morpheus_amd/csrc/_fpark_mlp_b3.hip
  python - "$name" "$src" <<'PY'
import sys
name, dst = sys.argv[1], sys.argv[2]
s = open("morpheus_amd/csrc/mlp_b3.hip").read()
a = "for (int k = 0; k < 48; k++) PARK_STORE(k < 40 ? bin0[k] : 0.f, &tile[(2 * k + h) * TILE + pt]);"
assert a in s
if name == "nopad":
    s = s.replace(a, "for (int k = 0; k < 40; k++) PARK_STORE(bin0[k], &tile[(2 * k + h) * TILE + pt]);")
else:
    s = s.replace(a, "for (int k = 0; k < 40; k++) if (k < 20 || k >= 36) PARK_STORE(bin0[k], &tile[(2 * k + h) * TILE + pt]);")
    b = "for (int k = 0; k < 32; k++) PARK_STORE(binc[k], &tile[(224 + 2 * k + h) * TILE + pt]);"
    assert b in s
    s = s.replace(b, "for (int k = 16; k < 32; k++) PARK_STORE(binc[k], &tile[(224 + 2 * k + h) * TILE + pt]);")
open(dst, "w").write(s)
PY
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -c $src -o $B/fpark_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmorpheus_fpark_$name.so $OBJS $B/fpark_$name.o
  rm -f $src $B/fpark_$name.o
  echo built $B/libmorpheus_fpark_$name.so
done
