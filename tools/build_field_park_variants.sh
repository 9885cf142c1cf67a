#!/bin/bash
# container-side (round 6): variants of the b3 field FORWARD that park fewer rows, for a same-box timing A/B of "is the field forward
# bound by its parking stores?".  Edits are made on a TEMPORARY copy of mlp_b3.hip (never in the product source):
#   nopad   rows 80..95 of the sdf net's input block (zeros nobody needs) not written               -> results unchanged
#   nohash  ... and the hash-feature rows (40..71, 224..255) not written: the backward then reads stale rows -- WRONG gradients on
#           purpose; only the forward's time is of interest (what reading the features from the encoder's output instead would save)
#   wrap / samerow   the field forward's stores into 256 tile slots (an L2-resident window) / onto two rows per tile: bytes or store path?
#   warpwrap / warpx4   the same question for the WARP forward: 512 tile slots; dwordx4 stores in accumulator order
#   (FPARK_VARIANTS="name ..." selects; every one of these except nopad computes WRONG results on purpose -- side libraries only)
# -> morpheus_amd/_build/libmorpheus_fpark_<name>.so (MORPHEUS_HIP_LIB); records: profiles/r06_ab_field_park_rows.txt, r06_ab_warp_fwd_parking.txt
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
B=morpheus_amd/_build
OBJS=$(ls $B/*.o | grep -v "/mlp_b3.o\|ab_\|fpark_")
for name in ${FPARK_VARIANTS:-nopad nohash samerow}; do
  src=morpheus_amd/csrc/_fpark_mlp_b3.hip
  python - "$name" "$src" <<'PY'
import sys
name, dst = sys.argv[1], sys.argv[2]
s = open("morpheus_amd/csrc/mlp_b3.hip").read()
a = "for (int k = 0; k < 40; k++) PARK_STORE(bin0[k], &tile[(2 * k + h) * TILE + pt]);"      # (the shipped form: no padding rows)
assert a in s
if name == "samerow":
    # WRONG results on purpose: every parking store of the field forward goes to rows 0 / 1 of its tile -- the same store INSTRUCTIONS,
    # ~1 % of the bytes reaching HBM: is the forward bound by the bytes it parks or by issuing the stores?
    import re
    i0 = s.index("__device__ __forceinline__ void fb3_epilogue(")
    i1 = s.index("// ---- fp32 fragments (b3 order")
    seg = s[i0:i1]
    seg = seg.replace("&ht[(32 * t + acc_row(r, h)) * TILE + pt]", "&ht[h * TILE + pt]")
    seg = seg.replace("&tile[(2 * k + h) * TILE + pt]", "&tile[h * TILE + pt]").replace("&tile[(224 + 2 * k + h) * TILE + pt]", "&tile[h * TILE + pt]")
    s = s[:i0] + seg + s[i1:]
elif name == "wrap":
    # WRONG results on purpose: the forward parks every tile into one of 256 tile slots (14 MB: stays in L2 / the memory-side cache), i.e.
    # the same store instructions on distinct lines and ~no HBM writes: HBM bytes or the store path?
    a2 = "float *tile = acts ? acts + tile_id * (int64_t)(FIELD_ACT_ROWS * TILE) : nullptr;"
    i0 = s.index("void field_fwd_b3_kernel(")
    assert a2 in s[i0:]
    s = s[:i0] + s[i0:].replace(a2, "float *tile = acts ? acts + (tile_id & 255) * (int64_t)(FIELD_ACT_ROWS * TILE) : nullptr;", 1)
elif name == "warpwrap":
    # WRONG results on purpose: the WARP forward parks every tile into one of 2048 tile slots (2048 x 176 KB = 360 MB > the 256 MB
    # memory-side cache: "warpwrap512" = 512 slots = 90 MB stays on chip): its store instructions on distinct lines without HBM writes
    a2 = "float *tile = PARK ? acts + tile_id * (int64_t)(WARP_ACT_ROWS * TILE) : nullptr;"
    assert a2 in s
    s = s.replace(a2, "float *tile = PARK ? acts + (tile_id & 511) * (int64_t)(WARP_ACT_ROWS * TILE) : nullptr;")
elif name == "warpx4":
    # WRONG results on purpose: the WARP forward parks a lane's 16 accumulator values of an output tile as four dwordx4 stores in
    # accumulator order ([tile t][lane][16] -- not the feature-major rows the weight-gradient kernels read): the same bytes, a quarter
    # of the store instructions, whole 64-byte pieces per lane.  The upper bound of what a dwordx4-friendly parked layout could buy.
    a2 = "#define B3_PARK_STORES_EIGHTH 8"
    assert a2 in s
    s = s.replace(a2, "#define B3_PARK_STORES_EIGHTH 2")
    a3 = "        for (int r = 8 * s2; r < 8 * s2 + B3_PARK_STORES_EIGHTH; r++) PARK_STORE(acc[t][r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);\n    }\n    uint32_t m = 0;"
    b3 = ("        for (int one_ = 0; one_ < 1; one_++) { float *q_ = ht + (32 * t) * TILE + (pt + 32 * h) * 16 + 8 * s2;\n"
          "          __builtin_nontemporal_store((f32x4){acc[t][8 * s2], acc[t][8 * s2 + 1], acc[t][8 * s2 + 2], acc[t][8 * s2 + 3]}, reinterpret_cast<f32x4 *>(q_));\n"
          "          __builtin_nontemporal_store((f32x4){acc[t][8 * s2 + 4], acc[t][8 * s2 + 5], acc[t][8 * s2 + 6], acc[t][8 * s2 + 7]}, reinterpret_cast<f32x4 *>(q_ + 4)); }\n    }\n    uint32_t m = 0;")
    assert a3 in s, "eighth"
    s = s.replace(a3, b3, 1)
    a4 = "            for (int r = 0; r < 16; r++) PARK_STORE(acc[t][r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);\n        }\n        uint32_t m = 0;"
    b4 = ("            for (int j_ = 0; j_ < 4; j_++) __builtin_nontemporal_store((f32x4){acc[t][4 * j_], acc[t][4 * j_ + 1], acc[t][4 * j_ + 2], acc[t][4 * j_ + 3]},\n"
          "                reinterpret_cast<f32x4 *>(ht + (32 * t) * TILE + (pt + 32 * h) * 16 + 4 * j_));\n        }\n        uint32_t m = 0;")
    assert a4 in s, "half"
    s = s.replace(a4, b4, 1)
elif name == "nopad":
    pass                                                                                      # = HEAD since the padding rows went
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
