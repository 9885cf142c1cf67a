"""Build libmorpheus_hip.so (gfx950) in-tree with hipcc.

`python -m morpheus_amd.build` or morpheus_amd.build.build().  hipcc cross-compiles without a
GPU; the .so lands in morpheus_amd/_build/ (git-ignored, shipped to the GPU box by gpurun).
Each source is compiled to its own object (in parallel, only when it or a header changed), then linked.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
SO = os.path.join(OUT_DIR, "libmorpheus_hip.so")
SOURCES = ["hashgrid.hip", "composite.hip", "sampler.hip", "mlp.hip", "mlp_b3.hip", "optim.hip", "wnorm.hip", "normal.hip", "graph.hip", "losses.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "morpheus_hip.h")]
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar fp32 adds / muls of the epilogues into v_pk_* instructions, which
# cost more than two plain ones beside MFMAs (MI355X_MICROARCH.md); measured on one box, whole library, cfg3: 15.61 -> 15.38
# ms/step (fused field backward 2.27 -> 2.17 ms, warp forward / backward-data -0.045 / -0.04, hash-grid backward -0.05).  Same
# IEEE results instruction for instruction.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"] + os.environ.get("MH_EXTRA_FLAGS", "").split()
# macros of timing experiments that compute WRONG results on purpose (tools/micro/hashgrid_brk_exp.patch, built by
# tools/build_grid_variants.sh into side libraries only) must never reach the product library
if any("BRK_EXP" in f for f in FLAGS):
    raise RuntimeError("MH_EXTRA_FLAGS carries a -DBRK_EXP_* timing-experiment macro: refused for libmorpheus_hip.so "
                       "(tools/build_grid_variants.sh builds those variants)")
# losses.hip restates chains of rounded fp32 operators (the pose correction: R from six sin / cos, then R d): with hipcc's default
# -ffp-contract=fast the compiler fuses its products and sums into FMAs and R moves by an ulp -- enough to move SDF values next to
# zero past the counted parity gate (tests/test_gpu_losses.py compares with the operator chain bit for bit)
FILE_FLAGS = {"losses.hip": ["-ffp-contract=off"]}


def _newer(deps, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _stale() -> bool:
    return _newer([os.path.join(CSRC, f) for f in os.listdir(CSRC)] + HEADERS, SO)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    extra = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".inc")]

    def compile_one(src):
        obj = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        if force or _newer([os.path.join(CSRC, src)] + HEADERS + extra, obj):
            cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
