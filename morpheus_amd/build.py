"""Build libmorpheus_hip.so (gfx950) in-tree with hipcc.

`python -m morpheus_amd.build` or morpheus_amd.build.build().  hipcc cross-compiles without a
GPU; the .so lands in morpheus_amd/_build/ (git-ignored, shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
SO = os.path.join(OUT_DIR, "libmorpheus_hip.so")
SOURCES = ["hashgrid.hip", "composite.hip", "sampler.hip", "mlp.hip", "optim.hip", "wnorm.hip"]


def _stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "morpheus_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
