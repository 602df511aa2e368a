"""Build libgsum_hip.so in-tree with hipcc for gfx950.   python -m gsum_amd.build"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "gsum_capi.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "gsum_kernels.hip.h"), os.path.join(ROOT, "include", "gsum_hip.h")]
OUT = os.path.join(HERE, "libgsum_hip.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"), "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
