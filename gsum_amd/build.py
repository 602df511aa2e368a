"""Build the HIP library in-tree with hipcc for gfx950.

    python -m gsum_amd.build [--force] [--lab]

``libgsum_hip.so``      the product: the C ABI of include/gsum_hip.h, nothing else exported
``libgsum_hip_lab.so``  the same sources with -DGSUM_LAB: plus include/gsum_hip_debug.h (diagnostics, probes, microbenchmarks,
                        schedule switches); what tools/ and the schedule-equivalence tests load.  Built by ``--lab`` / ``build(lab=True)``
                        and by ``__graft_entry__.build()``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "gsum_capi.hip")
KERNEL_PARTS = ("common", "build", "diag", "panel", "chain", "gemm_nt", "fused", "tile", "solve", "grad", "probes")
DEPS = [SRC, os.path.join(HERE, "csrc", "gsum_kernels.hip.h"), os.path.join(ROOT, "include", "gsum_hip.h"),
        os.path.join(ROOT, "include", "gsum_hip_debug.h")] + [os.path.join(HERE, "csrc", "kernels", f"{p}.hip.h") for p in KERNEL_PARTS]
HOST_PARTS = ("context", "gemm", "matrices", "potrf", "api_context", "api_operators", "api_fused", "wave", "api_lml", "api_multi", "api_grad", "api_measure")
DEPS += [os.path.join(HERE, "csrc", "host", f"{p}.hip.h") for p in HOST_PARTS]
OUT = os.path.join(HERE, "libgsum_hip.so")
OUT_LAB = os.path.join(HERE, "libgsum_hip_lab.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def up_to_date(out=OUT):
    return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS)


def build(force: bool = False, verbose: bool = False, lab: bool = False) -> str:
    out = OUT_LAB if lab else OUT
    if not force and up_to_date(out):
        return out
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-pthread",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")] + (["-DGSUM_LAB"] if lab else []) + ["-o", out, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--lab" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, lab=True))
