#!/usr/bin/env python3
"""Randomised differential check of the fused likelihood paths (run on the GPU box):  gpu_fuzz.py [cases] [seed]

For random (n, d, kernel family, amplitude / white / additive terms, number of right-hand sides, batch size):
  * the small (n <= 128) / medium (128 < n <= 4096) one-workgroup paths against the multi-kernel path
    (bit-identical above 128, rounding-level below, identical info codes);
  * the multi-kernel path against numpy (LAPACK Cholesky + solves) with tolerances scaled by cond(R): two valid fp64
    factorisations differ by ~eps cond(R) in the Gram matrix and in sum log L_ii.
Prints one line per failure and a summary; exit code 1 on any failure."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = gsum_amd.lab_context(0)
SPECIAL = [1, 2, 5, 16, 17, 127, 128, 129, 130, 255, 256, 257, 383, 384, 385, 511, 512, 513, 1000, 1024, 1025]
fails = 0
for it in range(cases):
    n = int(rng.choice(SPECIAL)) if rng.rand() < 0.5 else int(rng.randint(1, 1400))
    d = int(rng.randint(1, 9))
    k = int(rng.randint(1, 17))
    fam = rng.choice(["rbf", "m52", "m32", "m12"])
    ls = rng.uniform(0.3, 1.5, size=d) if rng.rand() < 0.5 else float(rng.uniform(0.3, 1.5))
    base = RBF(ls) if fam == "rbf" else Matern(ls, nu={"m52": 2.5, "m32": 1.5, "m12": 0.5}[fam])
    kern = base
    if rng.rand() < 0.5:
        kern = C(float(rng.uniform(0.3, 3.0))) * kern
    if rng.rand() < 0.5:
        kern = kern + WhiteKernel(float(10 ** rng.uniform(-8, -2)))
    if rng.rand() < 0.3:
        kern = kern + C(float(rng.uniform(0.05, 1.0)))
    nugget = float(10 ** rng.uniform(-10, -4))
    X = rng.rand(n, d) * (2.0 + 4.0 * rng.rand()) * max(1.0, n ** (1.0 / d) / 6.0)
    Z = rng.randn(n, k)
    nb = int(rng.randint(1, 6))
    descs = [gsum_amd.describe_kernel(kern.clone_with_theta(kern.theta + 0.05 * j), d) for j in range(nb)]
    tag = f"case {it}: n={n} d={d} k={k} {fam} nb={nb}"
    try:
        ctx.set_inputs(X, Z)
        ctx.set_option("small_path", 1); ctx.set_option("medium_path", 1); ctx.set_option("medium_min_batch", 1)
        fused = ctx.lml_resident(descs, nugget)
        ctx.set_option("small_path", 0); ctx.set_option("medium_path", 0)
        gen = ctx.lml_resident(descs, nugget)
        ctx.set_option("small_path", 1); ctx.set_option("medium_path", 1); ctx.set_option("medium_min_batch", -1)
        if not np.array_equal(fused[2], gen[2]):
            print("FAIL info", tag, fused[2], gen[2]); fails += 1; continue
        ok = gen[2] == 0
        if n > 128:
            if not (np.array_equal(fused[0][ok], gen[0][ok]) and np.array_equal(fused[1][ok], gen[1][ok])):
                print("FAIL medium != general", tag); fails += 1; continue
        R = kern(X) + nugget * np.eye(n)
        cond = np.linalg.cond(R)
        tol = max(1e-11, 1e-15 * cond)
        if n <= 128 and ok.all():
            if not (np.allclose(fused[0], gen[0], rtol=100 * tol, atol=100 * tol * np.abs(gen[0]).max())
                    and np.allclose(fused[1], gen[1], rtol=1e-12, atol=1e-11)):
                print("FAIL small vs general", tag, np.abs(fused[0] - gen[0]).max()); fails += 1; continue
        if ok[0] and cond < 1e13:
            L = np.linalg.cholesky(R)
            W = np.linalg.solve(L, Z)
            G = W.T @ W
            sld = np.log(np.diag(L)).sum()
            if not (np.allclose(gen[0][0], G, rtol=100 * tol, atol=100 * tol * np.abs(G).max()) and abs(gen[1][0] - sld) <= max(1e-10 * max(1.0, abs(sld)), 1e-14 * cond)):
                print("FAIL vs numpy", tag, "cond %.1e" % cond, np.abs(gen[0][0] - G).max() / np.abs(G).max(), gen[1][0] - sld); fails += 1; continue
    except Exception as exc:
        print("EXC", tag, repr(exc)); fails += 1
    if (it + 1) % 25 == 0:
        print(f"... {it + 1} cases, {fails} failures", flush=True)
print(f"fuzz: {cases} cases, {fails} failures", flush=True)
sys.exit(1 if fails else 0)
