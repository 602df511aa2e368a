#!/usr/bin/env python3
"""Extended-precision known answers for the large log-likelihood cases -> tests/golden/large_truth.json.

For the S2 / S3 inputs of SURVEY.md 8(c) (white-noise coefficients, the values the reference returns are in
large_lml.json) and their GP-drawn variants (large_lml_gp_drawn.json) this evaluates the SAME function of the SAME
fp64 inputs -- R = sklearn's RBF(0.2)(X) + 1e-10 I as fp64 bit patterns, c = coefficients(y, ...) in fp64 -- with the
Cholesky factorisation, the triangular solves, the Gram matrix and the log-determinant carried out in 80-bit long
double (oracle/truth_ld.c) and the O(k^2) algebra of models.py:1007-1039 in numpy.longdouble.  The result is the value
every fp64 factorisation approximates: |LAPACK - truth| and |HIP - truth| can then be compared instead of
|HIP - LAPACK| alone.  Needs gcc + numpy + scikit-learn only (no reference import).

    python tests/golden/make_truth.py            # ~2 minutes on 8 cores
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import gsum_oracle as orc  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

LD = np.longdouble


def load_lib():
    out = os.path.join(ROOT, "oracle", "_build", "libtruth_ld.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", out, os.path.join(ROOT, "oracle", "truth_ld.c"), "-lm"],
                   check=True)
    lib = C.CDLL(out)
    lib.truth_gram_ld.restype = C.c_int
    lib.truth_gram_ld.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def truth_lml(lib, R, c, jac):
    """models.py:1007-1039 for the default prior (center 0, disp 0, df 1, scale 1) in long double; c = coefficient curves."""
    assert np.finfo(LD).nmant == 63, "long double is not the x87 80-bit format here"
    n, r = c.shape
    Z = np.ascontiguousarray(np.concatenate([c, np.ones((n, 1))], axis=1))
    k = r + 1
    G = np.zeros((k, k), dtype=LD)
    sld = np.zeros(1, dtype=LD)
    info = lib.truth_gram_ld(R.ctypes.data, Z.ctypes.data, n, k, G.ctypes.data, sld.ctypes.data)
    assert info == 0, info
    ny = LD(r)
    N = LD(n)
    tr = np.trace(G[:r, :r])
    df = LD(1) + N * ny                                      # models.py:302
    scale_sq = (LD(1) + tr) / df                             # :448 with center0 = 0, disp0 = 0: quad + quad2 = tr
    var = df * scale_sq / (df - LD(2))                       # :500-503
    S = tr                                                   # eta = 0
    lml = -S / (LD(2) * var) - ny / LD(2) * (N * np.log(var) + LD(2) * sld[0]) - ny * N / LD(2) * np.log(LD(2) * LD(np.pi))
    # LD(np.pi) is the fp64 pi, as the reference's np.log(2 * np.pi) uses (:1038)
    return lml - LD(jac), G, sld[0]


def main():
    lib = load_lib()
    out = []
    for n, r in ((512, 4), (2048, 4), (8192, 6)):
        X = 0.1 * np.arange(n)[:, None]
        R = RBF(0.2)(X)
        R[np.diag_indices_from(R)] += 1e-10
        R = np.ascontiguousarray(R)
        orders = np.arange(r)
        for kind, seed in (("white_noise", 0), ("gp_drawn", 1)):
            z = np.random.RandomState(seed).randn(n, r)
            c0 = z if kind == "white_noise" else np.linalg.cholesky(R) @ z
            y = orc.partials(c0, ratio=0.5, ref=1.0, orders=orders)
            for q in (0.5, 0.45):
                c = orc.coefficients(y, q * np.ones(n), np.ones(n), orders)
                jac = np.sum(r * np.log(np.abs(np.ones(n))) + np.sum(orders) * np.log(np.abs(q * np.ones(n))))
                t, G, sld = truth_lml(lib, R, c, jac)
                lap = orc.trunc_lml(RBF(0.2), np.log([0.2]), X, y, orders, ratio=q, ref=1.0)
                rec = dict(n=n, r=r, kind=kind, seed=seed, ratio=q, lml_truth=repr(t), lml_truth_f64=float(t),
                           lml_lapack=float(lap), lapack_rel_err=float(abs(LD(lap) - t) / abs(t)),
                           sum_log_diag_truth=float(sld), trace_Gyy_truth=float(np.trace(G[:r, :r])))
                out.append(rec)
                print(rec, flush=True)
    with open(os.path.join(HERE, "large_truth.json"), "w") as f:
        json.dump(dict(method="80-bit long double Cholesky / solves / Gram (oracle/truth_ld.c) + numpy.longdouble algebra on the "
                              "fp64 inputs; recipe of the inputs: tests/golden/make_golden.py gen_large / gen_large_gp_drawn",
                       cases=out), f, indent=1)


if __name__ == "__main__":
    main()
