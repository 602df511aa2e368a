#!/usr/bin/env python3
"""Mixed soak on ONE context (the streams are shared between the schedules since round 4): batch calls, single evaluations (persistent chain),
value + gradient evaluations alone and in batches, operator-level factorisations, in random order for a given time; every result must
equal the first result of its kind bit for bit, no chain time-out may occur.  Round 5 adds: long calls (two cohorts per group), calls with
right-hand-side sets, predictive sweeps (look-ahead) on a kept factor, the device group over [this GPU] with the RCCL gather, and -- with
n >= 10240 -- the deep-grouped single factorisation.     python tools/gpu_mixed_soak.py [seconds] [n]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from gsum_amd.kernels import describe_gradient, describe_kernel  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = gsum_amd.HipContext(0)
rng = np.random.default_rng(1)
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
ctx.set_inputs(X, Z)
descs = [describe_kernel(RBF(0.19 + 1e-3 * i), 1) for i in range(24)]
gk = [C(1.0) * RBF(0.2 + 0.01 * i) + WhiteKernel(1e-8) for i in range(5)]
gd, gp = [describe_kernel(k, 1) for k in gk], [describe_gradient(k, 1) for k in gk]
ref, counts = {}, {}
Zs = np.concatenate([np.random.RandomState(5).randn(3, n, 6), np.ones((3, n, 1))], axis=2)
Zs[0] = Z
set_of = np.arange(24) % 3
Xs = 0.1 * n * np.random.RandomState(7).rand(1100, 1)
grp = gsum_amd.HipGroup([0], own=False) if False else None          # (the group adopts the default context: this soak owns its own -> gsum_group_adopt below)
from gsum_amd import _lib as _l  # noqa: E402
import ctypes as _C  # noqa: E402
_h = _l._p()
_arr = (_l._p * 1)(ctx._h)
assert ctx._lib.gsum_group_adopt(1, _arr, _C.byref(_h)) == 0
many = ctx.desc_array([describe_kernel(RBF(0.19 + 2e-4 * i), 1) for i in range(100)])
kept = {}


def check(kind, key, out):
    flat = np.concatenate([np.asarray(o, dtype=float).ravel() for o in out])
    assert np.all(np.isfinite(flat)), (kind, key)
    if (kind, key) in ref:
        assert np.array_equal(ref[(kind, key)], flat), f"{kind} {key}: result changed"
    else:
        ref[(kind, key)] = flat
    counts[kind] = counts.get(kind, 0) + 1


t_end = time.time() + seconds
while time.time() < t_end:
    what = rng.integers(0, 10)
    if what == 0:
        k = int(rng.choice([3, 7, 20, 24]))
        check("batch", k, ctx.lml_resident(descs[:k], 1e-10))
    elif what == 1:
        i = int(rng.integers(0, 3))
        check("single", i, ctx.lml_resident([descs[i]], 1e-10))
    elif what == 2:
        i = int(rng.integers(0, 2))
        check("grad", i, ctx.lml_grad(gd[i], gp[i], X, Z, 1e-10))
    elif what == 3:
        check("gradbatch", 5, ctx.lml_grad_batch(gd, gp, X, Z, 1e-10))
    elif what == 4:
        K, info = ctx.factorize(descs[0], X, diag_add=1e-10)
        G, sld = ctx.forward_gram(K, Z)
        K.free()
        check("factorize", 0, (G, [sld, info]))
    elif what == 5:
        check("two", 0, ctx.lml_resident(descs[:2], 1e-10))
    elif what == 6:                                   # a long call: two cohorts per group (n_kernels >= 96)
        check("long", 100, ctx.lml_resident(many, 1e-10))
    elif what == 7:                                   # right-hand-side sets (then the plain inputs again: set 0 is Z)
        ctx.set_inputs_sets(X, Zs)
        k = int(rng.choice([2, 9, 24]))
        check("sets", k, ctx.lml_resident_sets(descs[:k], set_of[:k], 1e-10))
    elif what == 8:                                   # predictive sweep on a kept factor (look-ahead sweep from 1024 new points up)
        if "L" not in kept:
            kept["L"], info = ctx.factorize(descs[0], X, diag_add=1e-10)
            assert info == 0
        check("predict", 0, ctx.predict_terms(kept["L"], descs[0], X, Xs, rhs=Z)[:2])
    else:                                             # the device group over this context, RCCL gather
        nk, k_ = 9, Z.shape[1]
        G, sld, info = np.empty((nk, k_, k_)), np.empty(nk), np.zeros(nk, dtype=np.int64)
        rc = ctx._lib.gsum_group_lml_resident(_h, ctx.desc_array(descs[:nk]), nk, 1e-10, _l._ptr(G), _l._ptr(sld), info.ctypes.data_as(_l._ip), 1)
        assert rc == 0, ctx._lib.gsum_group_last_error(_h)
        check("group", nk, (G, sld, info))
if "L" in kept:
    kept["L"].free()
ctx._lib.gsum_group_destroy(_h)
print(json.dumps({"n": n, "seconds": seconds, "calls": counts, "chain_aborts": ctx.get_option("chain_aborts"), "chain_probe": ctx.get_option("chain_probe"),
                  "chain_persist": ctx.get_option("chain_persist")}), flush=True)
assert ctx.get_option("chain_aborts") == 0
