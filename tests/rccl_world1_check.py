"""Child process of tests/test_gpu_config4.py::test_grid_gather_under_an_initialised_rccl_group (not collected by pytest):
initialises a world-size-1 process group on the nccl backend (RCCL), runs the sharded likelihood grid with the HIP
evaluator under it and checks the gathered result against the unsharded call and the reference's golden values."""
import functools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

g = json.load(open(os.path.join(ROOT, "tests", "golden", "cbar_ratio_grid.json")))
n, r = g["n"], g["r"]
X = g["dx"] * np.arange(n)[:, None]
K = RBF(g["length_scale"])(X)
K[np.diag_indices_from(K)] += g["nugget"]
c = np.linalg.cholesky(K) @ np.random.RandomState(g["seed"]).randn(n, r)
y = gsum_amd.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    gp = gsum_amd.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.fit(X, y, orders=np.arange(r))
    thetas = [np.log([e]) for e in g["ells"]]
    fn = functools.partial(gp.log_marginal_likelihood_grid, thetas, g["ratios"], mode="full")
    got = gsum_amd.lml_grid_distributed(fn, len(g["ratios"]), len(thetas))
    assert dist.get_backend() == "nccl"
    np.testing.assert_array_equal(got, fn())                                     # gathered == unsharded
    want = np.array(g["strip_ratio_by_ell"])
    for jj, e in enumerate(g["ells"]):                                            # cond(R) grows from 1e6 to 1e11 along the strip
        Kj = RBF(e)(X) + 1e-10 * np.eye(n)
        np.testing.assert_allclose(got[:, jj], want[:, jj], rtol=max(1e-10, 1e-15 * np.linalg.cond(Kj)))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    assert float(t.sum()) == 4.0
    print("RCCL_WORLD1_OK backend=%s" % dist.get_backend(), flush=True)
finally:
    dist.destroy_process_group()
