"""gsum_amd — MI355X-native GP hot path of buqeye/gsum (kernel build, jittered Cholesky,
multivariate-normal log-likelihood) behind gsum's ConjugateGaussianProcess / ConjugateStudentProcess / TruncationGP / TruncationTP surface.

Compute runs in libgsum_hip.so (hand-written HIP for gfx950, bound with ctypes); there is no CPU path.
"""
from .series import coefficients, partials, geometric_sum
from .conjugate import (ConjugateGaussianProcess, ConjugateStudentProcess, posterior_from_gram, lml_from_gram,
                        lml_from_gram_batch, student_lml_from_gram, cov_factor)
from .truncation import TruncationGP, TruncationTP
from .kernels import describe_kernel, describe_thetas
from .datasets import (make_gaussian_partial_sums, make_gaussian_partial_sums_uniform,
                       make_gaussian_partial_sums_on_grid, sample_mvn_cholesky)
from .grid import shard_range, gather_flat, lml_grid_distributed, predict_distributed
from ._lib import HipContext, HipGroup, KernelDesc, default_context, default_group, device_count, lab_context, load_library

__version__ = "0.1.0"
__all__ = [
    "coefficients", "partials", "geometric_sum", "ConjugateGaussianProcess", "ConjugateStudentProcess",
    "TruncationGP", "TruncationTP", "posterior_from_gram", "lml_from_gram", "lml_from_gram_batch", "student_lml_from_gram", "cov_factor", "describe_kernel", "describe_thetas", "make_gaussian_partial_sums",
    "make_gaussian_partial_sums_uniform", "make_gaussian_partial_sums_on_grid", "sample_mvn_cholesky", "shard_range", "gather_flat",
    "lml_grid_distributed", "predict_distributed", "HipContext", "HipGroup", "KernelDesc", "default_context", "default_group", "device_count", "lab_context", "load_library",
]
