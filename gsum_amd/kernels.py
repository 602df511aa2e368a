"""scikit-learn kernel tree -> ``gsum_kernel_desc`` (the POD the HIP kernel-build kernel consumes).

The reference calls ``kernel(X)`` / ``kernel(X, Y)`` on arbitrary scikit-learn kernels
(gsum/models.py:708, 822-824, 958-960).  The device kernel implements the family the reference's
own tests, notebooks and defaults use:

    [ConstantKernel *] (RBF | Matern(nu in {0.5, 1.5, 2.5}))  [+ WhiteKernel] [+ ConstantKernel]

with isotropic or anisotropic ``length_scale``.  Anything else raises ``NotImplementedError`` —
there is deliberately no host fallback.  ``theta`` handling (log-parameters, ordering) is scikit-learn's:
callers use ``kernel.clone_with_theta(theta)`` and describe the clone, or, for a stack of thetas,
:func:`describe_thetas`, which reproduces the setter's values without the per-theta clone.
"""
from __future__ import annotations

import numpy as np
from sklearn.gaussian_process.kernels import (RBF, ConstantKernel, Matern, Product, Sum, WhiteKernel)

from ._lib import FAMILY, GSUM_MAX_D, GradParam, KernelDesc

__all__ = ["describe_kernel", "describe_thetas", "describe_gradient", "describe_gradients", "default_kernel"]


def default_kernel():
    """The reference's default: ConstantKernel(1, fixed) * RBF(1, fixed).  models.py:146-147."""
    return ConstantKernel(1.0, constant_value_bounds="fixed") * RBF(1.0, length_scale_bounds="fixed")


def _sum_terms(k):
    if isinstance(k, Sum):
        return _sum_terms(k.k1) + _sum_terms(k.k2)
    return [k]


def _product_factors(k):
    if isinstance(k, Product):
        return _product_factors(k.k1) + _product_factors(k.k2)
    return [k]


class _Leaf:
    """One leaf of a kernel tree: its class, the value of its only hyperparameter, where that value sits in theta (None if fixed)."""
    __slots__ = ("kind", "value", "offset", "n_elements", "nu", "shown")

    def __init__(self, k, offset):
        if isinstance(k, WhiteKernel):
            self.kind, self.value, hyper = "white", k.noise_level, k.hyperparameter_noise_level
        elif isinstance(k, ConstantKernel):
            self.kind, self.value, hyper = "const", k.constant_value, k.hyperparameter_constant_value
        elif isinstance(k, (RBF, Matern)):
            self.kind, self.value, hyper = "stationary", k.length_scale, k.hyperparameter_length_scale
        else:
            self.kind, self.value, hyper = "unsupported", None, None
        self.shown = k
        self.nu = float(k.nu) if isinstance(k, Matern) else None          # Matern subclasses RBF: nu tells them apart
        free = hyper is not None and not hyper.fixed
        self.n_elements = hyper.n_elements if free else 0
        self.offset = offset if free else None


def _flatten(kernel):
    """Sum terms -> product factors -> leaves, left to right: the order in which scikit-learn lays theta out (KernelOperator.theta
    splits at k1.n_dims; every leaf of the supported family has exactly one hyperparameter)."""
    offset = 0
    terms = []
    for term in _sum_terms(kernel):
        leaves = []
        for f in _product_factors(term):
            leaf = _Leaf(f, offset)
            offset += leaf.n_elements
            leaves.append(leaf)
        terms.append(leaves)
    return terms, offset


def _describe(terms, values, n_features, shown) -> KernelDesc:
    """``values(leaf)`` -> the leaf's hyperparameter; ``shown`` only feeds error messages."""
    if n_features < 1 or n_features > GSUM_MAX_D:
        raise ValueError(f"number of features must be 1..{GSUM_MAX_D}, got {n_features}")
    desc = KernelDesc()
    desc.amplitude = 1.0
    desc.additive_const = 0.0
    desc.white_noise = 0.0
    base = None
    for factors in terms:
        if len(factors) == 1 and factors[0].kind == "white":
            desc.white_noise += float(values(factors[0]))
            continue
        if len(factors) == 1 and factors[0].kind == "const":
            desc.additive_const += float(values(factors[0]))
            continue
        if all(f.kind == "const" for f in factors):
            # a product of constants is one additive constant (C(2) * C(3) adds 6 everywhere): no stationary part
            desc.additive_const += float(np.prod([values(f) for f in factors]))
            continue
        if base is not None:
            raise NotImplementedError(f"sum of two stationary kernels is not supported on the device: {shown}")
        amp = 1.0
        for f in factors:
            if f.kind == "const":
                amp *= float(values(f))
            elif f.kind == "stationary" and base is None:
                base = f
            else:
                raise NotImplementedError(f"kernel factor {f.shown!r} is not supported on the device (in {shown})")
        desc.amplitude = amp                 # only the term that supplied the stationary base sets the amplitude
    if base is None:
        raise NotImplementedError(f"kernel {shown} has no RBF/Matern part")
    if base.nu is not None:
        fam = {0.5: "matern12", 1.5: "matern32", 2.5: "matern52"}.get(base.nu)
        if fam is None:
            raise NotImplementedError(f"Matern nu={base.nu} is not supported on the device (0.5, 1.5, 2.5 are)")
    else:
        fam = "rbf"
    desc.family = FAMILY[fam]
    ls = np.atleast_1d(np.asarray(values(base), dtype=float))
    if ls.size == 1:
        desc.anisotropic = 0
        desc.length_scale[0] = float(ls[0])
    else:
        if ls.size != n_features:
            raise ValueError(f"anisotropic kernel has {ls.size} length scales, X has {n_features} features")
        desc.anisotropic = 1
        for i, v in enumerate(ls):
            desc.length_scale[i] = float(v)
    return desc


def describe_kernel(kernel, n_features: int) -> KernelDesc:
    """Flatten ``kernel`` into a :class:`KernelDesc` for inputs with ``n_features`` columns."""
    terms, _ = _flatten(kernel)
    return _describe(terms, lambda leaf: leaf.value, n_features, kernel)


def describe_thetas(kernel, thetas, n_features: int):
    """``[describe_kernel(kernel.clone_with_theta(t), n_features) for t in thetas]`` without the clones: scikit-learn's theta setter
    costs 70-320 us per call (get_params / set_params over the tree), more than a batched n = 2048 evaluation takes on the device.
    The values are formed exactly as the setter forms them (kernels.py Kernel.theta: ``np.exp(theta[i])`` for a scalar hyperparameter,
    ``np.exp(theta[i:i+n])`` for a vector one), so the descriptors are equal byte for byte (tests/test_host_logic.py)."""
    describe_kernel(kernel, n_features)          # the family check first: an unsupported tree says so, not "wrong theta size"
    terms, n_dims = _flatten(kernel)
    out = []
    for theta in thetas:
        theta = np.atleast_1d(np.asarray(theta, dtype=float))
        if theta.ndim != 1 or theta.size != n_dims:
            raise ValueError("theta has not the correct number of entries. Should be %d; given are %d" % (n_dims, theta.size))

        def values(leaf, theta=theta):
            if leaf.offset is None:
                return leaf.value
            if leaf.n_elements > 1:
                return np.exp(theta[leaf.offset:leaf.offset + leaf.n_elements])
            return np.exp(theta[leaf.offset])

        out.append(_describe(terms, values, n_features, kernel))
    return out


def _gradient_params(terms, values, n_features):
    """The gradient parameters of a flattened kernel whose leaf hyperparameters are ``values(leaf)``: one per FREE hyperparameter
    element, leaves left to right (scikit-learn's theta order)."""
    out = []

    def gp(code, dim=0, weight=0.0):
        g = GradParam()
        g.code, g.dim, g.weight = code, dim, float(weight)
        return g

    for factors in terms:
        if len(factors) == 1 and factors[0].kind == "white":
            if factors[0].offset is not None:
                out.append(gp(GradParam.WHITE, weight=values(factors[0])))
            continue
        if len(factors) == 1 and factors[0].kind == "const":
            if factors[0].offset is not None:
                out.append(gp(GradParam.ADDITIVE, weight=values(factors[0])))
            continue
        if all(f.kind == "const" for f in factors):
            # d (c1 c2 ...) / d log c_i = the whole product, for every free factor
            value = float(np.prod([values(f) for f in factors]))
            out.extend(gp(GradParam.ADDITIVE, weight=value) for f in factors if f.offset is not None)
            continue
        for f in factors:
            if f.kind == "const":
                if f.offset is not None:
                    out.append(gp(GradParam.AMPLITUDE))
            elif f.offset is not None:                          # RBF / Matern: length_scale is the only free one
                if f.n_elements > 1:
                    out.extend(gp(GradParam.LENGTH_DIM, dim=m) for m in range(n_features))
                else:
                    out.append(gp(GradParam.LENGTH_ISO))
    return out


def describe_gradient(kernel, n_features: int):
    """One :class:`GradParam` per component of ``kernel.theta``, in scikit-learn's order (leaves left to right,
    a leaf's free hyperparameters in alphabetical order): what ``kernel(X, eval_gradient=True)``
    would put in ``K_gradient[:, :, p]``.  Same kernel family as :func:`describe_kernel`."""
    describe_kernel(kernel, n_features)                       # same validation, same error messages
    terms, n_dims = _flatten(kernel)
    out = _gradient_params(terms, lambda leaf: leaf.value, n_features)
    if len(out) != n_dims or n_dims != len(kernel.theta):
        raise NotImplementedError(f"could not map theta of {kernel} onto device gradient parameters")
    return out


def describe_gradients(kernel, thetas, n_features: int):
    """``[describe_gradient(kernel.clone_with_theta(t), n_features) for t in thetas]`` without the clones (see
    :func:`describe_thetas`): the weights of the additive / white parameters are the hyperparameter VALUES, so there is one list
    per theta."""
    describe_kernel(kernel, n_features)
    terms, n_dims = _flatten(kernel)
    out = []
    for theta in thetas:
        theta = np.atleast_1d(np.asarray(theta, dtype=float))
        if theta.ndim != 1 or theta.size != n_dims:
            raise ValueError("theta has not the correct number of entries. Should be %d; given are %d" % (n_dims, theta.size))

        def values(leaf, theta=theta):
            if leaf.offset is None:
                return leaf.value
            if leaf.n_elements > 1:
                return np.exp(theta[leaf.offset:leaf.offset + leaf.n_elements])
            return np.exp(theta[leaf.offset])

        params = _gradient_params(terms, values, n_features)
        if len(params) != n_dims:
            raise NotImplementedError(f"could not map theta of {kernel} onto device gradient parameters")
        out.append(params)
    return out
