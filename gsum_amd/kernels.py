"""scikit-learn kernel tree -> ``gsum_kernel_desc`` (the POD the HIP kernel-build kernels consume).

The reference calls ``kernel(X)`` / ``kernel(X, Y)`` on arbitrary scikit-learn kernels
(gsum/models.py:146-147, 686-688, 708, 822-824, 958-960).  Two descriptor forms:

* flattened -- ``[ConstantKernel *] (RBF | Matern(nu in {0.5, 1.5, 2.5})) [+ WhiteKernel] [+ ConstantKernel]``, the family the
  reference's own tests, notebooks and defaults use; it runs the templated fast kernels and the one-workgroup-per-evaluation paths;
* tree (round 4) -- any ``Sum`` / ``Product`` / ``Exponentiation`` tree over RBF, Matern(0.5, 1.5, 2.5, inf), RationalQuadratic,
  ExpSineSquared, DotProduct, ConstantKernel and WhiteKernel leaves (``RBF + RBF``, ``C * RBF + C * Matern``, ``ExpSineSquared * RBF``,
  ``RBF ** 2`` ...): a postfix program in scikit-learn's own evaluation order, at most 4 stationary leaves and 24 operations.

Other leaves (PairwiseKernel, Matern with another nu: Bessel functions) raise
``NotImplementedError`` -- on the 'hip' backend there is deliberately no host fallback.  ``theta`` handling (log-parameters, ordering) is scikit-learn's: leaves left to right, a leaf's
hyperparameters in alphabetical order.  :func:`describe_thetas` / :func:`describe_gradients` reproduce the
setter's values without a clone per theta.
"""
from __future__ import annotations

import numpy as np
from sklearn.gaussian_process.kernels import (RBF, ConstantKernel, DotProduct, Exponentiation, ExpSineSquared, Matern, Product,
                                              RationalQuadratic, Sum, WhiteKernel)

from ._lib import (FAMILY, GSUM_MAX_D, GSUM_MAX_LEAVES, GSUM_MAX_OPS, OP_ADD, OP_CONST, OP_LEAF, OP_MUL, OP_POW, OP_WHITE, GradParam,
                   KernelDesc)

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


# ---- general trees ----------------------------------------------------------------------------------------------------------------
class _TreeNode:
    """A leaf of the postfix program with where its free hyperparameters sit in theta."""
    __slots__ = ("kind", "kernel", "slot", "offsets")

    def __init__(self, kind, kernel, slot, offsets):
        self.kind, self.kernel, self.slot, self.offsets = kind, kernel, slot, offsets


def _compile_tree(kernel):
    """Postfix program of a Sum / Product tree: [("leaf" | "const" | "white", node) | "add" | "mul", ...] and theta's length.
    theta offsets follow scikit-learn: k1's parameters before k2's, a leaf's free hyperparameters in alphabetical order."""
    prog, offset, counts = [], [0], {"leaf": 0, "cval": 0}

    def free(hyper, off):
        return None if hyper.fixed else off

    def walk(k):
        if isinstance(k, (Sum, Product)):
            walk(k.k1)
            walk(k.k2)
            prog.append("add" if isinstance(k, Sum) else "mul")
            return
        if isinstance(k, Exponentiation):            # theta is the inner kernel's (kernels.py Exponentiation.theta); the exponent is a plain parameter
            walk(k.kernel)
            prog.append(_TreeNode("pow", k, counts["cval"], {}))
            counts["cval"] += 1
            return
        if isinstance(k, WhiteKernel):
            h = k.hyperparameter_noise_level
            node = _TreeNode("white", k, counts["cval"], {"value": free(h, offset[0])})
            counts["cval"] += 1
            offset[0] += 0 if h.fixed else 1
        elif isinstance(k, ConstantKernel):
            h = k.hyperparameter_constant_value
            node = _TreeNode("const", k, counts["cval"], {"value": free(h, offset[0])})
            counts["cval"] += 1
            offset[0] += 0 if h.fixed else 1
        elif isinstance(k, RationalQuadratic):
            ha, hl = k.hyperparameter_alpha, k.hyperparameter_length_scale          # alphabetical: alpha, length_scale
            offs = {"alpha": free(ha, offset[0])}
            offset[0] += 0 if ha.fixed else 1
            offs["length_scale"] = free(hl, offset[0])
            offset[0] += 0 if hl.fixed else 1
            node = _TreeNode("leaf", k, counts["leaf"], offs)
            counts["leaf"] += 1
        elif isinstance(k, DotProduct):
            h = k.hyperparameter_sigma_0
            node = _TreeNode("leaf", k, counts["leaf"], {"sigma_0": free(h, offset[0])})
            counts["leaf"] += 1
            offset[0] += 0 if h.fixed else 1
        elif isinstance(k, ExpSineSquared):
            hl, hp = k.hyperparameter_length_scale, k.hyperparameter_periodicity    # alphabetical: length_scale, periodicity
            offs = {"length_scale": free(hl, offset[0])}
            offset[0] += 0 if hl.fixed else 1
            offs["periodicity"] = free(hp, offset[0])
            offset[0] += 0 if hp.fixed else 1
            node = _TreeNode("leaf", k, counts["leaf"], offs)
            counts["leaf"] += 1
        elif isinstance(k, (RBF, Matern)):
            if isinstance(k, Matern) and float(k.nu) not in (0.5, 1.5, 2.5, np.inf):
                raise NotImplementedError(f"Matern nu={k.nu} is not supported on the device (0.5, 1.5, 2.5 and inf are)")
            h = k.hyperparameter_length_scale
            node = _TreeNode("leaf", k, counts["leaf"], {"length_scale": free(h, offset[0]), "n": h.n_elements})
            counts["leaf"] += 1
            offset[0] += 0 if h.fixed else h.n_elements
        else:
            raise NotImplementedError(f"kernel {k!r} is not supported on the device (Sum / Product / Exponentiation trees over RBF, Matern, "
                                      "RationalQuadratic, ExpSineSquared, DotProduct, ConstantKernel and WhiteKernel are)")
        prog.append(node)

    walk(kernel)
    # (a kernel of ConstantKernel / WhiteKernel terms alone -- c 1 1^T + w I -- is a legal scikit-learn kernel and a program without leaves)
    if counts["leaf"] > GSUM_MAX_LEAVES or len(prog) > GSUM_MAX_OPS:
        raise NotImplementedError(f"kernel {kernel} is too large for the device descriptor ({GSUM_MAX_LEAVES} stationary leaves, "
                                  f"{GSUM_MAX_OPS} operations)")
    return prog, offset[0]


def _tree_values(node, theta):
    """Hyperparameter values of a leaf for a theta (None: the kernel's own), formed like scikit-learn's setter forms them."""
    k = node.kernel

    def val(name, own):
        off = node.offsets.get(name)
        if theta is None or off is None:
            return own
        n = node.offsets.get("n", 1) if name == "length_scale" else 1
        return np.exp(theta[off:off + n]) if n > 1 else np.exp(theta[off])

    if node.kind == "white":
        return val("value", k.noise_level)
    if node.kind == "const":
        return val("value", k.constant_value)
    if node.kind == "pow":
        return float(k.exponent)
    if isinstance(k, RationalQuadratic):
        return val("length_scale", k.length_scale), val("alpha", k.alpha)
    if isinstance(k, ExpSineSquared):
        return val("length_scale", k.length_scale), val("periodicity", k.periodicity)
    if isinstance(k, DotProduct):
        return val("sigma_0", k.sigma_0), None
    return val("length_scale", k.length_scale), None


def _describe_tree(prog, theta, n_features, shown) -> KernelDesc:
    if n_features < 1 or n_features > GSUM_MAX_D:
        raise ValueError(f"number of features must be 1..{GSUM_MAX_D}, got {n_features}")
    desc = KernelDesc()
    desc.amplitude = 1.0
    n_leaves = 0
    for i, item in enumerate(prog):
        if item == "add":
            desc.op[i] = OP_ADD
        elif item == "mul":
            desc.op[i] = OP_MUL
        elif item.kind == "pow":
            desc.cval[item.slot] = float(_tree_values(item, theta))
            desc.op[i] = OP_POW + item.slot
        elif item.kind in ("white", "const"):
            desc.cval[item.slot] = float(_tree_values(item, theta))
            desc.op[i] = (OP_WHITE if item.kind == "white" else OP_CONST) + item.slot
        else:
            ls, alpha = _tree_values(item, theta)
            lf = desc.leaf[item.slot]
            k = item.kernel
            if isinstance(k, RationalQuadratic):
                lf.family, lf.alpha = FAMILY["rq"], float(alpha)
            elif isinstance(k, ExpSineSquared):
                lf.family, lf.alpha = FAMILY["expsine"], float(alpha)          # (the leaf's second parameter: the periodicity)
            elif isinstance(k, DotProduct):
                lf.family = FAMILY["dot"]                                      # (sigma_0 travels in length_scale[0])
            elif isinstance(k, Matern):
                lf.family = FAMILY[{0.5: "matern12", 1.5: "matern32", 2.5: "matern52", np.inf: "matern_inf"}[float(k.nu)]]
            else:
                lf.family = FAMILY["rbf"]
            ls = np.atleast_1d(np.asarray(ls, dtype=float))
            if ls.size == 1:
                lf.anisotropic = 0
                lf.length_scale[0] = float(ls[0])
            else:
                if ls.size != n_features:
                    raise ValueError(f"anisotropic kernel has {ls.size} length scales, X has {n_features} features")
                lf.anisotropic = 1
                for m, v in enumerate(ls):
                    lf.length_scale[m] = float(v)
            desc.op[i] = OP_LEAF + item.slot
            n_leaves += 1
    desc.n_ops, desc.n_leaves = len(prog), n_leaves
    return desc


def _tree_gradient_params(prog, theta, n_features):
    """One GradParam per element of theta, in theta's order."""
    slots = {}

    def gp(code, dim=0, weight=0.0):
        g = GradParam()
        g.code, g.dim, g.weight = code, dim, float(weight)
        return g

    for item in prog:
        if isinstance(item, str):
            continue
        if item.kind == "pow":
            continue
        if item.kind in ("white", "const"):
            off = item.offsets["value"]
            if off is not None:
                slots[off] = [gp(GradParam.TREE_WHITE if item.kind == "white" else GradParam.TREE_CONST, item.slot, _tree_values(item, theta))]
        elif isinstance(item.kernel, RationalQuadratic):
            if item.offsets["alpha"] is not None:
                slots[item.offsets["alpha"]] = [gp(GradParam.TREE_ALPHA, item.slot * 16)]
            if item.offsets["length_scale"] is not None:
                slots[item.offsets["length_scale"]] = [gp(GradParam.TREE_LENGTH_ISO, item.slot * 16)]
        elif isinstance(item.kernel, DotProduct):
            if item.offsets["sigma_0"] is not None:
                slots[item.offsets["sigma_0"]] = [gp(GradParam.TREE_LENGTH_ISO, item.slot * 16)]
        elif isinstance(item.kernel, ExpSineSquared):
            if item.offsets["length_scale"] is not None:
                slots[item.offsets["length_scale"]] = [gp(GradParam.TREE_LENGTH_ISO, item.slot * 16)]
            if item.offsets["periodicity"] is not None:
                slots[item.offsets["periodicity"]] = [gp(GradParam.TREE_ALPHA, item.slot * 16)]       # the leaf's second parameter
        else:
            off = item.offsets["length_scale"]
            if off is not None:
                if item.offsets["n"] > 1:
                    slots[off] = [gp(GradParam.TREE_LENGTH_DIM, item.slot * 16 + m) for m in range(n_features)]
                else:
                    slots[off] = [gp(GradParam.TREE_LENGTH_ISO, item.slot * 16)]
    out = []
    for off in sorted(slots):
        out.extend(slots[off])
    return out


def _is_flat(kernel, n_features):
    """Does the flattened descriptor cover this kernel?"""
    try:
        terms, _ = _flatten(kernel)
        _describe(terms, lambda leaf: leaf.value, n_features, kernel)
        return True
    except NotImplementedError:
        return False


# ---- the walk over a kernel, once per kernel ---------------------------------------------------------------------------------------
# An objective evaluation of ``fit`` describes the SAME kernel object at another theta, tens of times; the walk over scikit-learn's
# objects (every ``hyperparameter_*`` property builds a namedtuple) costs 30-150 us, at the reference's own sizes as much as the device
# call.  The result of the walk is kept per (fingerprint of the kernel, n_features): the fingerprint is every node's class and every
# leaf's parameters and bounds read from its ``__dict__`` (plain attributes: a few us), so an edited kernel is another key, and a hit is
# used only while the kernel it was compiled from still has that fingerprint (its nodes are what the cached walk reads values from).
_COMPILED = {}
_COMPILED_MAX = 128


def _fingerprint(k):
    if isinstance(k, (Sum, Product)):
        return (type(k), _fingerprint(k.k1), _fingerprint(k.k2))
    if isinstance(k, Exponentiation):
        return (type(k), float(k.exponent), _fingerprint(k.kernel))
    items = []
    for name, v in k.__dict__.items():
        if isinstance(v, np.ndarray):
            v = (v.shape, v.tobytes())
        elif isinstance(v, (list, tuple)):
            v = tuple((x.shape, x.tobytes()) if isinstance(x, np.ndarray) else x for x in v)
        items.append((name, v))
    return (type(k), tuple(items))


def _compiled(kernel, n_features):
    """(is_flat, terms | prog, n_dims) of a kernel: ``_flatten`` where the flattened descriptor covers it, else ``_compile_tree``
    (which raises NotImplementedError for what the device cannot evaluate)."""
    try:
        key = (_fingerprint(kernel), n_features)
        hit = _COMPILED.get(key)
    except TypeError:                              # (an unhashable parameter: no cache)
        key = hit = None
    if hit is not None and _fingerprint(hit[0]) == key[0]:
        return hit[1]
    if _is_flat(kernel, n_features):
        terms, n_dims = _flatten(kernel)
        out = (True, terms, n_dims)
    else:
        prog, n_dims = _compile_tree(kernel)
        out = (False, prog, n_dims)
    if key is not None:
        if len(_COMPILED) >= _COMPILED_MAX:
            _COMPILED.clear()
        _COMPILED[key] = (kernel, out)
    return out


def describe_kernel(kernel, n_features: int) -> KernelDesc:
    """``kernel`` as a :class:`KernelDesc` for inputs with ``n_features`` columns: the flattened form where it applies, else a tree."""
    flat, walk, _ = _compiled(kernel, n_features)
    if flat:
        return _describe(walk, lambda leaf: leaf.value, n_features, kernel)
    return _describe_tree(walk, None, n_features, kernel)


def describe_thetas(kernel, thetas, n_features: int):
    """``[describe_kernel(kernel.clone_with_theta(t), n_features) for t in thetas]`` without the clones: scikit-learn's theta setter
    costs 70-320 us per call (get_params / set_params over the tree), more than a batched n = 2048 evaluation takes on the device.
    The values are formed exactly as the setter forms them (kernels.py Kernel.theta: ``np.exp(theta[i])`` for a scalar hyperparameter,
    ``np.exp(theta[i:i+n])`` for a vector one), so the descriptors are equal byte for byte (tests/test_host_logic.py)."""
    flat, walk, n_dims = _compiled(kernel, n_features)     # the family check first: an unsupported tree says so, not "wrong theta size"
    if not flat:
        prog = walk
        out = []
        for theta in thetas:
            theta = np.atleast_1d(np.asarray(theta, dtype=float))
            if theta.ndim != 1 or theta.size != n_dims:
                raise ValueError("theta has not the correct number of entries. Should be %d; given are %d" % (n_dims, theta.size))
            out.append(_describe_tree(prog, theta, n_features, kernel))
        return out
    terms = walk
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
    flat, walk, n_dims = _compiled(kernel, n_features)
    if not flat:
        out = _tree_gradient_params(walk, None, n_features)
        if len(out) != n_dims or n_dims != len(kernel.theta):
            raise NotImplementedError(f"could not map theta of {kernel} onto device gradient parameters")
        return out
    terms = walk
    out = _gradient_params(terms, lambda leaf: leaf.value, n_features)
    if len(out) != n_dims or n_dims != len(kernel.theta):
        raise NotImplementedError(f"could not map theta of {kernel} onto device gradient parameters")
    return out


def describe_gradients(kernel, thetas, n_features: int):
    """``[describe_gradient(kernel.clone_with_theta(t), n_features) for t in thetas]`` without the clones (see
    :func:`describe_thetas`): the weights of the additive / white parameters are the hyperparameter VALUES, so there is one list
    per theta."""
    flat, walk, n_dims = _compiled(kernel, n_features)
    if not flat:
        prog = walk
        out = []
        for theta in thetas:
            theta = np.atleast_1d(np.asarray(theta, dtype=float))
            if theta.ndim != 1 or theta.size != n_dims:
                raise ValueError("theta has not the correct number of entries. Should be %d; given are %d" % (n_dims, theta.size))
            out.append(_tree_gradient_params(prog, theta, n_features))
        return out
    terms = walk
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
