"""Pairwise kernel functions with the reference's call signature `K(xs, x_primes) -> Tensor[B, S]`.

Drop-in for diffco/kernel.py (RQKernel 12-29, MultiQuadratic 45-57, Polyharmonic 59-79,
FKKernel 131-143): same constructor arguments, same shape rules (trailing dims flattened,
RQ squeezes a single query row, Polyharmonic does not).  The values come from the HIP
kernel-matrix kernel (libdcx.so `dcx_kernel_matrix`); inside DiffCo.score / poly_score /
rbf_score the kernel is never materialised — those fuse it with the FK and the weight
contraction (`dcx_score*`).  No CPU arithmetic lives here.
"""

from . import _ops
from ._fkdesc import DCX_K_MQ, DCX_K_POLY, DCX_K_RQ


class KernelFunc:
    """Base class.  Subclasses that define `dcx_spec()` run on the fused HIP path."""

    def __call__(self, xs, x_primes):
        raise NotImplementedError("You need to define your own __call__ function.")

    def dcx_spec(self):
        """(kernel_kind, p0, p1) understood by libdcx, or None for a foreign kernel"""
        return None


def _as_rows(xs, x_primes):
    if xs.ndim < x_primes.ndim:  # a single query: promote to a batch of one
        xs = xs.reshape((1,) * (x_primes.ndim - xs.ndim) + tuple(xs.shape))
    return xs.reshape(xs.shape[0], -1), x_primes.reshape(x_primes.shape[0], -1)


class RQKernel(KernelFunc):
    """(1 + gamma/p * ||x - s||^2)^(-p)"""

    def __init__(self, gamma: float, p: int = 2):
        self.gamma = gamma
        self.p = p

    def dcx_spec(self):
        return (DCX_K_RQ, float(self.gamma), float(self.p))

    def __call__(self, xs, x_primes):
        a, b = _as_rows(xs, x_primes)
        k = _ops.kernel_matrix(DCX_K_RQ, self.gamma, self.p, a, b)
        return k.squeeze(0) if k.shape[0] == 1 else k  # reference squeezes one query row (kernel.py:26-27)


class Polyharmonic(KernelFunc):
    """r^k / eps (k odd) or r^k log r / eps (k even, 0 at r = 0)"""

    def __init__(self, k, epsilon):
        if int(k) != k or k < 1:
            raise ValueError("Polyharmonic: k must be a positive integer")
        self.k = int(k)
        self.epsilon = epsilon

    def dcx_spec(self):
        return (DCX_K_POLY, float(self.k), float(self.epsilon))

    def __call__(self, xs, x_primes):
        a, b = _as_rows(xs, x_primes)
        return _ops.kernel_matrix(DCX_K_POLY, self.k, self.epsilon, a, b)


class MultiQuadratic(KernelFunc):
    """sqrt(||x - s||^2 / eps^2 + 1)"""

    def __init__(self, epsilon):
        self.epsilon = epsilon

    def dcx_spec(self):
        return (DCX_K_MQ, float(self.epsilon), 0.0)

    def __call__(self, xs, x_primes):
        if xs.ndim == 1:
            xs = xs[None, :]
        a, b = _as_rows(xs, x_primes)
        k = _ops.kernel_matrix(DCX_K_MQ, self.epsilon, 0.0, a, b)
        return k.squeeze(0) if k.shape[0] == 1 else k


class FKKernel(KernelFunc):
    """Old-API composition `rq_kernel(fkine(x), fkine(x'))` that scripts/*.py construct
    (scripts/speed_compare.py:220).  Unlike the reference's ctor (kernel.py:133) it does not raise.
    A checker given an FKKernel fuses `fkine` into the HIP score kernel when `fkine` is a bound
    method of a diffco_amd.model robot."""

    def __init__(self, fkine, rq_kernel):
        self.fkine = fkine
        self.rq_kernel = rq_kernel

    def dcx_spec(self):
        return self.rq_kernel.dcx_spec() if isinstance(self.rq_kernel, KernelFunc) else None

    def __call__(self, xs, x_primes=None, x_primes_controls=None):
        if xs.ndim == 1:
            xs = xs[None, :]
        xc = self.fkine(xs).reshape(len(xs), -1)
        if x_primes_controls is None:
            x_primes_controls = self.fkine(x_primes).reshape(len(x_primes), -1)
        return self.rq_kernel(xc, x_primes_controls)
