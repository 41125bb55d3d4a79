"""Host logic shared by the checkers: the sequential kernel-perceptron trainer and the glue that
turns a checker's state (transform, kernel, supports, weights) into a fused HIP score model.

The trainer is inherently sequential (one argmin per iteration) and stays on the host as in the
reference; its only heavy step, filling one kernel row K(x_i, X) per
iteration, is delegated to the kernel callable — for diffco_amd kernels that is the HIP
kernel-matrix kernel with the sample features kept resident on the GPU.
Behaviour restated from the reference: DiffCo.train_perceptron kernel_perceptrons.py:98-137 and
MultiDiffCo.train_perceptron deprecated/MultiDiffCo.py:50-83.
"""
import torch

from . import _ops
from .kernel import FKKernel, KernelFunc


# ----------------------------------------------------------------------------- row fill
class RowFiller:
    """`fill(i)` -> K(x_i, X) as a 1-D tensor on `out_device`.

    For a diffco_amd kernel the features are uploaded once and every row is one
    `dcx_kernel_matrix` launch (B = 1); a foreign callable is simply called like the reference does."""

    def __init__(self, kernel_func, feats, out_device):
        self.kernel_func, self.out_device = kernel_func, out_device
        self.spec = kernel_func.dcx_spec() if isinstance(kernel_func, KernelFunc) and not isinstance(
            kernel_func, FKKernel) else None
        if self.spec is not None:
            self.dev = _ops._device(feats.device)
            self.flat = feats.reshape(len(feats), -1).to(device=self.dev, dtype=torch.float32).contiguous()
            self.dtype = feats.dtype
        else:
            self.feats = feats

    def __call__(self, i):
        if self.spec is not None:
            kind, p0, p1 = self.spec
            row = _ops.kernel_matrix(kind, p0, p1, self.flat[i:i + 1], self.flat)
            return row.reshape(-1).to(device=self.out_device, dtype=self.dtype)
        row = self.kernel_func(self.feats[i], self.feats)
        return row.reshape(-1).to(self.out_device)


# ----------------------------------------------------------------------------- trainer
def _one_class_step(y, hypo, gains, K, fill, beta):
    """One perceptron step for one label column.  Returns True when this column has converged.
    y / hypo / gains are 1-D views that are updated in place."""
    margin = y * hypo
    worst = int(torch.argmin(margin))
    if K[worst, worst] == 0:  # row not computed yet (k(x, x) != 0 marks a filled row)
        row = fill(worst)
        K[worst] = row
        K[:, worst] = row
    if margin[worst] <= 0:
        target = beta ** ((1 + y[worst]) / 2) * y[worst]  # beta scales the positive target
        step = (target - hypo[worst]) / K[worst, worst]
        gains[worst] += step
        hypo += step * K[worst]
        return False
    # every sample is on the right side: try to retire a support that is classified correctly without itself
    active = gains != 0
    slack = y * (hypo - gains * torch.diagonal(K)) * active
    cand = int(torch.argmax(slack))
    if slack[cand] > 0 and int(active.sum()) > 1:
        hypo -= gains[cand] * K[cand]
        gains[cand] = 0
        return False
    return True


def train_perceptron(y, hypo, gains, K, fill, beta, max_iteration, progress=None):
    """Run the (multi-)label kernel perceptron in place.  1-D y: one column; 2-D y [N, C]: the
    columns take one step each per outer iteration and share K.  Returns iterations used."""
    if y.ndim == 1:
        it = 0
        for it in range(max_iteration):
            if progress is not None:
                progress.update(1)
            if _one_class_step(y, hypo, gains, K, fill, beta):
                break
        return it
    C = y.shape[1]
    done = torch.zeros(C, dtype=torch.bool)
    it = 0
    for it in range(max_iteration):
        if progress is not None:
            progress.update(1)
        for c in range(C):
            if _one_class_step(y[:, c], hypo[:, c], gains[:, c], K, fill, beta):
                done[c] = True
        if bool(done.all()):
            break
    return it


# ----------------------------------------------------------------------------- fused-model glue
def transform_desc(transform):
    """FK description when `transform` is the bound `fkine` of a diffco_amd.model robot, else None."""
    owner = getattr(transform, "__self__", None)
    if owner is not None and getattr(transform, "__name__", "") == "fkine":
        fk = getattr(owner, "fk_desc", None)
        if callable(fk):
            return fk()
    return None


def kernel_spec(kernel_func):
    spec = kernel_func.dcx_spec() if isinstance(kernel_func, KernelFunc) else None
    if spec is None:
        raise TypeError(
            f"{type(kernel_func).__name__} is not a diffco_amd kernel: the score path is HIP-only and supports "
            "kernel.RQKernel / Polyharmonic / MultiQuadratic (optionally wrapped in kernel.FKKernel)")
    return spec


class FusedScorer:
    """Caches the ScoreModel built from (transform, kernel, supports, weights) and rebuilds it when
    any of them changes (identity or in-place version)."""

    def __init__(self):
        self._key, self._model = None, None

    @staticmethod
    def _tkey(t):
        return (t.data_ptr(), t._version, tuple(t.shape), t.dtype, str(t.device))

    def model(self, transform, kernel_func, support_feat, weights, device=None):
        spec = kernel_spec(kernel_func)
        desc = transform_desc(transform)
        key = (None if desc is None else desc.key(), spec, self._tkey(support_feat), self._tkey(weights), str(device))
        if key != self._key:
            self._model = _ops.ScoreModel(desc, spec[0], spec[1], spec[2], support_feat, weights, device=device)
            self._key = key
        return self._model

    def score(self, transform, kernel_func, support_feat, weights, point):
        """[B, C] = K(T(point), supports) @ weights, differentiable w.r.t. `point`."""
        dev = point.device if point.device.type == "cuda" else (
            support_feat.device if support_feat.device.type == "cuda" else None)
        m = self.model(transform, kernel_func, support_feat, weights, dev)
        if m.desc.kind == 0 and transform is not None:
            # foreign transform: run it in torch (its own autograd), fuse everything after it
            feats = transform(point)
            return m.score(feats.reshape(len(feats), -1))
        return m.score(point.reshape(-1, m.dof))

    def invalidate(self):
        self._key, self._model = None, None
