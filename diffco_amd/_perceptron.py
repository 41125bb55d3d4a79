"""Logic shared by the checkers: the kernel-perceptron trainer and the glue that turns a checker's
state (transform, kernel, supports, weights) into a fused HIP score model.

The trainer is inherently sequential (one argmin per iteration).  For diffco_amd kernels the whole loop runs
in ONE persistent launch on the GPU (`dcx_train_perceptron`, csrc/train_kernels.hip — SURVEY.md §8f-1); for a
foreign kernel callable the same algorithm runs as a host loop below, filling kernel rows by calling it like
the reference does.  Behaviour restated from the reference: DiffCo.train_perceptron
kernel_perceptrons.py:98-137 and MultiDiffCo.train_perceptron deprecated/MultiDiffCo.py:50-83.
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


def device_trainer_spec(kernel_func):
    """(kind, p0, p1) when the training kernel can run inside the persistent device trainer, else None.
    DCX_HOST_TRAINER=1 forces the host loop (A/B and debugging)."""
    import os
    if os.environ.get("DCX_HOST_TRAINER") == "1":
        return None
    if isinstance(kernel_func, KernelFunc) and not isinstance(kernel_func, FKKernel):
        return kernel_func.dcx_spec()
    return None


def run_trainer(kernel_func, feats, y, gains, hypo, K, beta, max_iteration, progress=None, cold=False):
    """Train in place semantics, wherever it is fastest: the persistent device kernel for diffco_amd kernels
    (one launch for the whole loop), the host loop otherwise.  K may be None for a cold start (the N x N matrix is
    then created where the trainer runs).  Returns (gains, hypo, K, iterations)."""
    spec = device_trainer_spec(kernel_func)
    if K is None and (spec is None or not cold):
        K = torch.zeros((len(y), len(y)), dtype=gains.dtype, device=gains.device)  # lazily filled
    if spec is None:
        fill = RowFiller(kernel_func, feats, K.device)
        it = train_perceptron(y, hypo, gains, K, fill, beta, max_iteration, progress)
        return gains, hypo, K, it
    out_dev, out_dtype = gains.device, gains.dtype
    g, h, Kd, it, _ = _ops.train_perceptron_device(spec[0], spec[1], spec[2], beta, feats, y, gains, hypo,
                                                   None if cold else K, max_iteration)
    if progress is not None:
        progress.update(it)
    # gains / hypothesis go back to where the caller keeps them; the N x N kernel matrix stays on the GPU (callers
    # only ever take the support sub-block of it: see sub_block)
    return g.to(device=out_dev, dtype=out_dtype), h.to(device=out_dev, dtype=out_dtype), Kd, it


def solve_system(kernel_func, kmat, rhs):
    """fit_poly's S x S solve: on the GPU for diffco_amd kernels (`_ops.solve`); a foreign kernel callable is user
    code evaluated where its tensors live, like the host trainer above, and its system is solved there too."""
    if isinstance(kernel_func, KernelFunc) and kernel_func.dcx_spec() is not None or isinstance(kernel_func, FKKernel):
        return _ops.solve(kmat, rhs)
    return torch.linalg.solve(kmat, rhs.to(kmat.dtype))


def fit_system(kernel_func, feats, targets, reg=0.0):
    """fit_poly's nodes: (K(feats, feats) + reg I) nodes = targets.  For a diffco_amd kernel on plain feature rows the
    matrix is built and solved on the device in one go (`_ops.fit_nodes`: dcx_kernel_matrix + dcx_solve, nothing but the
    nodes comes back); anything else builds the matrix through the callable and goes through solve_system."""
    spec = kernel_func.dcx_spec() if isinstance(kernel_func, KernelFunc) else None
    if spec is not None and len(feats) >= 1:
        return _ops.fit_nodes(spec[0], spec[1], spec[2], feats, targets, reg)
    kmat = kernel_func(feats, feats)
    if reg:
        kmat = kmat + reg * torch.eye(len(kmat), dtype=kmat.dtype, device=kmat.device)
    return solve_system(kernel_func, kmat, targets.to(kmat.dtype))


def sub_block(K, idx, device, dtype):
    """K[idx][:, idx] gathered where K lives, then moved to (device, dtype)"""
    i = idx.to(K.device)
    return K[i[:, None], i[None, :]].to(device=device, dtype=dtype)


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
    """Caches the ScoreModel built from (transform, kernel, supports, weights) and rebuilds it when any of them
    changes.  The cache holds strong references to the two tensors it was built from and compares by identity plus
    in-place version, so a freed-and-reallocated tensor at the same address can never alias the cached model; edits
    that bypass the version counter (`t.data[...] = ...`, `set_`) must be followed by `invalidate()` — the checkers
    call it from train / fit_poly / to / filter_support_points_.

    Ownership (round 5: an explicit count instead of `sys.getrefcount`).  When the checker's state changes but the model's
    structure (transform, kernel, class count, feature width, device) does not, the cached model is REFILLED in place
    (`ScoreModel.update`: same storage, FK tables, per-stream scratch) - unless somebody holds a lease on it
    (`ScoreModel.acquire()` / `release()`: a ShardedAdamRun while it iterates, an optimiser's constraint terms), in which
    case the holder keeps the rows it started with and the cache builds a new model.  `model()` hands out the cache's own
    model: a caller that keeps it across a later train / fit_poly and needs it unchanged takes a lease.  (A multi-class score's
    backward pass re-sweeps the model: it checks `ScoreModel.revision` and raises if the rows were refilled after its forward.)

    The per-call check compares the transform by IDENTITY (a robot's bound `fkine`): a robot whose FK parameters (link lengths,
    DH table, URDF tree) are edited in place after the checker has scored keeps the same bound method, so such an edit must
    be followed by `invalidate()` - like the `.data` edits above.  The robots of diffco_amd.model / urdf do not change after
    construction."""

    def __init__(self):
        self._key, self._model, self._sup, self._w = None, None, None, None
        self._struct, self._retired = None, None
        self._fast = None   # (transform, kernel_func, device string, spec) of the cached model: the per-call check

    def model(self, transform, kernel_func, support_feat, weights, device=None):
        spec = kernel_spec(kernel_func)
        # the per-call path (an optimiser asks thousands of times between two state changes): same objects, same in-place
        # versions, same kernel parameters -> the cached model, without rebuilding the description key (bytes of a 5 KB struct)
        if (self._model is not None and self._sup is support_feat and self._w is weights and self._fast is not None
                and self._fast[0] is transform and self._fast[1] is kernel_func and self._fast[3] == spec
                and self._fast[2] == (device if device is None else str(device))
                and self._key[2] == support_feat._version and self._key[3] == weights._version
                and self._key[4] == tuple(support_feat.shape) and self._key[5] == tuple(weights.shape)):
            return self._model
        desc = transform_desc(transform)
        key = (None if desc is None else desc.key(), spec, support_feat._version, weights._version,
               tuple(support_feat.shape), tuple(weights.shape), str(device))
        if self._model is None or self._sup is not support_feat or self._w is not weights or key != self._key:
            old = self._model if self._model is not None else self._retired
            width = int(support_feat.reshape(len(support_feat), -1).shape[1]) if len(support_feat) else -1
            struct = (key[0], spec, int(weights.reshape(len(weights), -1).shape[1]), width, str(device))
            # refilled in place only while nobody holds a lease on the model (see the class docstring), and only for the same
            # structure INCLUDING the feature width (a refit on different-width features must rebuild, not raise)
            same_shape = (old is not None and self._struct == struct and width == old.D
                          and len(support_feat) <= 4 * max(old.capacity, 1) and old.leases == 0)
            if same_shape:
                # the same transform / kernel / class count with new supports or weights (train, fit_poly, update): the
                # model is refilled in place (dcx_model_update) - its storage, FK tables and per-stream scratch stay
                self._model = old.update(support_feat, weights)
            else:
                # (room to grow: an active-learning loop adds supports round by round)
                self._model = _ops.ScoreModel(desc, spec[0], spec[1], spec[2], support_feat, weights, device=device,
                                              capacity=len(support_feat) + len(support_feat) // 4)
            self._struct = (key[0], spec, self._model.C, self._model.D, str(device))
            self._retired = None
            self._key, self._sup, self._w = key, support_feat, weights
        self._fast = (transform, kernel_func, device if device is None else str(device), spec)
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
        # the state changed behind the version counters: the next call refills the model (kept aside) instead of rebuilding it
        if self._model is not None:
            self._retired = self._model
        self._key, self._model, self._sup, self._w, self._fast = None, None, None, None, None
