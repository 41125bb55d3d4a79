"""torch-facing wrappers of the C ABI (include/dcx.h): device buffers in, device buffers out.

PyTorch is plumbing here (allocation, streams, autograd bookkeeping); all arithmetic of the
score/grad path happens in libdcx.so.  CPU tensors are moved to the GPU, computed there and
moved back to the caller's device and dtype, so reference scripts written for CPU tensors run
unchanged on a GPU box.  Without a GPU every function raises (no CPU fallback).
"""
import ctypes as C

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._fkdesc import DCX_K_MQ, DCX_K_POLY, DCX_K_RQ, FkDesc, none_desc  # noqa: F401


def _device(dev=None):
    _lib.require_gpu()
    if dev is not None and torch.device(dev).type == "cuda":
        d = torch.device(dev)
        return torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())
    return torch.device("cuda", torch.cuda.current_device())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(dev):
    """torch's current stream on `dev` as the C ABI's `void* stream` (the raw handle straight from torch's C layer: building
    a torch.cuda.Stream object to read its handle cost 5 us per call, tools/api_profile.py)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(dev.index))
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _on_device:
    """`with _on_device(dev):` = torch.cuda.device(dev) when another device is current, nothing at all otherwise (the library
    selects the model's device itself; the guard only keeps the CALLER's current device what it was)"""
    __slots__ = ("guard",)

    def __init__(self, dev):
        cur = _raw_device() if _raw_device is not None else torch.cuda.current_device()
        self.guard = None if cur == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


def _f32(t, dev):
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


def _kparams(p0, p1=0.0):
    return (C.c_float * 2)(float(p0), float(p1))


# ----------------------------------------------------------------------------- FK
class _FkineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, desc):
        lib = _lib.require_gpu()
        dev = _device(q.device)
        q32 = _f32(q.reshape(-1, desc.dof), dev)
        B = q32.shape[0]
        X = torch.empty((B, desc.feature_dim), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.dcx_fkine(dev.index, C.byref(desc), _ptr(q32), B, _ptr(X), _stream(dev)))
        ctx.desc, ctx.dev, ctx.in_dtype, ctx.in_device, ctx.in_shape = desc, dev, q.dtype, q.device, q.shape
        ctx.save_for_backward(q32)
        return X.reshape(B, *desc.feature_shape).to(device=q.device, dtype=q.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gX):
        lib = _lib.require_gpu()
        (q32,) = ctx.saved_tensors
        desc, dev = ctx.desc, ctx.dev
        B = q32.shape[0]
        g32 = _f32(gX.reshape(B, -1), dev)
        gq = torch.empty((B, desc.dof), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.dcx_fkine_vjp(dev.index, C.byref(desc), _ptr(q32), _ptr(g32), B, _ptr(gq), _stream(dev)))
        return gq.to(device=ctx.in_device, dtype=ctx.in_dtype).reshape(ctx.in_shape), None


def fkine(desc: FkDesc, q: torch.Tensor) -> torch.Tensor:
    """control points [B, m, d] of configurations q [B, dof] (differentiable; HIP forward and vjp)."""
    return _FkineFn.apply(q, desc)


# ----------------------------------------------------------------------------- kernel matrix
def kernel_matrix(kind, p0, p1, x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """K[b, j] = K(x_b, s_j) for x [B, D], s [S, D]; result on x's device/dtype (not differentiable)."""
    lib = _lib.require_gpu()
    if x.requires_grad or s.requires_grad:
        raise _lib.DcxError("kernel callables are not differentiable on their own; differentiate through "
                            "DiffCo.score / poly_score / rbf_score (fused HIP gradient) instead")
    dev = _device(x.device if x.device.type == "cuda" else s.device)
    x32, s32 = _f32(x, dev), _f32(s, dev)
    B, D = x32.shape
    S = s32.shape[0]
    if s32.shape[1] != D:
        raise ValueError(f"kernel: feature widths differ ({D} vs {s32.shape[1]})")
    K = torch.empty((B, S), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.dcx_kernel_matrix(dev.index, kind, _kparams(p0, p1), _ptr(x32), B, _ptr(s32), S, D, _ptr(K),
                                         _stream(dev)))
    return K.to(device=x.device, dtype=x.dtype)


SOLVE_MAX_N = 3072   # dcx_solve takes up to DCX_SOLVE_MAX_N = 4096; beyond ~3000 unknowns the library LU is the faster one
SOLVE_MAX_RHS = 64


def _solve_device(a32: torch.Tensor, b32: torch.Tensor) -> torch.Tensor:
    """dcx_solve on device fp32 tensors a32 [n, n], b32 [n, r]: x [n, r] (device fp32).  Raises torch's LinAlgError on an
    exactly singular matrix, as torch.linalg.solve does."""
    lib = _lib.require_gpu()
    dev = a32.device
    n, r = b32.shape
    nbytes = int(lib.dcx_solve_work_bytes(n, r))
    work = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.float64)
    x = torch.empty((n, r), device=dev, dtype=torch.float32)
    info = torch.zeros(2, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        for flags in (0, 1):
            _lib.check(lib.dcx_solve(dev.index, _ptr(a32), _ptr(b32), n, r, _ptr(x), _ptr(work), nbytes, _ptr(info), flags,
                                     _stream(dev)))
            code = int(info[0].item())
            if code >= 0:
                break
            # a grid barrier gave up (another kernel held the CUs): the inputs are untouched, run again as one workgroup
    if code < 0:
        raise _lib.DcxError("dcx_solve: the solve did not complete")
    if code > 0:
        raise torch.linalg.LinAlgError(f"dcx_solve: the matrix is singular (pivot {code} is exactly zero)")
    return x


def solve(kmat: torch.Tensor, rhs: torch.Tensor) -> torch.Tensor:
    """x with kmat @ x = rhs, solved on the GPU: the S x S system of fit_poly (reference kernel_perceptrons.py:283,
    deprecated/MultiDiffCo.py:149).  float32 systems of up to 3072 unknowns: dcx_solve, one launch (LU with partial pivoting,
    fp64 inside, fp32 in and out: csrc/solve_kernels.hip); beyond, for more than 64 right-hand sides, and for FLOAT64 systems
    (dcx_solve reads fp32: a float64 `kmat + reg * eye` with a small `reg` would lose it on the way in - ADVICE r4) torch's
    hipSOLVER binding, a plain library factorisation in the caller's precision.  Result on kmat's device and dtype.  An exactly
    zero (or NaN) pivot raises torch.linalg.LinAlgError from dcx_solve, as LAPACK's info does; the library route returns what
    hipSOLVER returns.  Like every other op here it needs the GPU."""
    _lib.require_gpu()
    dev = _device(kmat.device)
    n = kmat.shape[-1]
    if kmat.dtype != torch.float64 and kmat.dim() == 2 and rhs.dim() in (1, 2) and 1 <= n <= SOLVE_MAX_N and rhs.shape[0] == n \
            and 1 <= rhs.numel() // n <= SOLVE_MAX_RHS:
        x = _solve_device(_f32(kmat, dev), _f32(rhs.reshape(n, -1), dev))
        return x.reshape(rhs.shape).to(device=kmat.device, dtype=kmat.dtype)
    x = torch.linalg.solve(kmat.detach().to(dev), rhs.detach().to(device=dev, dtype=kmat.dtype))
    return x.to(device=kmat.device)


def fit_nodes(kind, p0, p1, feats: torch.Tensor, targets: torch.Tensor, reg: float = 0.0) -> torch.Tensor:
    """fit_poly in one piece: nodes with (K(feats, feats) + reg I) nodes = targets, the kernel matrix built and solved on
    the device without leaving it (dcx_kernel_matrix + dcx_solve).  feats [S, ...], targets [S] or [S, C]; result on
    targets' device and dtype."""
    lib = _lib.require_gpu()
    dev = _device(feats.device)
    f = _f32(feats.reshape(len(feats), -1), dev)
    S, D = f.shape
    if S > SOLVE_MAX_N or targets.numel() // max(S, 1) > SOLVE_MAX_RHS:
        K = kernel_matrix(kind, p0, p1, f, f)
        if reg:
            K = K + reg * torch.eye(S, device=dev, dtype=K.dtype)
        return solve(K, _f32(targets, dev)).to(device=targets.device, dtype=targets.dtype)
    K = torch.empty((S, S), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.dcx_kernel_matrix(dev.index, kind, _kparams(p0, p1), _ptr(f), S, _ptr(f), S, D, _ptr(K), _stream(dev)))
    if reg:
        K.diagonal().add_(reg)
    x = _solve_device(K, _f32(targets.reshape(S, -1), dev))
    return x.reshape(targets.shape).to(device=targets.device, dtype=targets.dtype)


# ----------------------------------------------------------------------------- perceptron trainer
def train_perceptron_device(kind, p0, p1, beta, feats, y, gains, hypo, K, max_iteration):
    """The whole perceptron loop in one persistent launch (dcx_train_perceptron).  feats [N, D], y / gains / hypo
    [N] or [N, C], K [N, N] or None (cold start: allocated as zeros on the device).  Returns device tensors
    (gains, hypo, K) in fp32 with y's shape, and (iterations, converged)."""
    lib = _lib.require_gpu()
    dev = _device(feats.device)
    f = _f32(feats.reshape(len(feats), -1), dev)
    N, D = f.shape
    shape = tuple(y.shape)
    yy = _f32(y.reshape(N, -1), dev)
    Cn = yy.shape[1]
    def run(flags):
        g = _f32(gains.reshape(N, -1), dev).clone()
        h = _f32(hypo.reshape(N, -1), dev).clone()
        Kd = torch.zeros((N, N), device=dev, dtype=torch.float32) if K is None else _f32(K, dev).clone()
        info = torch.zeros(2, device=dev, dtype=torch.int32)
        with torch.cuda.device(dev):
            _lib.check(lib.dcx_train_perceptron_ex(dev.index, kind, _kparams(p0, p1), float(beta), _ptr(f), N, D, _ptr(yy), Cn,
                                                   _ptr(g), _ptr(h), _ptr(Kd), int(max_iteration), _ptr(info), flags, _stream(dev)))
        it, conv = (int(v) for v in info.tolist())
        return g, h, Kd, it, conv

    g, h, Kd, it, conv = run(0)
    if conv < 0:
        # a grid-wide barrier of the multi-workgroup trainer gave up (another kernel holding CUs for seconds): the run's
        # buffers are copies, so the whole training simply runs again on the one-workgroup kernels - asked for per call
        # (DCX_TRAIN_ONE_WORKGROUP), not through the process-wide knob other threads and the tests may have set
        g, h, Kd, it, conv = run(1)
        if conv < 0:
            raise _lib.DcxError("dcx_train_perceptron: the trainer did not complete")
    return g.reshape(shape), h.reshape(shape), Kd, it, bool(conv)


# ----------------------------------------------------------------------------- fused score model
class ScoreModel:
    """Owns one ``dcx_model`` (device copy of support rows + FK parameters).  `update()` refills it in place with new
    supports / weights (same transform, kernel and class count): no reallocation while they fit `capacity`."""

    @staticmethod
    def _rows(support_feat, weights, dev):
        sf = support_feat.detach().reshape(len(support_feat), -1).to(dtype=torch.float32).contiguous()
        w = weights.detach().to(dtype=torch.float32)
        w = (w.reshape(-1, 1) if w.ndim == 1 else w).contiguous()
        if len(sf) != len(w):
            raise ValueError(f"{len(sf)} supports but {len(w)} weight rows")
        # tensors already on the model's GPU are packed there by one kernel (dcx_model_create_ex); CPU tensors take the
        # library's host path; another GPU's tensors come over first
        if sf.is_cuda and sf.device != dev:
            sf = sf.to(dev)
        if w.is_cuda and w.device != dev:
            w = w.to(dev)
        if sf.is_cuda != w.is_cuda:
            sf, w = sf.to(dev), w.to(dev)
        return sf, w

    def __init__(self, desc, kind, p0, p1, support_feat, weights, device=None, capacity=0):
        lib = _lib.require_gpu()
        self.dev = _device(device)
        self.desc = desc if desc is not None else none_desc(int(support_feat.reshape(len(support_feat), -1).shape[1]))
        sf, w = self._rows(support_feat, weights, self.dev)
        self.S, self.C, self.dof, self.D = len(sf), int(w.shape[1]), self.desc.dof, self.desc.feature_dim
        if len(sf) and sf.shape[1] != self.D:
            raise ValueError(f"supports have {sf.shape[1]} features, the transform produces {self.D}")
        self.kernel = (int(kind), float(p0), float(p1))
        self.capacity = max(int(capacity), self.S)
        handle = C.c_void_p()
        with torch.cuda.device(self.dev):
            _lib.check(lib.dcx_model_create_ex(C.byref(handle), self.dev.index, C.byref(self.desc), kind,
                                               _kparams(p0, p1), _ptr(sf), _ptr(w), self.S, self.D, self.C,
                                               self.capacity, _stream(self.dev)))
        self._h = handle
        self._lib = lib
        self.leases = 0          # holders that need the rows to stay as they are (acquire / release)
        self.revision = 0        # bumped by update(): a backward pass that re-sweeps checks it still sees its forward's rows
        self._streams = set()    # raw handles of the streams that have launched this model since the last update()

    def acquire(self):
        """take a lease: while any is held, a FusedScorer builds a NEW model for changed state instead of refilling this one"""
        self.leases += 1
        return self

    def release(self):
        self.leases = max(0, self.leases - 1)

    def update(self, support_feat, weights):
        """new supports / weights into the same model (dcx_model_update): what train / fit_poly / update do to a checker's
        state every round of an active-learning loop.  The refill is enqueued on torch's CURRENT stream; launches of this
        model that went to other streams since the last update are waited for first (device-side, one event each), so a
        sweep still running elsewhere never reads half-packed rows."""
        sf, w = self._rows(support_feat, weights, self.dev)
        if int(w.shape[1]) != self.C or (len(sf) and sf.shape[1] != self.D):
            raise ValueError("update() keeps the model's feature width and class count")
        cur = torch.cuda.current_stream(self.dev)
        for h in self._streams:
            if h != cur.cuda_stream:
                cur.wait_stream(torch.cuda.ExternalStream(h, device=self.dev))
        self._streams.clear()
        with torch.cuda.device(self.dev):
            _lib.check(self._lib.dcx_model_update(self._h, _ptr(sf), _ptr(w), len(sf), _stream(self.dev)))
        self.S = len(sf)
        self.capacity = max(self.capacity, self.S)
        self.revision += 1
        return self

    def _st(self):
        """torch's current stream as the C ABI wants it, remembered for update()'s cross-stream wait"""
        st = _stream(self.dev)
        self._streams.add(st.value or 0)
        return st

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.dcx_model_destroy(h)
            except Exception:
                pass

    # raw entry points on fp32 device tensors -------------------------------------------
    def score_raw(self, q32):
        B = q32.shape[0]
        out = torch.empty((B, self.C), device=self.dev, dtype=torch.float32)
        with _on_device(self.dev):
            _lib.check(self._lib.dcx_score(self._h, _ptr(q32), B, _ptr(out), self._st()))
        return out

    def score_grad_raw(self, q32, upstream32=None, want_score=True):
        B = q32.shape[0]
        out = torch.empty((B, self.C), device=self.dev, dtype=torch.float32) if want_score else None
        grad = torch.empty((B, self.dof), device=self.dev, dtype=torch.float32)
        with _on_device(self.dev):
            _lib.check(self._lib.dcx_score_grad(self._h, _ptr(q32), B, _ptr(upstream32), _ptr(out), _ptr(grad),
                                                self._st()))
        return out, grad

    def score_jac_raw(self, q32):
        B = q32.shape[0]
        out = torch.empty((B, self.C), device=self.dev, dtype=torch.float32)
        jac = torch.empty((B, self.C, self.dof), device=self.dev, dtype=torch.float32)
        with _on_device(self.dev):
            _lib.check(self._lib.dcx_score_jac(self._h, _ptr(q32), B, _ptr(out), _ptr(jac), self._st()))
        return out, jac

    def score_hess_raw(self, q32, upstream32=None):
        """(gradient [B, dof], Hessian [B, dof, dof]) of sum_c upstream * score w.r.t. q — analytic second derivatives
        (dcx_score_hess; the reference double-backwards through dist_est, optim.py:380-391)"""
        B = q32.shape[0]
        grad = torch.empty((B, self.dof), device=self.dev, dtype=torch.float32)
        hess = torch.empty((B, self.dof, self.dof), device=self.dev, dtype=torch.float32)
        with _on_device(self.dev):
            _lib.check(self._lib.dcx_score_hess(self._h, _ptr(q32), B, _ptr(upstream32), _ptr(grad), _ptr(hess),
                                                self._st()))
        return grad, hess

    def margins(self, margin):
        """a safety margin as the C ABI wants it: C host floats (a scalar serves every class, like the reference's broadcast in
        `dist_est(p) - safety_margin`, optim.py:88-89)"""
        if torch.is_tensor(margin):
            margin = margin.detach().reshape(-1).tolist()
        elif not isinstance(margin, (list, tuple)):
            try:
                margin = [float(v) for v in margin.reshape(-1)]      # numpy
            except AttributeError:
                margin = [float(margin)]
        if len(margin) == 1:
            margin = list(margin) * self.C
        if len(margin) != self.C:
            raise ValueError(f"{len(margin)} safety margins for a score with {self.C} outputs")
        return (C.c_float * self.C)(*(float(v) for v in margin))

    def score_hinge_grad_raw(self, q32, margin, weight):
        """(score [B, C], d/dq of weight * sum_c clamp(score_c - margin_c, 0) [B, dof]) - the optimisers' collision term
        (optim.py:88-89); `margin`: a number or one per class.  One launch for one class, two (class scores, then the sweep
        whose upstream is the hinge's indicator) for several."""
        B = q32.shape[0]
        out = torch.empty((B, self.C), device=self.dev, dtype=torch.float32)
        grad = torch.empty((B, self.dof), device=self.dev, dtype=torch.float32)
        with _on_device(self.dev):
            _lib.check(self._lib.dcx_score_hinge_grad_mc(self._h, _ptr(q32), B, self.margins(margin), float(weight), _ptr(out),
                                                         _ptr(grad), self._st()))
        return out, grad

    # autograd-aware ------------------------------------------------------------------------
    def score(self, q: torch.Tensor) -> torch.Tensor:
        """[B, C] scores for q [B, dof]; differentiable w.r.t. q (gradient from the fused HIP pass)."""
        if not (q.requires_grad and torch.is_grad_enabled()):
            # nothing to differentiate: the score-only sweep, no autograd.Function around it (its apply() alone cost 10 us of
            # a 17 us call: profiles/r05_api_latency.txt)
            s = self.score_raw(_f32(q.reshape(-1, self.dof), self.dev))
            return s if (s.device == q.device and s.dtype == q.dtype) else s.to(device=q.device, dtype=q.dtype)
        return _ScoreFn.apply(q, self)

    def score_and_grad(self, q: torch.Tensor, upstream: torch.Tensor = None):
        """(score [B, C], d(sum_c upstream*score)/dq [B, dof]) in ONE launch, no autograd graph."""
        q32 = _f32(q.reshape(-1, self.dof), self.dev)
        up = None if upstream is None else _f32(upstream.reshape(-1, self.C), self.dev)
        s, g = self.score_grad_raw(q32, up)
        return s.to(device=q.device, dtype=q.dtype), g.to(device=q.device, dtype=q.dtype)


def _is_batched(t):
    """True inside torch.vmap (e.g. torch.autograd.functional.jacobian(vectorize=True) batching the backward)"""
    ft = getattr(torch._C, "_functorch", None)
    for name in ("is_batchedtensor", "is_legacy_batchedtensor"):  # torch.vmap / the legacy vmap autograd still uses
        f = getattr(ft, name, None)
        if f is not None and f(t):
            return True
    return False


class _ScoreFn(torch.autograd.Function):
    """score = model(q), differentiable w.r.t. q.

    C == 1: when q needs a gradient the forward launch is the fused score+gradient pass and backward is one
    elementwise multiply.  C > 1: forward is the score-only sweep; backward runs ONE sweep with the actual upstream
    (`dcx_score_grad`) — unless it is being vmapped (`torch.autograd.functional.jacobian(vectorize=True)`, the
    reference's optim.py:211-216, batches the backward over one-hot upstreams): then the full Jacobian is formed once
    (`dcx_score_jac`, one sweep per class) and contracted with torch ops, which vmap can batch."""

    @staticmethod
    def forward(ctx, q, model):
        q32 = _f32(q.reshape(-1, model.dof), model.dev)
        ctx.model, ctx.in_shape, ctx.in_dtype, ctx.in_device = model, q.shape, q.dtype, q.device
        if ctx.needs_input_grad[0] and model.C == 1:
            s, jac = model.score_jac_raw(q32)
            ctx.save_for_backward(jac.reshape(-1, model.dof).to(device=q.device, dtype=q.dtype))   # d score / d q  [B, dof]
        else:
            s = model.score_raw(q32)
            ctx.save_for_backward(q32)
            ctx.revision = model.revision   # (C > 1: backward sweeps again - against the rows this forward saw, or not at all)
        return s.to(device=q.device, dtype=q.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gs):
        (saved,) = ctx.saved_tensors
        model = ctx.model
        if model.C == 1:  # saved = d score / d q  [B, dof]; gs [B, 1]: ONE elementwise launch
            return (gs * saved).reshape(ctx.in_shape), None
        q32 = saved
        if model.revision != ctx.revision:
            # train / fit_poly / update refilled the model's rows in place between this forward and its backward:
            # the gradient would be the NEW model's
            raise RuntimeError("the checker's supports / weights changed between the forward pass of a multi-class score and its "
                               "backward pass: run backward before retraining (or score again)")
        if _is_batched(gs):
            _, jac = model.score_jac_raw(q32)  # [B, C, dof]
            jac = jac.to(device=ctx.in_device, dtype=ctx.in_dtype)
            return (gs.unsqueeze(-1) * jac).sum(dim=-2).reshape(ctx.in_shape), None
        _, g = model.score_grad_raw(q32, _f32(gs.reshape(-1, model.C), model.dev), want_score=False)
        return g.to(device=ctx.in_device, dtype=ctx.in_dtype).reshape(ctx.in_shape), None
