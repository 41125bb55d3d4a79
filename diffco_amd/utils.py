"""Host-side helpers the reference keeps in diffco/utils.py (behaviour restated, torch on the
caller's device; none of this is on the score/grad hot path).  Reference: utils.py:4-13 (rotz),
40-48 (rot_2d), 51-52 (wrap2pi), 54-55 (se2_wrap2pi), 60-64 (anglin), 79-85 (make_continue),
87-101 (dense_path).  DH2mat (utils.py:66-75) and euler2mat (15-38) - the two link-transform helpers robot classes written
against the reference build their FK from (model.py:230, 437) - are HIP kernels behind the same call signatures
(`dcx_dh_frames`, `dcx_euler_frames`), differentiable with respect to the angles; like every op of this package they need the
GPU (CPU tensors go there and come back)."""
import ctypes as C
import math

import torch
from torch.autograd.function import once_differentiable


class _DHFrames(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, a, d, s_alpha, c_alpha):
        from . import _lib, _ops
        lib = _lib.require_gpu()
        dev = _ops._device(q.device)
        B, dof = q.shape
        q32 = _ops._f32(q, dev)
        par = [_ops._f32(torch.as_tensor(t).reshape(-1).expand(dof), dev) for t in (a, d, s_alpha, c_alpha)]
        T = torch.empty((B, dof, 4, 4), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.dcx_dh_frames(dev.index, _ops._ptr(q32), B, dof, *(_ops._ptr(t) for t in par), _ops._ptr(T),
                                         _ops._stream(dev)))
        ctx.dev, ctx.in_dtype, ctx.in_device = dev, q.dtype, q.device
        ctx.save_for_backward(q32, par[0], par[2], par[3])
        return T.to(device=q.device, dtype=q.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gT):
        from . import _lib, _ops
        lib = _lib.require_gpu()
        q32, a, sa, ca = ctx.saved_tensors
        dev = ctx.dev
        B, dof = q32.shape
        g32 = _ops._f32(gT, dev)
        gq = torch.empty((B, dof), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.dcx_dh_frames_vjp(dev.index, _ops._ptr(q32), B, dof, _ops._ptr(a), _ops._ptr(sa), _ops._ptr(ca),
                                             _ops._ptr(g32), _ops._ptr(gq), _ops._stream(dev)))
        return gq.to(device=ctx.in_device, dtype=ctx.in_dtype), None, None, None, None


def DH2mat(q, a, d, s_alpha, c_alpha):
    """standard DH link transforms T[b, i] = Rz(q[b, i]) Tz(d_i) Tx(a_i) Rx(alpha_i) -> [B, dof, 4, 4] (reference utils.py:66-75),
    computed by `dcx_dh_frames`; differentiable with respect to q (the DH parameters are constants, as in the reference's models)"""
    if q.ndim != 2:
        raise ValueError("DH2mat: q is [B, dof]")
    return _DHFrames.apply(q, a, d, s_alpha, c_alpha)


class _EulerFrames(torch.autograd.Function):
    @staticmethod
    def forward(ctx, phi):
        from . import _lib, _ops
        lib = _lib.require_gpu()
        dev = _ops._device(phi.device)
        p32 = _ops._f32(phi, dev)
        B = p32.shape[0]
        R = torch.empty((B, 3, 3), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.dcx_euler_frames(dev.index, _ops._ptr(p32), B, _ops._ptr(R), _ops._stream(dev)))
        ctx.dev, ctx.in_dtype, ctx.in_device = dev, phi.dtype, phi.device
        ctx.save_for_backward(p32)
        return R.to(device=phi.device, dtype=phi.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gR):
        from . import _lib, _ops
        lib = _lib.require_gpu()
        (p32,) = ctx.saved_tensors
        dev = ctx.dev
        B = p32.shape[0]
        g32 = _ops._f32(gR, dev)
        gp = torch.empty((B, 3), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.dcx_euler_frames_vjp(dev.index, _ops._ptr(p32), _ops._ptr(g32), B, _ops._ptr(gp), _ops._stream(dev)))
        return gp.to(device=ctx.in_device, dtype=ctx.in_dtype)


def euler2mat(phi):
    """rotation matrices Rz(yaw) Ry(pitch) Rx(roll) of (roll, pitch, yaw) rows -> [N, 3, 3] (reference utils.py:15-38), computed
    by `dcx_euler_frames`; differentiable with respect to phi"""
    return _EulerFrames.apply(phi.reshape((-1, 3)))


def wrap2pi(theta):
    """angle(s) -> [-pi, pi)"""
    return (math.pi + theta) % (2 * math.pi) - math.pi


def se2_wrap2pi(x):
    return torch.cat([x[..., :2], wrap2pi(x[..., 2:3])], dim=-1)


def rotz(phi):
    phi = torch.as_tensor(phi, dtype=torch.float32).reshape(-1)
    c, s = torch.cos(phi), torch.sin(phi)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack([c, -s, z, s, c, z, z, z, o], dim=1).reshape(-1, 3, 3)


def rot_2d(phi):
    phi = torch.as_tensor(phi, dtype=torch.float32).reshape(-1)
    c, s = torch.cos(phi), torch.sin(phi)
    return torch.stack([c, -s, s, c], dim=1).reshape(-1, 2, 2)


def anglin(q1, q2, num=50, endpoint=True):
    """linspace between two angle vectors along the shorter arc, wrapped to [-pi, pi)"""
    q1 = torch.as_tensor(q1, dtype=torch.float32)
    q2 = torch.as_tensor(q2, dtype=torch.float32)
    delta = wrap2pi(q2 - q1)
    n = num - 1 if endpoint else num
    t = torch.arange(num, dtype=torch.float64).reshape(-1, *([1] * q1.ndim)) / max(n, 1)
    return wrap2pi(q1 + t * delta.double())


def make_continue(q, max_gap=math.pi):
    """undo +-2pi jumps between consecutive rows (for plotting angular paths)"""
    q = torch.as_tensor(q, dtype=torch.float32)
    jump = torch.zeros_like(q)
    step = q[1:] - q[:-1]
    jump[1:] = (step.abs() > max_gap) * torch.sign(step)
    return q - torch.cumsum(jump, dim=0) * 2 * math.pi


def dense_path(q, max_step=2.0, max_step_num=None):
    """Insert ceil(|segment| / max_step) equally spaced points per segment, keep both endpoints.
    With max_step_num the step is enlarged so that about that many steps cover the whole path."""
    if max_step_num is not None:
        alt = torch.norm(q[1:] - q[:-1], dim=-1).sum().item() / max_step_num
        max_step = max(max_step, alt)
    pieces = []
    for i in range(len(q) - 1):
        seg = q[i + 1] - q[i]
        length = seg.norm()
        n = int(torch.ceil(length / max_step).item())
        idx = torch.arange(n, device=q.device, dtype=q.dtype).reshape(-1, 1)
        pieces.append(q[i] + idx * (seg * (max_step / length)))
    pieces.append(q[-1:])
    out = torch.cat(pieces)
    assert torch.all(out[0] == q[0]) and torch.all(out[-1] == q[-1])
    return out


def dense_path_indexed(q, max_step=2.0):
    """`dense_path(q, max_step)` plus, for every dense point, the segment it lies on and its step index:
    dense[n] = q[seg[n]] + step[n] * max_step * unit(q[seg[n]+1] - q[seg[n]]); the final point q[-1] is reported
    with seg = len(q) - 1, step = 0.  Used to build the constraint Jacobian analytically."""
    pieces, segs, steps = [], [], []
    for i in range(len(q) - 1):
        seg = q[i + 1] - q[i]
        length = seg.norm()
        n = int(torch.ceil(length / max_step).item())
        idx = torch.arange(n, device=q.device, dtype=q.dtype).reshape(-1, 1)
        pieces.append(q[i] + idx * (seg * (max_step / length)))
        segs += [i] * n
        steps += list(range(n))
    pieces.append(q[-1:])
    segs.append(len(q) - 1)
    steps.append(0)
    return torch.cat(pieces), torch.tensor(segs, dtype=torch.long), torch.tensor(steps, dtype=q.dtype)
