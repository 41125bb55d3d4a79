"""Host-side helpers the reference keeps in diffco/utils.py (behaviour restated, torch on the
caller's device; none of this is on the score/grad hot path).  Reference: utils.py:4-13 (rotz),
40-48 (rot_2d), 51-52 (wrap2pi), 54-55 (se2_wrap2pi), 60-64 (anglin), 79-85 (make_continue),
87-101 (dense_path).  DH2mat / euler2mat live on the device (csrc/fk_device.h)."""
import math

import torch


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
