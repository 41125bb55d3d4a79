"""Robot models with the reference's interface: `.fkine(q, reuse=False) -> [N, m, d]`, `.limits`,
`.dof`, `.wrap(q)`.

Drop-in for diffco/model.py (and its stale twin diffco/robot_fkine.py): RevolutePlanarRobot 23-76,
RigidPlanarBody 78-116, RigidBody 118-171, BaxterLeftArmFK 188-244, BaxterRightArmFK 246-308,
BaxterDualArmFK 310-386, PandaFK 390-453 (robot_fkine.py:388-444 for the 5-point variant),
DualPandaFK 456-502, PointRobot1D 505-.  The numbers below (link lengths, DH tables, joint limits)
restate the reference's robot definitions; the FK itself runs in HIP (`dcx_fkine`, differentiable
through `dcx_fkine_vjp`), and when a robot's `fkine` is handed to DiffCo as `transform` the chain is
fused into the score kernel.  Ground-truth geometry (`update_polygons`, FCL) is out of scope.
"""
import math

import numpy as np
import torch

from . import _fkdesc as fd
from . import _ops
from .utils import wrap2pi

pi = math.pi


class Model:
    dof = None
    limits = None
    _desc = None

    def fk_desc(self):
        """plain-data description of this robot's transform (ctypes dcx_fk_desc)"""
        return self._desc

    def fkine(self, q, reuse=False):
        if reuse:
            return self.fkine_backup
        self.fkine_backup = _ops.fkine(self._desc, torch.reshape(q, (-1, self.dof)))
        return self.fkine_backup

    fkine_backup = None

    def polygons(self, q):
        raise NotImplementedError

    def update_polygons(self, q):
        raise NotImplementedError("ground-truth geometry (FCL) is outside diffco_amd's scope")

    def wrap(self, q):
        return wrap2pi(q)


class RevolutePlanarRobot(Model):
    """Planar serial arm of revolute joints: q_i are relative angles, the control points are the link ends (D = 2 dof).
    Signature and attributes as in the reference (model.py:23-38): `link_length` a number (with `dof`) or a list,
    `limits` one [lo, hi] pair for every joint or one pair per joint (default [-pi, pi])."""

    def __init__(self, link_length, link_width, dof=None, limits=None):
        scalar_length = isinstance(link_length, (int, float))
        n = int(dof) if dof is not None else len(link_length)
        lengths = [float(link_length)] * n if scalar_length else [float(v) for v in link_length]
        if limits is None:
            limits = (-np.pi, np.pi)
        one_pair = len(limits) == 2 and all(isinstance(v, (int, float)) for v in limits)
        bounds = [list(limits) for _ in range(n)] if one_pair else [list(row) for row in limits]
        if len(lengths) != n or len(bounds) != n:
            raise AssertionError(f"planar arm: {len(lengths)} link lengths and {len(bounds)} limit rows for dof {n}")
        self.dof = n
        self.link_width = link_width
        self.link_length = torch.FloatTensor(lengths)
        self.limits = torch.FloatTensor(bounds)
        self.collision_objs = None
        self._desc = fd.planar_desc(lengths)


class RigidPlanarBody(Model):
    """q = (x, y, theta); parts = [(type, (kx, ky), (w, h)), ...] — keypoints are the part centres"""

    def __init__(self, parts, limits=None):
        self.parts = parts
        self.dof = 3
        self.limits = torch.FloatTensor(limits) if limits is not None else torch.FloatTensor(
            [[-10, 10], [-10, 10], [-pi, pi]])
        self.keypoints = torch.FloatTensor([p[1] for p in parts]).T  # 2 x M, as in the reference
        self.collision_objs = None
        self._desc = fd.keypoint_desc(self.keypoints.T.numpy(), 2)

    def wrap(self, q):
        return torch.cat((q[..., :2], wrap2pi(q[..., 2:])), dim=-1)


class RigidBody(Model):
    """q = (x, y, z, roll, pitch, yaw), R = Rz(yaw) Ry(pitch) Rx(roll).  Unlike the reference (which
    loads a mesh with trimesh to take its bounding-box corners) the keypoints [3, M] or [M, 3] are
    given directly; `body_path` is kept for signature compatibility and ignored."""

    def __init__(self, body_path=None, keypoints=None, limits=None, transform=None, center=True):
        if keypoints is None:
            raise ValueError("diffco_amd.model.RigidBody needs explicit keypoints (mesh loading is out of scope)")
        self.body_path = body_path
        self.dof = 6
        self.limits = torch.FloatTensor(limits) if limits is not None else torch.FloatTensor(
            [[-10, 10], [-10, 10], [-10, 10], [-pi, pi], [-pi, pi], [-pi, pi]])
        kp = torch.as_tensor(keypoints, dtype=torch.float32)
        if kp.shape[0] != 3:
            kp = kp.T
        self.keypoints = kp  # 3 x M
        self.collision_objs = []
        self._desc = fd.keypoint_desc(self.keypoints.T.numpy(), 3)

    def wrap(self, q):
        return torch.cat((q[..., :3], wrap2pi(q[..., 3:])), dim=-1)


class DHParameters:
    """Denavit-Hartenberg table of one chain as fp32 tensors; sin / cos of alpha are taken in fp32 from the fp32 alpha, as
    the reference does (model.py:172-180), so the constants the kernels see are the reference's bit for bit"""

    def __init__(self, a=0, alpha=0, d=0, theta=0):
        for name, values in (("a", a), ("alpha", alpha), ("d", d), ("theta", theta)):
            setattr(self, name, torch.tensor(values, dtype=torch.float32).reshape(-1))
        self.s_alpha, self.c_alpha = torch.sin(self.alpha), torch.cos(self.alpha)

    def chain(self, joint_q, base=None):
        ch = dict(a=self.a.tolist(), d=self.d.tolist(), alpha=self.alpha.tolist(), theta0=self.theta.tolist(),
                  joint_q=list(joint_q))
        if base is not None:
            ch["base"] = base
        return ch


_BAXTER_LIMITS = [[-1.70167993878, 1.70167993878], [-2.147, 1.047], [-3.05417993878, 3.05417993878],
                  [-0.05, 2.618], [-3.059, 3.059], [-1.57079632679, 2.094], [-3.059, 3.059]]
_BAXTER_L = [270.35, 69, 364.35, 69, 374.29, 10, 387.35]  # mm; L6 = wrist-pitch centre to tool tip


def _baxter_dh():
    L = torch.FloatTensor(_BAXTER_L) / 1000
    return L, DHParameters(
        a=[L[1], 0, L[3], 0, L[5], 0, 0],
        alpha=[-pi / 2, pi / 2, -pi / 2, pi / 2, -pi / 2, pi / 2, 0],
        d=[L[0], 0, L[2], 0, L[4], 0, L[6]],
        theta=[0, pi / 2, 0, 0, 0, 0, 0])


class BaxterLeftArmFK(Model):
    """7-DoF DH chain; control points = origins of frames 0, 2, 4, 6 (D = 12)"""

    def __init__(self):
        self.limits = torch.FloatTensor(_BAXTER_LIMITS)
        self.L, self.dhparams = _baxter_dh()
        self.c_alpha, self.s_alpha = self.dhparams.c_alpha, self.dhparams.s_alpha
        self.dof = 7
        self.fk_mask = [True, False, True, False, True, False, True]
        self._desc = fd.dh_desc(7, [self.dhparams.chain(range(7))],
                                [(0, i, (0, 0, 0)) for i, m in enumerate(self.fk_mask) if m])


class BaxterRightArmFK(BaxterLeftArmFK):
    """same numbers as the left arm in the reference (model.py:246-281)"""


BaxterFK = BaxterLeftArmFK


class BaxterDualArmFK(Model):
    """q = (left 7, right 7); points interleaved per frame: L0, R0, L2, R2, L4, R4, L6, R6 (D = 24)"""

    def __init__(self):
        self.limits = torch.FloatTensor(_BAXTER_LIMITS).repeat(2, 1)
        self.L, self.left_dhparams = _baxter_dh()
        _, self.right_dhparams = _baxter_dh()
        off = torch.FloatTensor([278, 64, 1104]) / 1000
        left_base = fd.rotz_base(-pi / 4, (float(off[0]), -float(off[1]), float(off[2])))
        right_base = fd.rotz_base(-3 * pi / 4, (-float(off[0]), -float(off[1]), float(off[2])))
        # the reference builds the bases from torch fp32 sin/cos (utils.rotz); do the same
        for base, ang in ((left_base, -pi / 4), (right_base, -3 * pi / 4)):
            c, s = float(torch.cos(torch.tensor([ang]))[0]), float(torch.sin(torch.tensor([ang]))[0])
            base[0], base[1], base[4], base[5] = c, -s, s, c
        self.dof = 14
        self.fk_mask = [True, False, True, False, True, False, True]
        frames = [i for i, m in enumerate(self.fk_mask) if m]
        pts = []
        for f in frames:
            pts += [(0, f, (0, 0, 0)), (1, f, (0, 0, 0))]
        self._desc = fd.dh_desc(14, [self.left_dhparams.chain(range(0, 7), left_base),
                                     self.right_dhparams.chain(range(7, 14), right_base)], pts)


_PANDA_LIMITS = [[-2.8973, 2.8973], [-1.7628, 1.7628], [-2.8973, 2.8973], [-3.0718, -0.0698],
                 [-2.8973, 2.8973], [-0.0175, 3.7525], [-2.8973, 2.8973]]


def _panda_dh():
    L = torch.FloatTensor([0.3330, 0.3160, 0.0825, 0.3840, 0.0880, 0.1070 * 2])
    return L, DHParameters(
        a=[0, 0, L[2], -L[2], 0, L[4], 0],
        alpha=[-pi / 2, pi / 2, pi / 2, -pi / 2, pi / 2, pi / 2, 0],
        d=[L[0], 0, L[1], 0, L[3], 0, L[5]],
        theta=[0, 0, 0, 0, 0, 0, 0])


def _panda_points(chain, fk_mask, d_last, fingers):
    pts = [(chain, i, (0, 0, 0)) for i, m in enumerate(fk_mask) if m]
    if fingers:  # two finger points in the last frame at y = +-d7/2 (model.py:445-450)
        pts += [(chain, 6, (0, 0.5 * d_last, 0)), (chain, 6, (0, -0.5 * d_last, 0))]
    return pts


class PandaFK(Model):
    """Franka Panda, 7-DoF DH chain.  fingers=True (default, diffco/model.py): 5 frame origins + 2
    finger points (D = 21); fingers=False reproduces diffco/robot_fkine.py's 5-point variant (D = 15)."""

    def __init__(self, fingers=True):
        self.limits = torch.FloatTensor(_PANDA_LIMITS)
        self.L, self.dhparams = _panda_dh()
        self.c_alpha, self.s_alpha = self.dhparams.c_alpha, self.dhparams.s_alpha
        self.dof = 7
        self.fk_mask = [True, False, True, True, True, False, True]
        self._desc = fd.dh_desc(7, [self.dhparams.chain(range(7))],
                                _panda_points(0, self.fk_mask, float(self.dhparams.d[-1]), fingers))


class DualPandaFK(Model):
    """q interleaves the arms: odd columns -> left arm (base y = +0.84), even -> right; output is
    the left arm's 7 points followed by the right arm's (D = 42)"""

    def __init__(self):
        self.limits = torch.FloatTensor([lim for lim in _PANDA_LIMITS for _ in range(2)])
        self.left_panda, self.right_panda = PandaFK(), PandaFK()
        self.bases = torch.FloatTensor([[0.0, 0.84, 0.0], [0.0, 0.0, 0.0]])
        self.dhparams = self.left_panda.dhparams
        self.dof = 14
        mask, d7 = self.left_panda.fk_mask, float(self.dhparams.d[-1])
        lb = list(fd.IDENTITY_BASE)
        lb[7] = 0.84
        self._desc = fd.dh_desc(14, [self.dhparams.chain(range(1, 14, 2), lb),
                                     self.dhparams.chain(range(0, 14, 2), list(fd.IDENTITY_BASE))],
                                _panda_points(0, mask, d7, True) + _panda_points(1, mask, d7, True))


class PointRobot1D(Model):
    """q in [0, 1]^dof mapped affinely onto limits[:-1] (last limits row is time); pure host arithmetic"""

    def __init__(self, limits):
        self.limits = torch.FloatTensor(limits)
        self.dof = 1

    def fkine(self, q, reuse=False):
        q = torch.reshape(q, (-1, self.dof))
        return q * (self.limits[:-1, 1] - self.limits[:-1, 0]) + self.limits[:-1, 0]

    def normalize(self, q):
        return (q - self.limits[:, 0]) / (self.limits[:, 1] - self.limits[:, 0])
