"""ctypes mirror of ``dcx_fk_desc`` (include/dcx.h) and builders for the reference's robots.

Host logic only: plain-data descriptions of `transform(q) -> control points`; no compute.
Parameter values restate the reference's robot definitions (paths under /root/reference/diffco):
Baxter model.py:193-222, 250-281, 312-363; Panda model.py:394-427 (7 points) and
robot_fkine.py:392-425 (5 points); DualPanda model.py:456-484; planar model.py:23-38;
SE(2)/SE(3) bodies model.py:78-88, 118-153.
"""
import ctypes as C
import math

import numpy as np
import torch

DCX_FK_NONE, DCX_FK_PLANAR, DCX_FK_DH, DCX_FK_SE2, DCX_FK_SE3, DCX_FK_TREE = range(6)
DCX_J_FIXED, DCX_J_REV_X, DCX_J_REV_Y, DCX_J_REV_Z, DCX_J_PRISMATIC = range(5)
DCX_K_RQ, DCX_K_POLY, DCX_K_MQ = range(3)
MAX_JOINTS, MAX_CHAINS, MAX_POINTS, MAX_DOF, MAX_D, MAX_C = 16, 2, 32, 32, 96, 8
MAX_TREE_CHAINS, MAX_TREE_JOINTS, MAX_TREE_BASES = 16, 64, 4


class FkDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dof", C.c_int32), ("n_points", C.c_int32), ("point_dim", C.c_int32),
        ("link_length", C.c_float * MAX_DOF),
        ("n_chains", C.c_int32), ("chain_len", C.c_int32 * MAX_CHAINS),
        ("joint_q", (C.c_int32 * MAX_JOINTS) * MAX_CHAINS),
        ("a", (C.c_float * MAX_JOINTS) * MAX_CHAINS),
        ("d", (C.c_float * MAX_JOINTS) * MAX_CHAINS),
        ("sin_alpha", (C.c_float * MAX_JOINTS) * MAX_CHAINS),
        ("cos_alpha", (C.c_float * MAX_JOINTS) * MAX_CHAINS),
        ("theta0", (C.c_float * MAX_JOINTS) * MAX_CHAINS),
        ("base", (C.c_float * 12) * MAX_CHAINS),
        ("pt_chain", C.c_int32 * MAX_POINTS), ("pt_frame", C.c_int32 * MAX_POINTS),
        ("pt_off", (C.c_float * 3) * MAX_POINTS),
        ("keypoints", (C.c_float * 3) * MAX_POINTS),
        ("t_n_chains", C.c_int32), ("t_coord_major", C.c_int32),
        ("t_chain_len", C.c_int32 * MAX_TREE_CHAINS),
        ("t_base", (C.c_float * 12) * MAX_TREE_CHAINS),
        ("t_type", C.c_int32 * MAX_TREE_JOINTS), ("t_q", C.c_int32 * MAX_TREE_JOINTS),
        ("t_scale", C.c_float * MAX_TREE_JOINTS), ("t_offset", C.c_float * MAX_TREE_JOINTS),
        ("t_fixed", (C.c_float * 12) * MAX_TREE_JOINTS),
        ("t_axis", (C.c_float * 3) * MAX_TREE_JOINTS),
    ]

    @property
    def feature_dim(self):
        return self.n_points * self.point_dim

    @property
    def feature_shape(self):
        """shape of one configuration's feature block as the reference lays it out"""
        if self.kind == DCX_FK_TREE and self.t_coord_major:
            return (self.point_dim, self.n_points)
        return (self.n_points, self.point_dim)

    def key(self):
        return bytes(self)


def none_desc(dof):
    d = FkDesc()
    d.kind, d.dof, d.n_points, d.point_dim = DCX_FK_NONE, dof, dof, 1
    return d


def planar_desc(link_length):
    ll = [float(x) for x in link_length]
    if len(ll) > MAX_DOF:
        raise ValueError(f"planar arm: dof {len(ll)} > {MAX_DOF}")
    d = FkDesc()
    d.kind, d.dof, d.n_points, d.point_dim = DCX_FK_PLANAR, len(ll), len(ll), 2
    for i, v in enumerate(ll):
        d.link_length[i] = v
    return d


def keypoint_desc(keypoints, dim):
    """SE(2) (dim=2, q=(x,y,theta)) / SE(3) (dim=3, q=(x,y,z,roll,pitch,yaw)); keypoints [M, dim]."""
    kp = np.asarray(keypoints, dtype=np.float32).reshape(-1, dim)
    if len(kp) > MAX_POINTS:
        raise ValueError(f"rigid body: {len(kp)} keypoints > {MAX_POINTS}")
    d = FkDesc()
    d.kind = DCX_FK_SE2 if dim == 2 else DCX_FK_SE3
    d.dof, d.n_points, d.point_dim = (3 if dim == 2 else 6), len(kp), dim
    for k in range(len(kp)):
        for j in range(dim):
            d.keypoints[k][j] = float(kp[k, j])
    return d


IDENTITY_BASE = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]


def dh_desc(dof, chains, points):
    """chains: list of dict(a, d, alpha, theta0, joint_q, base[12]); points: list of (chain, frame, (ox,oy,oz))."""
    if len(chains) > MAX_CHAINS or len(points) > MAX_POINTS or dof > MAX_DOF:
        raise ValueError("DH description exceeds the compiled limits")
    d = FkDesc()
    d.kind, d.dof, d.n_points, d.point_dim = DCX_FK_DH, dof, len(points), 3
    d.n_chains = len(chains)
    for c, ch in enumerate(chains):
        n = len(ch["a"])
        if n > MAX_JOINTS:
            raise ValueError("chain too long")
        d.chain_len[c] = n
        # sin/cos of alpha in fp32 from the fp32 alpha, as DHParameters does (model.py:179-180)
        alpha = torch.tensor(ch["alpha"], dtype=torch.float32)
        sa, ca = alpha.sin(), alpha.cos()
        a32 = torch.tensor(ch["a"], dtype=torch.float32)
        d32 = torch.tensor(ch["d"], dtype=torch.float32)
        t32 = torch.tensor(ch["theta0"], dtype=torch.float32)
        for i in range(n):
            d.joint_q[c][i] = int(ch["joint_q"][i])
            d.a[c][i], d.d[c][i] = float(a32[i]), float(d32[i])
            d.sin_alpha[c][i], d.cos_alpha[c][i] = float(sa[i]), float(ca[i])
            d.theta0[c][i] = float(t32[i])
        for e, v in enumerate(ch.get("base", IDENTITY_BASE)):
            d.base[c][e] = float(v)
    for k, (c, f, off) in enumerate(points):
        d.pt_chain[k], d.pt_frame[k] = int(c), int(f)
        for j in range(3):
            d.pt_off[k][j] = float(off[j])
    return d


def rotz_base(angle, t):
    c, s = math.cos(angle), math.sin(angle)
    # fp32 rounding of the entries happens when they are stored in the struct
    return [c, -s, 0, t[0], s, c, 0, t[1], 0, 0, 1, t[2]]


def tree_desc(dof, chains, points, coord_major=True):
    """DCX_FK_TREE description.

    chains: list of dict(base=[12 floats] (optional), joints=[dict(type=DCX_J_*, q=int, scale=float, offset=float,
            fixed=[12 floats, row-major 3x4], axis=(x, y, z))]) — one entry per root-to-leaf path;
    points: list of (chain, frame, (ox, oy, oz)) in feature order."""
    n_j = sum(len(ch["joints"]) for ch in chains)
    n_bases = len({tuple(float(v) for v in ch.get("base", IDENTITY_BASE)) for ch in chains})
    if n_bases > MAX_TREE_BASES:
        raise ValueError(f"kinematic tree: {n_bases} distinct base transforms (max {MAX_TREE_BASES})")
    if (len(chains) > MAX_TREE_CHAINS or n_j > MAX_TREE_JOINTS or len(points) > MAX_POINTS or dof > MAX_DOF
            or 3 * len(points) > MAX_D):
        raise ValueError(
            f"kinematic tree exceeds the compiled limits: {len(chains)} chains (max {MAX_TREE_CHAINS}), {n_j} joints "
            f"over all root-to-leaf paths (max {MAX_TREE_JOINTS}), {len(points)} control points (max {MAX_POINTS}), "
            f"dof {dof} (max {MAX_DOF})")
    d = FkDesc()
    d.kind, d.dof, d.n_points, d.point_dim = DCX_FK_TREE, dof, len(points), 3
    d.t_n_chains, d.t_coord_major = len(chains), int(bool(coord_major))
    j = 0
    for c, ch in enumerate(chains):
        d.t_chain_len[c] = len(ch["joints"])
        for e, v in enumerate(ch.get("base", IDENTITY_BASE)):
            d.t_base[c][e] = float(v)
        for jt in ch["joints"]:
            d.t_type[j] = int(jt["type"])
            movable = jt["type"] != DCX_J_FIXED
            if movable and not 0 <= int(jt["q"]) < dof:
                raise ValueError("joint reads a configuration entry outside [0, dof)")
            d.t_q[j] = int(jt["q"]) if movable else 0
            d.t_scale[j] = float(jt.get("scale", 1.0)) if movable else 0.0
            d.t_offset[j] = float(jt.get("offset", 0.0)) if movable else 0.0
            for e, v in enumerate(jt.get("fixed", IDENTITY_BASE)):
                d.t_fixed[j][e] = float(v)
            for e, v in enumerate(jt.get("axis", (0.0, 0.0, 0.0))):
                d.t_axis[j][e] = float(v)
            j += 1
    for k, (c, f, off) in enumerate(points):
        if not (0 <= c < len(chains) and 0 <= f < len(chains[c]["joints"])):
            raise ValueError("control point attached to a frame that does not exist")
        d.pt_chain[k], d.pt_frame[k] = int(c), int(f)
        for e in range(3):
            d.pt_off[k][e] = float(off[e])
    return d
