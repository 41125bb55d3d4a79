#!/usr/bin/env python3
"""Generate golden input/output vectors for the score(+grad) hot path from the REFERENCE.

Runs ONLY in the build container (needs /root/reference).  It imports the reference's
Python modules with the stub recipe of SURVEY.md §8c (fcl/trimesh stubbed, diffco/__init__
bypassed), evaluates the reference functions on seeded inputs and writes small .npz / .json
fixtures under tests/golden/.  No reference source or bytecode is copied: fixtures hold
only inputs and expected outputs.

Every fixture also carries an fp64 "referee" evaluation (reference FK run in float64 +
direct-difference kernels in float64) because the reference's own fp32 path (torch.cdist
GEMM form) is ~1e-5 away from the true value.

Usage:  python tools/make_golden.py [--out tests/golden]
"""
import argparse
import importlib
import json
import math
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"


def import_reference():
    for n in ("fcl", "trimesh"):
        sys.modules.setdefault(n, types.ModuleType(n))
    pkg = types.ModuleType("diffco")
    pkg.__path__ = [f"{REF}/diffco"]
    sys.modules["diffco"] = pkg
    mods = {m: importlib.import_module("diffco." + m)
            for m in ("kernel", "utils", "kernel_perceptrons", "model", "robot_fkine", "optim")}
    old = types.ModuleType("olddiffco")
    old.__path__ = [f"{REF}/diffco/deprecated", f"{REF}/diffco"]
    sys.modules["olddiffco"] = old
    for m in ("kernel", "Obstacles", "DiffCo", "MultiDiffCo", "DiffCoBeta"):
        mods["old_" + m] = importlib.import_module("olddiffco." + m)
    return types.SimpleNamespace(**mods)


R = import_reference()


class FKKernelRestated:
    """kernel.py:137-143 behaviour (the reference ctor raises at kernel.py:133)."""

    def __init__(self, fkine, rq_kernel):
        self.fkine, self.rq_kernel = fkine, rq_kernel

    def __call__(self, xs, x_primes=None):
        if xs.ndim == 1:
            xs = xs[None, :]
        a = self.fkine(xs).reshape(len(xs), -1)
        b = self.fkine(x_primes).reshape(len(x_primes), -1)
        return self.rq_kernel(a, b)


# ----------------------------------------------------------------------------- robots
def make_robots():
    m, rf = R.model, R.robot_fkine
    robots = {}
    robots["planar2"] = m.RevolutePlanarRobot(1.0, 0.3, dof=2)
    robots["planar3"] = m.RevolutePlanarRobot([1.0, 0.7, 0.5], 0.2)
    robots["planar7"] = m.RevolutePlanarRobot(0.3, 0.1, dof=7)
    parts = [("box", (0.0, 0.0), (1, 1)), ("box", (1.2, 0.3), (1, 1)), ("box", (-0.8, 0.9), (1, 1))]
    robots["se2"] = m.RigidPlanarBody(parts)
    se3 = m.RigidBody.__new__(m.RigidBody)  # ctor needs trimesh; set the fields fkine reads
    se3.dof = 6
    se3.limits = torch.FloatTensor([[-10, 10]] * 3 + [[-math.pi, math.pi]] * 3)
    corners = torch.tensor([[sx * 0.8, sy * 0.5, sz * 0.3] for sx in (-1, 1) for sy in (-1, 1)
                            for sz in (-1, 1)], dtype=torch.float32).T
    se3.keypoints = corners / corners.norm(dim=0).max()
    robots["se3"] = se3
    robots["baxter_left"] = m.BaxterLeftArmFK()
    robots["baxter_right"] = m.BaxterRightArmFK()
    robots["baxter_dual"] = m.BaxterDualArmFK()
    robots["panda"] = m.PandaFK()
    robots["panda5"] = rf.PandaFK()
    robots["dual_panda"] = m.DualPandaFK()
    return robots


def robot_params(name, rob):
    """Plain-data description of the robot so tests can rebuild it without the reference."""
    p = {}
    if name.startswith("planar"):
        p["link_length"] = rob.link_length.numpy()
    elif name in ("se2", "se3"):
        p["keypoints"] = rob.keypoints.numpy()  # [d, M]
    return p


def rand_cfgs(rob, n, gen, dtype=torch.float32):
    lim = rob.limits
    u = torch.rand((n, lim.shape[0]), generator=gen)
    return (u * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dtype)


def fk64_fn(rob):
    """Reference FK evaluated with float64 configurations (params stay fp32-rounded values).

    Three reference classes cannot run in float64 as written (fp32 keypoints / bases, and
    utils.rot_2d allocates an fp32 result), so for those the fp64 referee casts the fp32
    parameters up, or (SE(2) only) restates R(theta) @ keypoints + t in float64.
    """
    m = R.model
    if isinstance(rob, m.RigidPlanarBody):
        kp = rob.keypoints.double()  # [2, M]

        def f(q):
            q = q.reshape(-1, 3)
            c, s = q[:, 2].cos(), q[:, 2].sin()
            x = c[:, None] * kp[0] - s[:, None] * kp[1] + q[:, 0:1]
            y = s[:, None] * kp[0] + c[:, None] * kp[1] + q[:, 1:2]
            return torch.stack([x, y], dim=2)
        return f
    if isinstance(rob, m.RigidBody):
        def f(q):
            k32 = rob.keypoints
            rob.keypoints = k32.double()
            try:
                return rob.fkine(q)
            finally:
                rob.keypoints = k32
        return f
    if isinstance(rob, m.BaxterDualArmFK):
        def f(q):
            b32 = rob.arm_bases
            rob.arm_bases = b32.double()
            try:
                return rob.fkine(q)
            finally:
                rob.arm_bases = b32
        return f
    if isinstance(rob, m.DualPandaFK):
        def f(q):
            b32 = rob.bases
            rob.bases = b32.double()
            try:
                return rob.fkine(q)
            finally:
                rob.bases = b32
        return f
    return rob.fkine


def fk64(rob, q):
    return fk64_fn(rob)(q.double()).double().clone()


def fk32(rob, q):
    return rob.fkine(q.float()).clone()


# ----------------------------------------------------------------------------- fp64 referee kernels
def k64(kind, params, x, s):
    x = x.reshape(len(x), -1).double()
    s = s.reshape(len(s), -1).double()
    d2 = ((x[:, None, :] - s[None, :, :]) ** 2).sum(-1)
    if kind == "rq":
        g, p = params
        return (1 + g / p * d2) ** (-p)
    if kind == "poly":
        k, eps = params
        # sqrt with a zero (sub)gradient at coincident points, like torch.cdist's backward
        r = torch.where(d2 > 0, d2.clamp_min(1e-300).sqrt(), torch.zeros_like(d2))
        if k % 2 == 0:
            v = r ** k * torch.log(r.clamp_min(1e-300))
            v = torch.where(r == 0, torch.zeros_like(v), v)
            return v / eps
        return r ** k / eps
    if kind == "mq":
        (eps,) = params
        return (d2 / eps ** 2 + 1).sqrt()
    raise ValueError(kind)


def make_kernel(kind, params):
    if kind == "rq":
        return R.kernel.RQKernel(*params)
    if kind == "poly":
        return R.kernel.Polyharmonic(*params)
    if kind == "mq":
        return R.kernel.MultiQuadratic(*params)
    raise ValueError(kind)


# ----------------------------------------------------------------------------- writers
def save(out, name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(os.path.join(out, name + ".npz"), **conv)
    print(f"  wrote {name}.npz  ({sum(a.nbytes for a in conv.values()) / 1024:.0f} KiB raw)")


# ----------------------------------------------------------------------------- A. forward kinematics
def gen_fk(out, robots):
    gen = torch.Generator().manual_seed(100)
    for name, rob in robots.items():
        q = rand_cfgs(rob, 64, gen)
        q[0] = 0.0
        # analytic Jacobian reference: autograd of the reference FK in fp64
        qd = q.double().requires_grad_(True)
        X = fk64_fn(rob)(qd)
        gX = torch.randn(X.shape, generator=gen, dtype=torch.float32).double()  # fp32-representable
        (gq,) = torch.autograd.grad((X * gX).sum(), qd)
        save(out, f"fk_{name}", q=q, x32=fk32(rob, q), x64=fk64(rob, q), gx=gX.float(), gq64=gq,
             limits=rob.limits, **robot_params(name, rob))


# ----------------------------------------------------------------------------- B. kernels
KERNELS = [("rq", (10.0, 2)), ("rq", (3.0, 3)), ("poly", (1, 1.0)), ("poly", (3, 2.0)),
           ("poly", (2, 1.0)), ("mq", (1.0,)), ("mq", (0.5,))]


def gen_kernels(out):
    gen = torch.Generator().manual_seed(200)
    arrs = {}
    for D in (4, 12, 21, 6):
        x = torch.randn((64, D), generator=gen)
        s = torch.randn((96, D), generator=gen)
        s[5] = x[7]  # an exact coincidence (r = 0)
        arrs[f"x_D{D}"], arrs[f"s_D{D}"] = x, s
        for i, (kind, params) in enumerate(KERNELS):
            kf = make_kernel(kind, params)
            arrs[f"k32_D{D}_{i}"] = kf(x, s)
            arrs[f"k64_D{D}_{i}"] = k64(kind, params, x, s)
    arrs["kernel_kinds"] = np.array([k for k, _ in KERNELS])
    arrs["kernel_params"] = np.array([list(p) + [0.0] * (2 - len(p)) for _, p in KERNELS], dtype=np.float64)
    # known answers
    a = torch.zeros(1, 1)
    b = torch.tensor([[1.0], [2.0]])
    arrs["known_rq10"] = R.kernel.RQKernel(10)(a, b).reshape(-1)
    arrs["known_poly11"] = R.kernel.Polyharmonic(1, 1)(a, b).reshape(-1)
    arrs["known_poly32"] = R.kernel.Polyharmonic(3, 2)(a, b).reshape(-1)
    arrs["known_poly21"] = R.kernel.Polyharmonic(2, 1)(a, b).reshape(-1)
    save(out, "kernels", **arrs)


# ----------------------------------------------------------------------------- C. score + gradient
def new_diffco(rob, kind, params, sup_q, weights, which):
    """A reference DiffCo (new API) with its inference state set directly."""
    dc = R.kernel_perceptrons.DiffCo(kernel_func=make_kernel(kind, params) if which == "score" else "rq",
                                     transform=None if rob is None else rob.fkine)
    dc.support_points = sup_q
    dc.support_transformed = sup_q if rob is None else fk32(rob, sup_q)
    if which == "score":
        dc.gains = weights
    else:
        dc.rbf_kernel = make_kernel(kind, params)
        dc.rbf_nodes = weights
    return dc


def score_case(out, name, rob, kind, params, S, B, C, gen, which, zero_frac=0.0, coincide=False):
    if rob is None:
        lim = torch.tensor([[-10.0, 10.0]] * 3 + [[-math.pi, math.pi]] * 3)
        fake = types.SimpleNamespace(limits=lim)
        q, sup_q = rand_cfgs(fake, B, gen), rand_cfgs(fake, S, gen)
    else:
        q, sup_q = rand_cfgs(rob, B, gen), rand_cfgs(rob, S, gen)
    if coincide:
        q[3] = sup_q[11]
    W = torch.randn((S, C), generator=gen)
    if zero_frac > 0:
        W = W * (torch.rand((S, C), generator=gen) >= zero_frac)
    T = (lambda t: t) if rob is None else fk64_fn(rob)
    sup_x32 = sup_q if rob is None else fk32(rob, sup_q)
    arrs = dict(q=q, sup_q=sup_q, sup_x32=sup_x32, weights=W, kind=np.array(kind),
                kparams=np.array(list(params), dtype=np.float64))

    # ---- reference fp32 path, exactly as the reference classes compute it
    qv = q.clone().requires_grad_(True)
    if which != "multi":
        dc = new_diffco(rob, kind, params, sup_q, W[:, 0].clone(), which)
        s = dc.score(qv) if which == "score" else dc.poly_score(qv)
        arrs["score32"] = s.detach().clone()
        (g,) = torch.autograd.grad(s.sum(), qv)
        arrs["grad32"] = g
    else:
        md = R.old_MultiDiffCo.MultiDiffCo.__new__(R.old_MultiDiffCo.MultiDiffCo)
        md.fkine = None if rob is None else rob.fkine
        md.support_points = sup_q
        md.support_fkine = sup_x32.reshape(S, -1)
        md.rbf_kernel = make_kernel(kind, params)
        md.rbf_nodes = W
        md.num_class = C
        s = md.rbf_score(qv)  # [B, C]
        arrs["score32"] = s.detach().clone()
        (g,) = torch.autograd.grad(s.sum(), qv, retain_graph=True)
        arrs["grad32"] = g
        up = torch.randn((B, C), generator=gen)
        (gv,) = torch.autograd.grad((s * up).sum(), qv, retain_graph=True)
        arrs["upstream"], arrs["vjp32"] = up, gv
        nj = min(B, 32)
        jac = torch.stack([torch.autograd.grad(s[:nj, c].sum(), qv, retain_graph=True)[0][:nj]
                           for c in range(C)], dim=1)
        arrs["jac32"] = jac  # [nj, C, dof]

    # ---- fp64 referee (direct differences)
    qd = q.double().requires_grad_(True)
    Xd = T(qd)
    Sd = (sup_q.double() if rob is None else fk64(rob, sup_q))
    K = k64(kind, params, Xd, Sd)
    s64 = K @ W.double()
    arrs["score64"] = s64.detach()
    (g64,) = torch.autograd.grad(s64.sum(), qd, retain_graph=True)
    arrs["grad64"] = g64
    if which == "multi":
        (gv64,) = torch.autograd.grad((s64 * arrs["upstream"].double()).sum(), qd, retain_graph=True)
        arrs["vjp64"] = gv64
    save(out, name, **arrs)


def gen_scores(out, robots):
    gen = torch.Generator().manual_seed(300)
    # BASELINE config #1 (full size): planar 2-DoF, RQ(10) gains, S=200, B=256
    score_case(out, "cfg1_planar2_rq", robots["planar2"], "rq", (10.0, 2), 200, 256, 1, gen, "score")
    # config #2: Baxter / Panda, S=1000; B reduced to 512 except the Baxter poly case (full 4096)
    score_case(out, "cfg2_baxter_poly1", robots["baxter_left"], "poly", (1, 1.0), 1000, 4096, 1, gen, "poly")
    score_case(out, "cfg2_baxter_rq", robots["baxter_left"], "rq", (10.0, 2), 1000, 512, 1, gen, "score")
    score_case(out, "cfg2_panda_poly1", robots["panda"], "poly", (1, 1.0), 1000, 512, 1, gen, "poly")
    score_case(out, "cfg2_panda_rq", robots["panda"], "rq", (10.0, 2), 1000, 512, 1, gen, "score")
    # headline shape at reduced batch: Baxter, Polyharmonic(1,1), S=2000
    score_case(out, "headline_baxter_poly1_s2000", robots["baxter_left"], "poly", (1, 1.0), 2000, 512, 1, gen, "poly")
    # config #3: C=5, S=2000, 40 % zeros per class, B reduced
    score_case(out, "cfg3_baxter_rq_c5", robots["baxter_left"], "rq", (10.0, 2), 2000, 256, 5, gen, "multi", 0.4)
    score_case(out, "cfg3_baxter_poly1_c5", robots["baxter_left"], "poly", (1, 1.0), 2000, 256, 5, gen, "multi", 0.4)
    # config #4: SE(3), RQ(10), S=10k, no FK (D=6) and 8-keypoint variant (D=24); B reduced
    score_case(out, "cfg4_se3_nofk_rq", None, "rq", (10.0, 2), 10000, 256, 1, gen, "score")
    score_case(out, "cfg4_se3_keypts_rq", robots["se3"], "rq", (10.0, 2), 2000, 256, 1, gen, "score")
    # the other FK classes x the other kernels (small)
    score_case(out, "misc_dualbaxter_poly1", robots["baxter_dual"], "poly", (1, 1.0), 300, 128, 1, gen, "poly")
    score_case(out, "misc_dualpanda_rq", robots["dual_panda"], "rq", (10.0, 2), 300, 128, 1, gen, "score")
    score_case(out, "misc_panda5_mq", robots["panda5"], "mq", (1.0,), 300, 128, 1, gen, "multi")
    score_case(out, "misc_se2_poly3", robots["se2"], "poly", (3, 2.0), 300, 128, 1, gen, "multi")
    score_case(out, "misc_planar3_poly2", robots["planar3"], "poly", (2, 1.0), 300, 128, 1, gen, "poly")
    score_case(out, "misc_planar7_rq_p3", robots["planar7"], "rq", (3.0, 3), 300, 128, 1, gen, "score")
    score_case(out, "misc_baxterR_mq_c2", robots["baxter_right"], "mq", (0.5,), 300, 128, 2, gen, "multi")
    # edge: query coincides with a support (r = 0) for the kinked kernel
    score_case(out, "edge_r0_baxter_poly1", robots["baxter_left"], "poly", (1, 1.0), 64, 16, 1, gen, "poly",
               coincide=True)
    score_case(out, "edge_r0_planar3_poly2", robots["planar3"], "poly", (2, 1.0), 64, 16, 1, gen, "poly",
               coincide=True)


def gen_edges(out, robots):
    gen = torch.Generator().manual_seed(400)
    rob = robots["baxter_left"]
    sup_q = rand_cfgs(rob, 50, gen)
    w = torch.randn(50, generator=gen)
    arrs = dict(sup_q=sup_q, weights=w)
    # B == 1 : RQKernel squeezes (kernel.py:26-27) -> score is 0-dim; poly_score stays [1,1]
    q1 = rand_cfgs(rob, 1, gen)[0]
    dc = new_diffco(rob, "rq", (10.0, 2), sup_q, w, "score")
    s = dc.score(q1)
    arrs["q1"], arrs["score_b1"], arrs["score_b1_shape"] = q1, s.reshape(-1), np.array(s.shape, dtype=np.int64)
    dp = new_diffco(rob, "poly", (1, 1.0), sup_q, w, "poly")
    s = dp.poly_score(q1)
    arrs["poly_b1"], arrs["poly_b1_shape"] = s.reshape(-1), np.array(s.shape, dtype=np.int64)
    # fp64 input to poly_score: cast to fp32 nodes' dtype (kernel_perceptrons.py:313), grad comes back fp64
    qd = rand_cfgs(rob, 8, gen, torch.float64).requires_grad_(True)
    s = dp.poly_score(qd)
    (g,) = torch.autograd.grad(s.sum(), qd)
    arrs["q_f64"], arrs["poly_f64in"], arrs["grad_f64in"] = qd.detach(), s.detach(), g
    arrs["poly_f64in_dtype"] = np.array(str(s.dtype))
    arrs["grad_f64in_dtype"] = np.array(str(g.dtype))
    # transformed_point bypass (kernel_perceptrons.py:316-317)
    qb = rand_cfgs(rob, 8, gen)
    arrs["q_tp"], arrs["poly_tp"] = qb, dp.poly_score(transformed_point=fk32(rob, qb))
    # zero padding of max_num_supports (rows of zeros with zero weight)
    sup_pad = torch.cat([sup_q, torch.zeros(14, 7)])
    w_pad = torch.cat([w, torch.zeros(14)])
    dz = new_diffco(rob, "poly", (1, 1.0), sup_pad, w_pad, "poly")
    dz.support_transformed = torch.cat([fk32(rob, sup_q), torch.zeros(14, 4, 3)])
    arrs["poly_padded"] = dz.poly_score(qb)
    save(out, "edges", **arrs)


# ----------------------------------------------------------------------------- E. trained models
def synth_labels(rob, X, centers, radius):
    P = fk32(rob, X)  # [N, m, d]
    d = (P[:, :, None, :] - centers[None, None]).norm(dim=-1) - radius  # [N, m, n_obs]
    dist = -d.min(dim=1).values  # [N, n_obs]   > 0 inside
    return dist


def gen_trained(out, robots):
    gen = torch.Generator().manual_seed(500)
    rob = robots["baxter_left"]
    X = rand_cfgs(rob, 3000, gen)
    centers = torch.tensor([[0.7, 0.3, 0.3], [0.4, -0.5, 0.0]])
    dist = synth_labels(rob, X, centers, 0.25).max(dim=1).values
    y = torch.where(dist > 0, 1.0, -1.0)
    dc = R.kernel_perceptrons.DiffCo(kernel_func=R.kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine)
    dc.train(X, y, max_iteration=3000, distance=dist)
    arrs = dict(X=X, y=y, dist=dist, support_points=dc.support_points, support_transformed=dc.support_transformed,
                gains=dc.gains, hypothesis=dc.hypothesis, kernel_matrix_diag=torch.diag(dc.kernel_matrix),
                sup_y=dc.y, sup_dist=dc.distance)
    print(f"  trained Baxter DiffCo: {len(dc.gains)} supports")
    for tgt in ("label", "hypo", "dist"):
        dc.fit_poly(R.kernel.Polyharmonic(1, 1.0), target=tgt)
        arrs[f"rbf_nodes_{tgt}"] = dc.rbf_nodes.clone()
    dc.fit_poly(R.kernel.Polyharmonic(1, 1.0), target="label")
    qt = rand_cfgs(rob, 256, gen)
    qv = qt.clone().requires_grad_(True)
    s = dc.poly_score(qv)
    (g,) = torch.autograd.grad(s.sum(), qv)
    arrs.update(q_test=qt, poly_test=s.detach(), poly_grad_test=g, score_test=dc.score(qt))
    # active-learning update (jump start): new samples + existing supports, exist_mask marks the last S rows
    Xn = rand_cfgs(rob, 500, gen)
    centers2 = centers + torch.tensor([[0.0, 0.1, 0.05], [0.05, 0.0, 0.1]])
    Xu = torch.cat([Xn, dc.support_points])
    du = synth_labels(rob, Xu, centers2, 0.25).max(dim=1).values
    yu = torch.where(du > 0, 1.0, -1.0)
    mask = torch.zeros(len(Xu), dtype=torch.bool)
    mask[-len(dc.support_points):] = True
    dc.train(Xu, yu, update=True, exist_mask=mask, max_iteration=2000, distance=du)
    arrs.update(Xu=Xu, yu=yu, du=du, exist_mask=mask, upd_support_points=dc.support_points, upd_gains=dc.gains,
                upd_hypothesis=dc.hypothesis)
    print(f"  after update: {len(dc.gains)} supports")
    # max_num_supports variant (zero padded / truncated)
    dm = R.kernel_perceptrons.DiffCo(kernel_func=R.kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine,
                                     max_num_supports=300)
    dm.train(X, y, max_iteration=3000, distance=dist)
    dm.fit_poly(R.kernel.Polyharmonic(1, 1.0), target="label")
    arrs.update(mns_support_points=dm.support_points, mns_gains=dm.gains, mns_rbf_nodes=dm.rbf_nodes,
                mns_valid=np.array(dm.valid_supports), mns_poly_test=dm.poly_score(qt))
    save(out, "trained_baxter", **arrs)

    # old API: MultiDiffCo, C=2, planar 2-DoF with FKKernel (scripts/active.py:605-672 call pattern)
    rob2 = robots["planar2"]
    X2 = rand_cfgs(rob2, 1500, gen)
    c2 = torch.tensor([[1.2, 0.8], [-0.9, -1.0]])
    d2 = synth_labels(rob2, X2, c2, 0.45)  # [N, 2]
    y2 = torch.where(d2 > 0, 1.0, -1.0)
    fkk = FKKernelRestated(rob2.fkine, R.kernel.RQKernel(10.0))
    md = R.old_MultiDiffCo.MultiDiffCo(None, kernel_func=fkk, beta=1.0)
    md.train(X2, y2, max_iteration=1500, distance=d2)
    md.fit_poly(kernel_func=R.kernel.Polyharmonic(1, 1.0), target="label", fkine=rob2.fkine, reg=0.0)
    qt2 = rand_cfgs(rob2, 256, gen)
    qv2 = qt2.clone().requires_grad_(True)
    s2 = md.rbf_score(qv2)
    (g2,) = torch.autograd.grad(s2.sum(), qv2)
    print(f"  trained planar MultiDiffCo: {len(md.gains)} supports")
    save(out, "trained_multi_planar2", X=X2, y=y2, dist=d2, support_points=md.support_points, gains=md.gains,
         hypothesis=md.hypothesis, rbf_nodes=md.rbf_nodes, q_test=qt2, rbf_test=s2.detach(), rbf_grad_test=g2,
         score_test=md.score(qt2))
    return dc


# ----------------------------------------------------------------------------- E2. old single-class API
def gen_old_single(out, robots):
    """deprecated/DiffCo.py through the exact call sequence of scripts/speed_compare.py:220-257 (FKKernel perceptron ->
    score -> fit_poly(Polyharmonic(1, 1), 'label') WITHOUT fkine -> the spline score), the same with fkine, and
    deprecated/DiffCoBeta.py: rbf_score on directly set state (:173-181) and its train() (:23-111).  DiffCoBeta.train
    calls torch.solve, which torch >= 1.13 no longer has; for that one call this script binds the documented
    replacement (torch.linalg.solve with the arguments swapped) to the old name — the routine itself runs unmodified."""
    gen = torch.Generator().manual_seed(900)
    rob = robots["baxter_left"]
    n_train, n_test = 2000, 400
    X = rand_cfgs(rob, n_train + n_test, gen)
    centers = torch.tensor([[0.7, 0.3, 0.3], [0.4, -0.5, 0.0]])
    dist = synth_labels(rob, X, centers, 0.25).max(dim=1).values
    y = torch.where(dist > 0, 1.0, -1.0)
    fkk = FKKernelRestated(rob.fkine, R.kernel.RQKernel(10.0))
    ck = R.old_DiffCo.DiffCo(None, kernel_func=fkk, beta=1.0)
    ck.train(X[:n_train], y[:n_train], max_iteration=n_train, distance=dist[:n_train])
    qt = X[n_train:]

    def with_grad(fn):
        qv = qt.clone().requires_grad_(True)
        sv = fn(qv)
        (g,) = torch.autograd.grad(sv.sum(), qv)
        return sv.detach(), g

    arrs = dict(X=X[:n_train], y=y[:n_train], dist=dist[:n_train], q_test=qt, support_points=ck.support_points,
                gains=ck.gains, hypothesis=ck.hypothesis, sup_y=ck.y, sup_dist=ck.distance)
    arrs["score_test"], arrs["score_grad_test"] = with_grad(ck.score)
    arrs["score_single"] = ck.score(qt[0])
    ck.fit_poly(kernel_func=R.kernel.Polyharmonic(1, 1.0), target="label")        # speed_compare.py:236: no fkine
    arrs["nodes_nofk"] = ck.rbf_nodes.clone()
    arrs["poly_nofk_test"], arrs["poly_nofk_grad_test"] = with_grad(ck.poly_score)
    for tgt in ("dist", "hypo"):
        ck.fit_poly(kernel_func=R.kernel.Polyharmonic(1, 1.0), target=tgt)
        arrs[f"nodes_nofk_{tgt}"] = ck.rbf_nodes.clone()
    ck.fit_poly(kernel_func=R.kernel.Polyharmonic(1, 1.0), target="label", fkine=rob.fkine)
    arrs["nodes_fk"] = ck.rbf_nodes.clone()
    arrs["support_fkine"] = ck.support_fkine
    arrs["poly_fk_test"], arrs["poly_fk_grad_test"] = with_grad(ck.poly_score)
    print(f"  old DiffCo on Baxter: {len(ck.gains)} supports")
    save(out, "old_single_baxter", **arrs)

    # DiffCoBeta.rbf_score on directly set state, with and without fkine
    Beta = R.old_DiffCoBeta.DiffCoBeta
    S = 300
    sup = rand_cfgs(rob, S, gen)
    nodes = torch.randn(S, generator=gen)
    b = Beta.__new__(Beta)
    b.rbf_kernel = R.kernel.Polyharmonic(1, 1.0)
    b.support_points, b.rbf_nodes = sup, nodes
    b.fkine, b.support_fkine = rob.fkine, rob.fkine(sup).reshape(S, -1)
    arrs = dict(support_points=sup, rbf_nodes=nodes, q_test=qt)
    ck = b
    arrs["rbf_fk_test"], arrs["rbf_fk_grad_test"] = with_grad(b.rbf_score)
    b.fkine = None
    b.rbf_kernel = R.kernel.MultiQuadratic(1.5)
    arrs["rbf_nofk_mq_test"], arrs["rbf_nofk_mq_grad_test"] = with_grad(b.rbf_score)
    arrs["rbf_single"] = b.rbf_score(qt[0])
    # DiffCoBeta.train (perceptron on sign(d), then rbf_nodes = solve(K_rbf + 0.1 I, d) over supports + left-out samples)
    import collections
    Sol = collections.namedtuple("solve", ["solution", "LU"])
    stub = getattr(torch, "solve", None)   # torch >= 1.13 keeps a stub that raises and names the replacement
    torch.solve = lambda B, A: Sol(torch.linalg.solve(A, B), None)   # torch < 1.13 signature: solve(B, A)
    try:
        bt = Beta(None, kernel_func=fkk, beta=1.0)
        nb = 800
        bt.train(X[:nb], dist[:nb], fkine=rob.fkine, max_iteration=nb, n_left_out_points=100)
    finally:
        if stub is not None:
            torch.solve = stub
        else:
            del torch.solve
    ck = bt
    arrs.update(train_X=X[:nb], train_d=dist[:nb], tr_support_points=bt.support_points, tr_gains=bt.gains,
                tr_rbf_nodes=bt.rbf_nodes, tr_hypothesis=bt.hypothesis, tr_distance=bt.distance,
                tr_num_origin_supports=np.array(bt.num_origin_supports))
    arrs["tr_rbf_test"], arrs["tr_rbf_grad_test"] = with_grad(bt.rbf_score)
    print(f"  DiffCoBeta on Baxter: {bt.num_origin_supports} perceptron supports + 100 left-out samples")
    save(out, "old_beta_baxter", **arrs)


# ----------------------------------------------------------------------------- F. optimiser records
HESS_CASES = {  # fixture -> (robot, kind, params, which); inputs are the ones the score fixture already holds
    "cfg1_planar2_rq": ("planar2", "rq", (10.0, 2), "score"), "cfg2_baxter_poly1": ("baxter_left", "poly", (1, 1.0), "poly"),
    "cfg2_baxter_rq": ("baxter_left", "rq", (10.0, 2), "score"), "cfg2_panda_rq": ("panda", "rq", (10.0, 2), "score"),
    "cfg3_baxter_rq_c5": ("baxter_left", "rq", (10.0, 2), "multi"), "cfg4_se3_nofk_rq": (None, "rq", (10.0, 2), "score"),
    "cfg4_se3_keypts_rq": ("se3", "rq", (10.0, 2), "score"), "misc_dualbaxter_poly1": ("baxter_dual", "poly", (1, 1.0), "poly"),
    "misc_dualpanda_rq": ("dual_panda", "rq", (10.0, 2), "score"), "misc_panda5_mq": ("panda5", "mq", (1.0,), "multi"),
    "misc_se2_poly3": ("se2", "poly", (3, 2.0), "multi"), "misc_planar3_poly2": ("planar3", "poly", (2, 1.0), "poly"),
    "misc_planar7_rq_p3": ("planar7", "rq", (3.0, 3), "score"), "misc_baxterR_mq_c2": ("baxter_right", "mq", (0.5,), "multi"),
}


def gen_hess(out, robots):
    """Second derivatives of the score at the first 6 configurations of existing score fixtures, the way the reference
    obtains them for trust-constr's constraint Hessian (optim.py:383-388): torch.autograd.functional.hessian over
    the reference's own dist_est (fp32, `hess32`; skipped where its graph is not twice differentiable) and over the
    float64 referee of score_case (`hess64`).  For C > 1 the function is sum_c upstream[b, c] * score[b, c]."""
    n = 6
    arrs = {}
    for name, (rname, kind, params, which) in HESS_CASES.items():
        d = np.load(os.path.join(out, name + ".npz"))
        rob = None if rname is None else robots[rname]
        q, sup_q, W = (torch.from_numpy(d[k]) for k in ("q", "sup_q", "weights"))
        sup_x32 = torch.from_numpy(d["sup_x32"])
        S, C = W.shape
        up = torch.from_numpy(d["upstream"])[:n] if "upstream" in d.files else torch.ones((n, C))
        T = (lambda t: t) if rob is None else fk64_fn(rob)
        Sd = sup_q.double() if rob is None else fk64(rob, sup_q)

        def f64(qb, b):
            return (k64(kind, params, T(qb[None]), Sd) @ W.double() * up[b].double()).sum()
        arrs[name + "/hess64"] = torch.stack([torch.autograd.functional.hessian(lambda z, b=b: f64(z, b), q[b].double())
                                              for b in range(n)])
        if which != "multi":
            dc = new_diffco(rob, kind, params, sup_q, W[:, 0].clone(), which)
            dist_est = dc.score if which == "score" else dc.poly_score
        else:
            md = R.old_MultiDiffCo.MultiDiffCo.__new__(R.old_MultiDiffCo.MultiDiffCo)
            md.fkine = None if rob is None else rob.fkine
            md.support_points, md.support_fkine = sup_q, sup_x32.reshape(S, -1)
            md.rbf_kernel, md.rbf_nodes, md.num_class = make_kernel(kind, params), W, C
            dist_est = md.rbf_score
        try:
            h32 = torch.stack([torch.autograd.functional.hessian(
                lambda z, b=b: (dist_est(z[None]).reshape(1, -1) * up[b]).sum(), q[b]) for b in range(n)])
            if torch.isfinite(h32).all():
                arrs[name + "/hess32"] = h32
        except RuntimeError as e:
            print(f"  {name}: the reference's fp32 graph is not twice differentiable here ({str(e)[:60]})")
        rel = (arrs[name + "/hess32"].double() - arrs[name + "/hess64"]).abs().max() / arrs[name + "/hess64"].abs().max() \
            if name + "/hess32" in arrs else float("nan")
        print(f"  {name}: |H|max {float(arrs[name + '/hess64'].abs().max()):.3g}, fp32 reference vs fp64 referee {float(rel):.1e}")
    save(out, "hess_points", n=np.array(n), **arrs)


def gen_optim(out, robots):
    gen = torch.Generator().manual_seed(600)
    rob = robots["baxter_left"]
    sup_q = rand_cfgs(rob, 200, gen)
    w = torch.randn(200, generator=gen) * 0.05
    dc = new_diffco(rob, "poly", (1, 1.0), sup_q, w, "poly")
    start, target = rand_cfgs(rob, 1, gen)[0], rand_cfgs(rob, 1, gen)[0]
    n_wp = 20
    t = torch.linspace(0, 1, n_wp)[:, None].double()
    init = start.double() * (1 - t) + target.double() * t
    init[1:-1] += 0.05 * torch.randn((n_wp - 2, 7), generator=gen).double()
    options = {"N_WAYPOINTS": n_wp, "NUM_RE_TRIALS": 1, "MAXITER": 50, "safety_margin": 0.0, "max_speed": 0.3,
               "seed": 1234, "history": False, "extra_optimizer_options": {"lr": 0.05},
               "init_solution": init.clone()}
    # one fused-loss evaluation at the initial path (the quantity the fused Adam step differentiates)
    p = init.clone().requires_grad_(True)
    col = torch.clamp(dc.poly_score(p) - 0.0, min=0).sum()
    cp = rob.fkine(p)
    mm = torch.clamp((cp[1:] - cp[:-1]).square().sum(dim=2) - 0.3 ** 2, min=0).sum()
    jl = (torch.clamp(rob.limits[:, 0] - p, min=0) + torch.clamp(p - rob.limits[:, 1], min=0)).sum()
    diff = (cp[1:] - cp[:-1]).square().sum()
    loss = diff + 10 * col + 10 * mm + 10 * jl
    (gl,) = torch.autograd.grad(loss, p)
    rec = R.optim.adam_traj_optimize(rob, dc.poly_score, start, target, dict(options))
    save(out, "optim_adam_baxter", sup_q=sup_q, weights=w, start=start, target=target, init=init,
         loss0=loss.detach(), loss0_terms=torch.stack([diff, col, mm, jl]).detach(), grad0=gl,
         solution=np.array(rec["solution"]), cost=np.array(rec["cost"]), cnt_check=np.array(rec["cnt_check"]),
         success=np.array(rec["success"]))
    with open(os.path.join(out, "optim_adam_baxter_options.json"), "w") as f:
        json.dump({k: v for k, v in options.items() if k != "init_solution"}, f, indent=1)
    print(f"  adam record: success={rec['success']} cost={rec['cost']:.6f} cnt_check={rec['cnt_check']}")

    # dense_path (utils.py:87-101) vectors
    path = rand_cfgs(rob, 6, gen).double()
    save(out, "dense_path", path=path, dense_0p3=R.utils.dense_path(path, 0.3),
         dense_2p0=R.utils.dense_path(path, 2.0))


def gen_optim_scipy(out, robots):
    """Row f4: the reference's SLSQP and trust-constr drivers (optim.py:166-321, 324-516) run end to end on
    dc.poly_score with `init_solution`, and the collision constraint they build - con_collision_free (:190-207), its
    Jacobian (:209-218) and the Hessian of v . con (:380-391) - re-evaluated at the initial path through the imported
    reference dist_est / utils.dense_path (the drivers' closures cannot be called from outside).  The constraint is
    stored twice: through the reference's own fp32 dist_est (poly_score casts its input, kernel_perceptrons.py:313) and
    through the same model held in float64 (the tolerance referee)."""
    gen = torch.Generator().manual_seed(601)
    rob = robots["baxter_left"]
    S = 300
    sup_q = rand_cfgs(rob, S, gen)
    w = torch.randn(S, generator=gen) * 0.05
    dc = new_diffco(rob, "poly", (1, 1.0), sup_q, w, "poly")
    # the same model in float64 (reference classes, double state)
    dc64 = R.kernel_perceptrons.DiffCo(kernel_func="rq", transform=rob.fkine)
    dc64.support_points = sup_q.double()
    dc64.support_transformed = rob.fkine(sup_q.double())
    dc64.rbf_kernel = make_kernel("poly", (1, 1.0))
    dc64.rbf_nodes = w.double()
    start, target = rand_cfgs(rob, 1, gen)[0], rand_cfgs(rob, 1, gen)[0]
    n_wp, max_speed = 12, 0.3
    t = torch.linspace(0, 1, n_wp)[:, None].double()
    init = start.double() * (1 - t) + target.double() * t
    init[1:-1] += 0.05 * torch.randn((n_wp - 2, 7), generator=gen).double()
    dense = R.utils.dense_path(init, max_speed)
    with torch.no_grad():
        s_dense = dc64.poly_score(dense[1:-1]).reshape(-1)
    # ~60 % of the dense points violate the margin: an active constraint.  The margin sits BETWEEN two scores (round 6: the 0.4
    # quantile of 16 points is the 7th score itself - a dense point exactly on the hinge, whose side fp32 rounding decided)
    srt = s_dense.sort().values
    k40 = int(0.4 * (len(srt) - 1))
    margin = float(0.5 * (srt[k40] + srt[k40 + 1]))

    def con(p, dist_est):  # optim.py:190-207 with return_tensor=True
        dense_p = R.utils.dense_path(p, max_speed)
        cost = -(dist_est(dense_p[1:-1]) - margin)
        cost = torch.clamp(cost, max=0).reshape(-1)
        n_segment, n_point = len(p) - 1, len(dense_p) - 2
        mult = n_point // n_segment
        if n_point % n_segment != 0:
            mult += 1
            cost = torch.cat([cost, torch.zeros(n_segment * mult - n_point, dtype=cost.dtype)])
        return cost.reshape(n_segment, -1).sum(dim=1)

    vvec = torch.rand(n_wp - 1, generator=gen).double()
    pieces = {}
    for tag, de in (("ref", dc.poly_score), ("f64", dc64.poly_score)):
        p = init.clone().requires_grad_(True)
        c0 = con(p, de).detach()
        jac = torch.autograd.functional.jacobian(lambda x: con(x, de), p, create_graph=False, strict=False,
                                                 vectorize=True, strategy="reverse-mode")
        jac = jac[:, 1:-1].reshape(jac.shape[0], -1)                       # optim.py:217-218
        vv = vvec.to(c0.dtype)
        hess = torch.autograd.functional.hessian(lambda x: torch.dot(con(x, de), vv), p, create_graph=False, strict=False,
                                                 vectorize=True, outer_jacobian_strategy="reverse-mode")
        hess = hess[1:-1, :, 1:-1, :].reshape((n_wp - 2) * 7, -1)          # optim.py:389-390
        pieces[tag] = (c0.double(), jac.double(), hess.double())
    print(f"  constraint at the initial path: {int((pieces['f64'][0] < 0).sum())} of {n_wp - 1} segments active, "
          f"fp32-vs-fp64 jac {float((pieces['ref'][1] - pieces['f64'][1]).abs().max() / pieces['f64'][1].abs().max()):.2e}, "
          f"hess {float((pieces['ref'][2] - pieces['f64'][2]).abs().max() / pieces['f64'][2].abs().max()):.2e}")
    options = {"N_WAYPOINTS": n_wp, "NUM_RE_TRIALS": 1, "MAXITER": 12, "safety_margin": margin, "max_speed": max_speed,
               "seed": 4321, "history": False, "extra_optimizer_options": {"disp": False}}
    recs = {}
    for name, fn in (("slsqp", R.optim.givengrad_traj_optimize), ("trustconstr", R.optim.trustconstr_traj_optimize)):
        o = dict(options, init_solution=init.clone())
        if name == "trustconstr":
            o["MAXITER"] = 6
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            rec = fn(rob, dc.poly_score, start.double(), target.double(), o)
        recs[name] = rec
        print(f"  {name}: success={rec['success']} cost={float(rec['cost']):.6f} cnt_check={rec['cnt_check']}")
    save(out, "optim_scipy_baxter", sup_q=sup_q, weights=w, start=start, target=target, init=init, margin=np.array(margin),
         max_speed=np.array(max_speed), n_dense=np.array(len(dense)), v=vvec,
         con0_ref=pieces["ref"][0], jac0_ref=pieces["ref"][1], hess0_ref=pieces["ref"][2],
         con0_f64=pieces["f64"][0], jac0_f64=pieces["f64"][1], hess0_f64=pieces["f64"][2],
         slsqp_solution=np.array(recs["slsqp"]["solution"]), slsqp_cost=np.array(float(recs["slsqp"]["cost"])),
         slsqp_cnt_check=np.array(recs["slsqp"]["cnt_check"]), slsqp_success=np.array(recs["slsqp"]["success"]),
         slsqp_maxiter=np.array(12),
         tc_solution=np.array(recs["trustconstr"]["solution"]), tc_cost=np.array(float(recs["trustconstr"]["cost"])),
         tc_cnt_check=np.array(recs["trustconstr"]["cnt_check"]), tc_success=np.array(recs["trustconstr"]["success"]),
         tc_maxiter=np.array(6))


def gen_frames(out, robots):
    """utils.DH2mat (utils.py:66-75) and utils.euler2mat (utils.py:15-38) called as the reference's robot classes call them
    (model.py:230 with Baxter's DH table, model.py:437 with Panda's; RigidBody.fkine's euler2mat, model.py:156-159): fp32 outputs,
    an fp64 evaluation of the same functions, and vector-Jacobian products from autograd in fp64"""
    gen = torch.Generator().manual_seed(1900)
    arrs = {}
    for name in ("baxter_left", "panda"):
        rob = robots[name]
        q = rand_cfgs(rob, 48, gen)
        q[0] = 0.0
        ang = q + rob.dhparams.theta                                       # model.py:229
        a, dd, sa, ca = rob.dhparams.a, rob.dhparams.d, rob.s_alpha, rob.c_alpha
        T32 = R.utils.DH2mat(ang, a, dd, sa, ca)
        qd = ang.double().requires_grad_(True)
        T64 = R.utils.DH2mat(qd, a.double(), dd.double(), sa.double(), ca.double())
        gT = torch.randn(T64.shape, generator=gen, dtype=torch.float32).double()
        (gq,) = torch.autograd.grad((T64 * gT).sum(), qd)
        arrs.update({f"{name}_q": ang, f"{name}_a": a, f"{name}_d": dd, f"{name}_sa": sa, f"{name}_ca": ca, f"{name}_T32": T32,
                     f"{name}_T64": T64.detach(), f"{name}_gT": gT.float(), f"{name}_gq64": gq})
    phi = (torch.rand((64, 3), generator=gen) * 2 - 1) * math.pi
    phi[0] = 0.0
    pd = phi.double().requires_grad_(True)
    R64 = R.utils.euler2mat(pd)
    gR = torch.randn(R64.shape, generator=gen, dtype=torch.float32).double()
    (gp,) = torch.autograd.grad((R64 * gR).sum(), pd)
    arrs.update(euler_phi=phi, euler_R32=R.utils.euler2mat(phi), euler_R64=R64.detach(), euler_gR=gR.float(), euler_gphi64=gp)
    save(out, "frames", **arrs)


def gen_optim_multi(out, robots):
    """Rows f2 / f4 for a MULTI-CLASS checker (round 6): the reference's adam_traj_optimize (optim.py:13-163) run unmodified on an
    old-API MultiDiffCo's rbf_score ([W, C] scores, deprecated/MultiDiffCo.py:156-169) with options['safety_margin'] a [C] tensor -
    the call of scripts/2d_trajopt.py:94-102 / scripts/active.py:28-121, on BASELINE config #3's shape (Baxter, five classes,
    Polyharmonic nodes with 40 % of the entries zero) - and the collision constraint of the scipy drivers (optim.py:190-218,
    380-391) on the same checker at a path whose dense point count divides by the segment count (the reference's flat reshape
    only exists then for C > 1).  The checker's state is held in float64 (rbf_score does not cast: the optimisers' float64
    waypoints need float64 supports)."""
    gen = torch.Generator().manual_seed(1800)
    rob = robots["baxter_left"]
    S, C = 300, 5
    sup_q = rand_cfgs(rob, S, gen)
    Wn = (torch.randn((S, C), generator=gen) * 0.05 + 0.002) * (torch.rand((S, C), generator=gen) >= 0.4)
    md = R.old_MultiDiffCo.MultiDiffCo.__new__(R.old_MultiDiffCo.MultiDiffCo)
    md.fkine, md.support_points = rob.fkine, sup_q.double()
    md.support_fkine = rob.fkine(sup_q.double()).reshape(S, -1)
    md.rbf_kernel, md.rbf_nodes, md.num_class = make_kernel("poly", (1, 1.0)), Wn.double(), C
    start, target = rand_cfgs(rob, 1, gen)[0], rand_cfgs(rob, 1, gen)[0]
    n_wp = 20
    t = torch.linspace(0, 1, n_wp)[:, None].double()
    init = start.double() * (1 - t) + target.double() * t
    init[1:-1] += 0.05 * torch.randn((n_wp - 2, 7), generator=gen).double()
    with torch.no_grad():
        s_init = md.rbf_score(init)
    # per-class margins that leave the hinge active on part of the path in every class (the 60 % quantile per class)
    margin = s_init.quantile(0.6, dim=0).float()
    options = {"N_WAYPOINTS": n_wp, "NUM_RE_TRIALS": 1, "MAXITER": 50, "safety_margin": margin.double(), "max_speed": 0.3,
               "seed": 1234, "history": False, "extra_optimizer_options": {"lr": 0.05}, "init_solution": init.clone()}
    p = init.clone().requires_grad_(True)
    col = torch.clamp(md.rbf_score(p) - margin.double(), min=0).sum()
    cp = rob.fkine(p)
    mm = torch.clamp((cp[1:] - cp[:-1]).square().sum(dim=2) - 0.3 ** 2, min=0).sum()
    jl = (torch.clamp(rob.limits[:, 0] - p, min=0) + torch.clamp(p - rob.limits[:, 1], min=0)).sum()
    diff = (cp[1:] - cp[:-1]).square().sum()
    loss = diff + 10 * col + 10 * mm + 10 * jl
    (gl,) = torch.autograd.grad(loss, p)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        rec = R.optim.adam_traj_optimize(rob, md.rbf_score, start.double(), target.double(), dict(options))
    print(f"  multi-class adam record: success={rec['success']} cost={rec['cost']:.6f} cnt_check={rec['cnt_check']} "
          f"active at the initial path: {int(((s_init - margin.double()) > 0).sum())} of {s_init.numel()}")

    # ---- the scipy drivers' constraint on the same checker ----------------------------------------------------------------
    n_wp2 = 12
    t2 = torch.linspace(0, 1, n_wp2)[:, None].double()
    init2 = start.double() * (1 - t2) + target.double() * t2
    init2[1:-1] += 0.05 * torch.randn((n_wp2 - 2, 7), generator=gen).double()
    n_seg = n_wp2 - 1
    max_speed = None
    for ms in np.arange(0.30, 0.02, -0.0005):      # the first speed whose dense path has a multiple of n_seg inner points
        n_pt = len(R.utils.dense_path(init2, float(ms))) - 2
        if n_pt % n_seg == 0 and n_pt >= 2 * n_seg:
            max_speed = float(ms)
            break
    assert max_speed is not None
    dense = R.utils.dense_path(init2, max_speed)
    with torch.no_grad():
        s_dense = md.rbf_score(dense[1:-1])
    margin2 = s_dense.quantile(0.5, dim=0).float()

    def con(pp):  # optim.py:190-207 with return_tensor=True
        dense_p = R.utils.dense_path(pp, max_speed)
        cost = -(md.rbf_score(dense_p[1:-1]) - margin2.double())
        cost = torch.clamp(cost, max=0).reshape(-1)
        n_segment, n_point = len(pp) - 1, len(dense_p) - 2
        mult = n_point // n_segment
        if n_point % n_segment != 0:
            mult += 1
            cost = torch.cat([cost, torch.zeros(n_segment * mult - n_point, dtype=cost.dtype)])
        return cost.reshape(n_segment, -1).sum(dim=1)

    vvec = torch.rand(n_seg, generator=gen).double()
    p2 = init2.clone().requires_grad_(True)
    c0 = con(p2).detach()
    jac = torch.autograd.functional.jacobian(con, p2, create_graph=False, strict=False, vectorize=True, strategy="reverse-mode")
    jac = jac[:, 1:-1].reshape(jac.shape[0], -1)
    hess = torch.autograd.functional.hessian(lambda x: torch.dot(con(x), vvec), p2, create_graph=False, strict=False,
                                             vectorize=True, outer_jacobian_strategy="reverse-mode")
    hess = hess[1:-1, :, 1:-1, :].reshape((n_wp2 - 2) * 7, -1)
    print(f"  multi-class constraint: {len(dense) - 2} dense points over {n_seg} segments at max_speed {max_speed:.4f}, "
          f"{int((c0 < 0).sum())} rows active")
    save(out, "optim_multi_baxter", sup_q=sup_q, weights=Wn, start=start, target=target, init=init, margin=margin,
         score_init=s_init, loss0=loss.detach(), loss0_terms=torch.stack([diff, col, mm, jl]).detach(), grad0=gl,
         solution=np.array(rec["solution"]), cost=np.array(rec["cost"]), cnt_check=np.array(rec["cnt_check"]),
         success=np.array(rec["success"]), lr=np.array(0.05), maxiter=np.array(50), max_speed=np.array(0.3), seed=np.array(1234),
         init2=init2, margin2=margin2, max_speed2=np.array(max_speed), n_dense2=np.array(len(dense)), v2=vvec,
         con0=c0, jac0=jac, hess0=hess)


def gen_escape(out, robots):
    """Row f2's escape variant: the reference's OptimSampler.optim_escape (scripts/escape.py:19-38), imported
    from where it lies and run unmodified on reference checkers, in the three call patterns the reference's scripts use:
    scripts/2d_escape.py:98-110 (record every step, per-class margins), scripts/compare_sampling.py:177-195 (three Adam steps
    with wrap2pi on one random configuration at a time, last configuration only) and a batch that is ONE loop."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_escape", f"{REF}/scripts/escape.py")
    esc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(esc)
    gen = torch.Generator().manual_seed(1700)
    arrs = {}

    def run(tag, rob, dist_est, starts, args, one_loop):
        """one_loop: `starts` is one start_cfg; otherwise every row is its own call with a [1, dof] start"""
        sampler = esc.OptimSampler(rob, dist_est, dict(args))
        if one_loop:
            h, n = sampler.optim_escape(starts.clone())
            arrs[tag + "_hist"], arrs[tag + "_checks"] = h.detach(), np.array(n)
            print(f"  {tag}: {n} checks, {len(h)} records")
            return
        finals, checks, hists = [], [], []
        for b in range(len(starts)):
            h, n = sampler.optim_escape(starts[b:b + 1].clone())
            finals.append(h[-1, 0].detach()); checks.append(n); hists.append(h[:, 0].detach())
        width = max(len(h) for h in hists)
        arrs[tag + "_final"], arrs[tag + "_checks"] = torch.stack(finals), np.array(checks)
        arrs[tag + "_nrec"] = np.array([len(h) for h in hists])
        arrs[tag + "_hist"] = torch.stack([torch.cat([h, h[-1:].expand(width - len(h), -1)]) for h in hists], dim=1)
        print(f"  {tag}: checks {checks}")

    # ---- Baxter, Polyharmonic(1, 1) spline score (new API poly_score), C = 1 ---------------------------------------
    rob = robots["baxter_left"]
    sup_q = rand_cfgs(rob, 200, gen)
    w = torch.randn(200, generator=gen) * 0.05 + 0.004
    dc = new_diffco(rob, "poly", (1, 1.0), sup_q, w, "poly")
    starts = rand_cfgs(rob, 12, gen)
    s0 = dc.poly_score(starts).detach()
    arrs.update(bx_sup_q=sup_q, bx_w=w, bx_starts=starts, bx_score0=s0)
    m1 = float(s0[0]) - 0.12            # a margin the first start reaches after some steps
    arrs["bx_margin1"] = np.array(m1, dtype=np.float32)
    run("bx_single", rob, dc.poly_score, starts[:1], {"N_WAYPOINTS": 20, "safety_margin": m1, "lr": 5e-2, "record_freq": 1}, True)
    run("bx_flat", rob, dc.poly_score, starts[0], {"N_WAYPOINTS": 20, "safety_margin": m1, "lr": 5e-2, "record_freq": 3}, True)
    m3 = float(s0[:3].mean()) - 1e3      # never reached: the loop runs out of steps
    arrs["bx_margin3"] = np.array(m3, dtype=np.float32)
    run("bx_three", rob, dc.poly_score, starts[:3], {"N_WAYPOINTS": 12, "safety_margin": m3, "lr": 2e-2, "record_freq": 2,
                                                      "opt_args": {"lr": 2e-2, "betas": (0.8, 0.99), "eps": 1e-6}}, True)
    mb = float(s0.median())
    arrs["bx_marginb"] = np.array(mb, dtype=np.float32)
    run("bx_batch", rob, dc.poly_score, starts, {"N_WAYPOINTS": 15, "safety_margin": mb - 0.05, "lr": 5e-2, "record_freq": 4}, False)

    # ---- planar 3-link arm, three classes (old MultiDiffCo.rbf_score), per-class margins, wrap2pi: compare_sampling's options
    rob = robots["planar3"]
    S, C = 150, 3
    sup_q = rand_cfgs(rob, S, gen)
    W = torch.randn((S, C), generator=gen) * 0.1 + 0.01
    md = R.old_MultiDiffCo.MultiDiffCo.__new__(R.old_MultiDiffCo.MultiDiffCo)
    md.fkine, md.support_points, md.support_fkine = rob.fkine, sup_q, fk32(rob, sup_q).reshape(S, -1)
    md.rbf_kernel, md.rbf_nodes, md.num_class = make_kernel("poly", (1, 1.0)), W, C
    starts = rand_cfgs(rob, 16, gen) * 1.5          # some start outside [-pi, pi): the wrap matters
    s0 = md.rbf_score(starts).detach()
    margin = s0.median(dim=0).values - 0.3
    arrs.update(pl_sup_q=sup_q, pl_w=W, pl_starts=starts, pl_score0=s0, pl_margin=margin)
    opts = {"N_WAYPOINTS": 3, "safety_margin": margin, "lr": 0.2, "record_freq": None, "post_transform": R.utils.wrap2pi,
            "optimizer": torch.optim.Adam}
    run("pl_batch", rob, md.rbf_score, starts, opts, False)
    run("pl_long", rob, md.rbf_score, starts[:6], dict(opts, N_WAYPOINTS=20, record_freq=1, lr=0.1), False)

    # ---- Baxter, five classes (config #3's shape: old MultiDiffCo.rbf_score, Polyharmonic nodes), per-class margins, wrap2pi
    rob = robots["baxter_left"]
    S, C = 400, 5
    sup_q = rand_cfgs(rob, S, gen)
    W = (torch.randn((S, C), generator=gen) * 0.05 + 0.002) * (torch.rand((S, C), generator=gen) >= 0.4)
    md5 = R.old_MultiDiffCo.MultiDiffCo.__new__(R.old_MultiDiffCo.MultiDiffCo)
    md5.fkine, md5.support_points, md5.support_fkine = rob.fkine, sup_q, fk32(rob, sup_q).reshape(S, -1)
    md5.rbf_kernel, md5.rbf_nodes, md5.num_class = make_kernel("poly", (1, 1.0)), W, C
    starts = rand_cfgs(rob, 10, gen)
    s0 = md5.rbf_score(starts).detach()
    margin = s0.median(dim=0).values - 0.05
    arrs.update(b5_sup_q=sup_q, b5_w=W, b5_starts=starts, b5_score0=s0, b5_margin=margin)
    o5 = {"N_WAYPOINTS": 10, "safety_margin": margin, "lr": 5e-2, "record_freq": 2, "post_transform": R.utils.wrap2pi}
    run("b5_batch", rob, md5.rbf_score, starts, o5, False)
    run("b5_joint", rob, md5.rbf_score, starts[:4], dict(o5, N_WAYPOINTS=6, record_freq=1), True)

    # ---- SE(2) body, RQ perceptron score (new API score), se2_wrap2pi ------------------------------------------
    rob = robots["se2"]
    sup_q = rand_cfgs(rob, 120, gen)
    g = torch.randn(120, generator=gen) * 0.5 + 0.2
    dc2 = new_diffco(rob, "rq", (0.02, 2), sup_q, g, "score")   # a kernel as wide as the workspace (limits +-10)
    starts = rand_cfgs(rob, 8, gen)
    starts[:, 2] *= 1.8
    s0 = dc2.score(starts).detach()
    arrs.update(se2_sup_q=sup_q, se2_w=g, se2_starts=starts, se2_score0=s0, se2_kparams=np.array([0.02, 2.0]))
    print("  se2 scores at the starts:", [round(float(v), 3) for v in s0])
    run("se2_batch", rob, dc2.score, starts, {"N_WAYPOINTS": 10, "safety_margin": float(s0.median()) - 0.5, "lr": 0.1,
                                              "record_freq": 3, "post_transform": R.utils.se2_wrap2pi}, False)
    arrs["se2_margin"] = np.array(float(s0.median()) - 0.5, dtype=np.float32)
    save(out, "escape", **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default=None, help="run one generator only (e.g. old_single); MANIFEST.json is rewritten")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    robots = make_robots()
    if args.only in (None, "all"):
        print("FK");        gen_fk(out, robots)
        print("kernels");   gen_kernels(out)
        print("scores");    gen_scores(out, robots)
        print("edges");     gen_edges(out, robots)
        print("trained");   gen_trained(out, robots)
        print("optim");     gen_optim(out, robots)
    if args.only in (None, "all", "old_single"):   # last: the earlier fixtures do not depend on it
        print("old single-class API"); gen_old_single(out, robots)
    if args.only in (None, "all", "hess"):
        print("second derivatives"); gen_hess(out, robots)
    if args.only in (None, "all", "optim_scipy"):
        print("SLSQP / trust-constr drivers and their collision constraint"); gen_optim_scipy(out, robots)
    if args.only in (None, "all", "frames"):
        print("DH2mat / euler2mat"); gen_frames(out, robots)
    if args.only in (None, "all", "optim_multi"):
        print("Adam loop and scipy constraint on a multi-class checker"); gen_optim_multi(out, robots)
    if args.only in (None, "all", "escape"):
        print("escape loops (scripts/escape.py)"); gen_escape(out, robots)
    with open(os.path.join(out, "MANIFEST.json"), "w") as f:
        json.dump({"generator": "tools/make_golden.py", "torch": torch.__version__, "numpy": np.__version__,
                   "reference": "ucsdarclab/diffco @ /root/reference (2025-03-21)",
                   "files": sorted(x for x in os.listdir(out) if x.endswith((".npz", ".json")))}, f, indent=1)


if __name__ == "__main__":
    main()
