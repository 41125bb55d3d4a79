"""GPU tests of the hinge-gradient entry point and the fused Adam trajectory step, against a
float64 torch restatement of the reference's loop body (optim.py:86-127) and the golden Adam record."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, TorchDHRobot, TorchKernel, load, make_robot, relerr

pytestmark = pytest.mark.gpu


def _setup():
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    d = load("optim_adam_baxter")
    rob = make_robot("baxter_left")
    dc = DiffCo(transform=rob.fkine)
    dc.support_points = torch.from_numpy(d["sup_q"])
    dc.support_transformed = rob.fkine(dc.support_points)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["weights"])
    return d, rob, dc


def test_hinge_gradient_equals_masked_gradient():
    from diffco_amd import traj
    d, rob, dc = _setup()
    m = traj._resolve_model(dc.poly_score)
    g = torch.Generator().manual_seed(5)
    lim = rob.limits
    q = (torch.rand((777, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    s, gr = m.score_grad_raw(q)
    for margin, weight in ((0.0, 10.0), (float(s.median()), 2.5)):
        sh, gh = m.score_hinge_grad_raw(q, margin, weight)
        assert torch.equal(sh, s)
        mask = ((s - margin) > 0).float() * weight
        # the weight scales the feature gradient BEFORE J^T here and the joint gradient AFTER it there: equal up
        # to fp32 rounding, and exactly zero wherever the hinge is inactive
        assert relerr(gh.cpu().numpy(), (gr * mask).cpu().numpy()) < 1e-6
        assert float(gh[(mask == 0).reshape(-1)].abs().max()) == 0.0
    # both launch geometries (split / unsplit) agree
    sh2, gh2 = m.score_hinge_grad_raw(q[:100].contiguous(), 0.0, 10.0)
    assert relerr(gh2.cpu().numpy(), (gr * ((s > 0).float() * 10.0))[:100].cpu().numpy()) < 3e-6


def _torch_step(rob64, dist_est, p, m, v, t, lr, margin, max_speed):
    """one iteration of optim.py:86-103 in float64 torch; returns new p, m, v and the loss terms"""
    p = p.clone().requires_grad_(True)
    col = torch.clamp(dist_est(p) - margin, min=0).sum()
    cp = rob64.fkine(p)
    mm = torch.clamp((cp[1:] - cp[:-1]).square().sum(dim=2) - max_speed ** 2, min=0).sum()
    lim = rob64.limits.double()
    jl = (torch.clamp(lim[:, 0] - p, min=0) + torch.clamp(p - lim[:, 1], min=0)).sum()
    diff = (cp[1:] - cp[:-1]).square().sum()
    loss = diff + 10 * col + 10 * mm + 10 * jl
    (g,) = torch.autograd.grad(loss, p)
    g[[0, -1]] = 0.0
    m = 0.9 * m + 0.1 * g
    v = 0.999 * v + 0.001 * g * g
    denom = v.sqrt() / np.sqrt(1 - 0.999 ** t) + 1e-8
    pn = p.detach() - lr / (1 - 0.9 ** t) * m / denom
    return pn, m, v, torch.stack([loss, diff, 10 * col + 10 * mm + 10 * jl, g.norm(), col, mm, jl]).detach()


def test_single_adam_step_matches_float64_restatement():
    import ctypes as C
    from diffco_amd import _lib, traj
    d, rob, dc = _setup()
    model = traj._resolve_model(dc.poly_score)
    lib = _lib.require_gpu()
    rob64 = TorchDHRobot(rob)
    sup = rob64.fkine(torch.from_numpy(d["sup_q"]).double()).reshape(len(d["sup_q"]), -1)
    w = torch.from_numpy(d["weights"]).double()
    kern = TorchKernel("poly1", 1, 1.0)
    dist_est = lambda p: kern(rob64.fkine(p).reshape(len(p), -1), sup) @ w[:, None]
    # three paths: the golden init, a perturbed copy pushed outside the joint limits, and a long-stride one
    g = torch.Generator().manual_seed(9)
    init = torch.from_numpy(d["init"])
    p1 = init.clone()
    p1[5:9] += 2.0 * torch.randn((4, 7), generator=g).double()
    p2 = init + 0.4 * torch.randn(init.shape, generator=g).double()
    paths64 = torch.stack([init, p1, p2])
    R, W, dof = paths64.shape
    lr, margin, ms = 0.05, 0.02, 0.3
    dev = model.dev
    f32 = dict(device=dev, dtype=torch.float32)
    path = paths64.to(**f32).contiguous()
    am, av = torch.zeros_like(path), torch.zeros_like(path)
    bufs = dict(limits=rob.limits.to(**f32).contiguous(), col_score=torch.empty(R * W, **f32),
                col_grad=torch.empty((R * W, dof), **f32), stats=torch.zeros((R, 8), **f32),
                lowest_loss=torch.full((R,), float("inf"), **f32), lowest_obj=torch.full((R,), float("inf"), **f32),
                lowest_path=path.clone(), best_valid_obj=torch.full((R,), float("inf"), **f32),
                best_valid_path=path.clone(), done=torch.zeros(R, device=dev, dtype=torch.int32),
                steps=torch.zeros(R, device=dev, dtype=torch.int32))
    st = _lib.TrajState(R, W, *(C.c_void_p(t.data_ptr()) for t in (
        path, am, av, bufs["limits"], bufs["col_score"], bufs["col_grad"], bufs["stats"], bufs["lowest_loss"],
        bufs["lowest_obj"], bufs["lowest_path"], bufs["best_valid_obj"], bufs["best_valid_path"], bufs["done"],
        bufs["steps"])))
    opt = _lib.TrajOpts(lr, 0.9, 0.999, 1e-8, 1, 10, 10, 10, margin, ms, 1e-2, 1e-4)
    ref_p = [paths64[r].clone() for r in range(R)]
    ref_m = [torch.zeros(W, dof, dtype=torch.float64) for _ in range(R)]
    ref_v = [torch.zeros(W, dof, dtype=torch.float64) for _ in range(R)]
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for t in (1, 2, 3):
        _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), t, 1, stream))
        torch.cuda.synchronize()
        for r in range(R):
            ref_p[r], ref_m[r], ref_v[r], terms = _torch_step(rob64, dist_est, ref_p[r], ref_m[r], ref_v[r], t, lr, margin, ms)
            got = bufs["stats"][r, :7].cpu().double()
            assert relerr(got.numpy(), terms.numpy()) < 5e-5, (t, r, got, terms)
            assert relerr(path[r].cpu().numpy(), ref_p[r].numpy()) < 2e-5, (t, r)
            assert torch.equal(path[r, 0].cpu(), paths64[r, 0].float()) and torch.equal(path[r, -1].cpu(), paths64[r, -1].float())
    assert bufs["steps"].tolist() == [3, 3, 3]
    assert torch.isfinite(bufs["lowest_loss"]).all()


def test_fused_optimizer_reproduces_the_reference_record():
    from diffco_amd import fused_adam_traj_optimize, optim
    d, rob, dc = _setup()
    options = json.load(open(os.path.join(GOLDEN, "optim_adam_baxter_options.json")))
    options["init_solution"] = torch.from_numpy(d["init"]).clone()
    start, target = torch.from_numpy(d["start"]), torch.from_numpy(d["target"])
    rec = fused_adam_traj_optimize(rob, dc.poly_score, start, target, dict(options))
    assert rec["success"] == bool(d["success"]) and rec["cnt_check"] == int(d["cnt_check"])
    assert abs(rec["cost"] - float(d["cost"])) < 5e-3 * float(d["cost"])
    assert relerr(np.array(rec["solution"]), d["solution"]) < 5e-3
    # and agrees with the un-fused optimiser on the same HIP path
    rec2 = optim.adam_traj_optimize(rob, dc.poly_score, start, target, dict(options))
    assert relerr(np.array(rec["solution"]), np.array(rec2["solution"])) < 2e-3
    assert {"start_cfg", "target_cfg", "cnt_check", "cost", "time", "success", "seed", "solution"} <= set(rec)


def test_batched_restarts_follow_the_reference_policy():
    from diffco_amd import fused_adam_traj_optimize, optim
    d, rob, dc = _setup()
    start, target = torch.from_numpy(d["start"]), torch.from_numpy(d["target"])
    base = {"N_WAYPOINTS": 20, "NUM_RE_TRIALS": 6, "MAXITER": 60, "max_speed": 0.3, "seed": 77, "history": False,
            "extra_optimizer_options": {"lr": 0.05}}
    # an unreachable margin: no trial is ever valid -> the lowest loss over ALL trials is returned
    hard = dict(base, safety_margin=-1e3)
    a = fused_adam_traj_optimize(rob, dc.poly_score, start, target, dict(hard))
    b = optim.adam_traj_optimize(rob, dc.poly_score, start, target, dict(hard))
    assert not a["success"] and not b["success"]
    assert a["cnt_check"] == b["cnt_check"] == 6 * 60 * 20
    assert abs(a["cost"] - b["cost"]) < 2e-2 * abs(b["cost"]) and relerr(np.array(a["solution"]), np.array(b["solution"])) < 2e-2
    # a feasible margin: the first trial that becomes valid wins, like the sequential reference
    easy = dict(base, safety_margin=0.0)
    a = fused_adam_traj_optimize(rob, dc.poly_score, start, target, dict(easy))
    b = optim.adam_traj_optimize(rob, dc.poly_score, start, target, dict(easy))
    assert a["success"] == b["success"]
    if a["success"]:
        assert a["cnt_check"] == b["cnt_check"]
        assert abs(a["cost"] - b["cost"]) < 1e-2 * abs(b["cost"])


@pytest.mark.parametrize("robot_name,W", [("baxter_left", 130), ("planar3", 70), ("se3", 33)])
def test_adam_step_multi_wave_paths_and_other_transforms(robot_name, W):
    """paths longer than one wave (waypoints spread over several LDS slabs) and non-DH transforms: one step of the
    fused kernel against float64 autograd of the same loss built from torch ops on the HIP FK"""
    import ctypes as C
    from diffco_amd import _lib, _ops, kernel
    rob = make_robot(robot_name)
    lib = _lib.require_gpu()
    g = torch.Generator().manual_seed(W)
    lim = rob.limits
    dof = rob.dof
    S = 150
    sup_q = torch.rand((S, dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    w = 0.05 * torch.randn(S, generator=g)
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
    model = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, w.cuda())
    R = 3
    t = torch.linspace(0, 1, W)[None, :, None]
    a = torch.rand((R, 1, dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    b = torch.rand((R, 1, dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    paths = (a * (1 - t) + b * t + 0.05 * torch.randn((R, W, dof), generator=g)).float()
    paths[:, 3] = lim[:, 1] + 0.2   # one waypoint outside the joint limits
    dev = model.dev
    f32 = dict(device=dev, dtype=torch.float32)
    path = paths.to(**f32).contiguous()
    am, av = torch.zeros_like(path), torch.zeros_like(path)
    bufs = [rob.limits.to(**f32).contiguous(), torch.empty(R * W, **f32), torch.empty((R * W, dof), **f32),
            torch.zeros((R, 8), **f32), torch.full((R,), float("inf"), **f32), torch.full((R,), float("inf"), **f32),
            path.clone(), torch.full((R,), float("inf"), **f32), path.clone(),
            torch.zeros(R, device=dev, dtype=torch.int32), torch.zeros(R, device=dev, dtype=torch.int32)]
    st = _lib.TrajState(R, W, *(C.c_void_p(x.data_ptr()) for x in [path, am, av] + bufs))
    lr, margin, ms = 0.02, 0.01, 0.15
    opt = _lib.TrajOpts(lr, 0.9, 0.999, 1e-8, 1, 10, 10, 10, margin, ms, 1e-2, 1e-4)
    # float64 reference of the same step: torch ops on top of the (already verified) HIP FK and score ops
    p64 = paths.double().clone().requires_grad_(True)
    flat = p64.reshape(R * W, dof)
    col = torch.clamp(model.score(flat).reshape(R, W) - margin, min=0).sum(dim=1)
    cp = rob.fkine(flat).reshape(R, W, desc.n_points, desc.point_dim)
    seg = (cp[:, 1:] - cp[:, :-1]).square().sum(dim=3)
    mm = torch.clamp(seg - ms ** 2, min=0).sum(dim=(1, 2))
    l64 = rob.limits.double()
    jl = (torch.clamp(l64[:, 0] - p64, min=0) + torch.clamp(p64 - l64[:, 1], min=0)).sum(dim=(1, 2))
    diff = seg.sum(dim=(1, 2))
    loss = diff + 10 * col + 10 * mm + 10 * jl
    (gr,) = torch.autograd.grad(loss.sum(), p64)
    gr[:, 0] = 0
    gr[:, -1] = 0
    m1 = 0.1 * gr
    v1 = 0.001 * gr * gr
    ref = paths.double() - lr / 0.1 * m1 / (v1.sqrt() / np.sqrt(0.001) + 1e-8)
    _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), 1, 1, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize()
    stats = bufs[3].cpu().double()
    assert relerr(stats[:, 0].numpy(), loss.detach().numpy()) < 5e-5
    assert relerr(stats[:, 1].numpy(), diff.detach().numpy()) < 5e-5
    assert relerr(stats[:, 6].numpy(), jl.detach().numpy()) < 5e-5 and float(jl.detach().min()) > 0
    # Adam's first step is lr * sign(g) wherever |g| >> eps: compare where the gradient is not tiny
    big = gr.abs() > 1e-4 * gr.abs().max()
    assert float((path.cpu().double() - ref)[big].abs().max()) < 1e-5
    assert torch.equal(path[:, 0].cpu(), paths[:, 0]) and torch.equal(path[:, -1].cpu(), paths[:, -1])


def _traj_state(model, rob, paths, seed=0):
    """device buffers + ctypes state of R paths for dcx_traj_adam_run"""
    import ctypes as C
    from diffco_amd import _lib
    R, W, dof = paths.shape
    dev = model.dev
    f32 = dict(device=dev, dtype=torch.float32)
    path = paths.to(**f32).contiguous().clone()
    bufs = dict(path=path, adam_m=torch.zeros_like(path), adam_v=torch.zeros_like(path),
                limits=rob.limits.to(**f32).contiguous(), col_score=torch.zeros(R * W, **f32),
                col_grad=torch.zeros((R * W, dof), **f32), stats=torch.zeros((R, 8), **f32),
                lowest_loss=torch.full((R,), float("inf"), **f32), lowest_obj=torch.full((R,), float("inf"), **f32),
                lowest_path=path.clone(), best_valid_obj=torch.full((R,), float("inf"), **f32),
                best_valid_path=path.clone(), done=torch.zeros(R, device=dev, dtype=torch.int32),
                steps=torch.zeros(R, device=dev, dtype=torch.int32))
    st = _lib.TrajState(R, W, *(C.c_void_p(t.data_ptr()) for t in bufs.values()))
    return st, bufs


def _random_paths(rob, R, W, seed):
    g = torch.Generator().manual_seed(seed)
    lim = rob.limits
    t = torch.linspace(0, 1, W)[None, :, None]
    a = torch.rand((R, 1, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    b = torch.rand((R, 1, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    return (a * (1 - t) + b * t + 0.1 * torch.randn((R, W, rob.dof), generator=g)).float()


@pytest.mark.parametrize("robot_name,R,W,iters,S", [("baxter_left", 7, 20, 40, 500), ("baxter_left", 256, 50, 200, 2000),
                                                    ("planar3", 5, 64, 25, 300), ("se3", 4, 33, 25, 200),
                                                    ("urdf_panda", 6, 30, 30, 400)])
def test_persistent_launch_is_bit_identical_to_the_two_launch_loop(robot_name, R, W, iters, S, knob):
    """BASELINE config #5 (second case: its full size, 256 restarts x 50 waypoints x 200 iterations) as ONE persistent
    launch per <= 192 iterations (traj_fused.h) against the two-launches-per-iteration loop: with the same support
    slicing (16 waves, unsplit) every output — paths, Adam moments, loss terms, best-so-far records, stop flags — is
    bit-identical, DH arms, planar arms, rigid bodies and URDF trees alike"""
    import ctypes as C
    from diffco_amd import _lib, _ops
    from helpers import urdf_robot
    rob = urdf_robot(robot_name) if robot_name.startswith("urdf_") else make_robot(robot_name)
    lib = _lib.require_gpu()
    g = torch.Generator().manual_seed(S)
    lim = rob.limits
    sup_q = torch.rand((S, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
    model = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, (0.02 * torch.randn(S, generator=g)).cuda())
    paths = _random_paths(rob, R, W, seed=R * W)
    # margin near the median score so the hinge is active on about half of the waypoints; grad_tol large enough that
    # some paths stop early (exercises the per-path freeze)
    s0, _ = model.score_grad_raw(paths.reshape(-1, rob.dof).cuda())
    opt = _lib.TrajOpts(0.02, 0.9, 0.999, 1e-8, 1, 10, 10, 10, float(s0.median()), 0.3, 1e9, 0.35)
    outs = []
    knob("nw", 16)
    knob("ys", 1)
    knob("traj_ys", 1)
    for fused in (0, 1):
        knob("traj_fused", fused)
        st, bufs = _traj_state(model, rob, paths)
        stream = C.c_void_p(torch.cuda.current_stream(model.dev).cuda_stream)
        _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), 1, iters - 7, stream))
        _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), iters - 6, 7, stream))  # resumes mid-run
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in bufs.items() if k not in ("col_score", "col_grad", "limits")})
    a, b = outs
    assert int(a["steps"].min()) >= 1 and int(a["steps"].max()) == iters
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))
    assert float((a["path"].cpu() - paths).abs().max()) > 1e-3            # the paths moved
    assert torch.equal(a["path"][:, 0].cpu(), paths[:, 0]) and torch.equal(a["path"][:, -1].cpu(), paths[:, -1])
    if robot_name == "baxter_left" and R == 7:
        assert int(a["done"].sum()) >= 0


@pytest.mark.parametrize("robot_name,R,W,iters,S,ys", [("baxter_left", 32, 50, 200, 2000, 8), ("baxter_left", 7, 20, 40, 500, 2),
                                                       ("baxter_left", 64, 50, 30, 2000, 4), ("panda", 5, 33, 30, 1000, 4),
                                                       ("planar3", 5, 64, 25, 600, 2), ("se3", 4, 33, 25, 300, 2),
                                                       ("urdf_panda", 6, 30, 30, 1000, 4)])
def test_cluster_form_is_bit_identical_to_the_two_launch_loop(robot_name, R, W, iters, S, ys, knob):
    """the persistent launch with a path's supports split over ys workgroups (traj_fused.h, cluster form: config #5's
    8-GPU shard of 32 restarts on all 256 CUs) against the two-launch loop whose sweep is split the same way: same
    slices, same order of the sums -> every output bit-identical, early stops included"""
    import ctypes as C
    from diffco_amd import _lib, _ops
    from helpers import urdf_robot
    rob = urdf_robot(robot_name) if robot_name.startswith("urdf_") else make_robot(robot_name)
    lib = _lib.require_gpu()
    g = torch.Generator().manual_seed(S + ys)
    lim = rob.limits
    sup_q = torch.rand((S, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
    model = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, (0.02 * torch.randn(S, generator=g)).cuda())
    paths = _random_paths(rob, R, W, seed=R * W + ys)
    s0, _ = model.score_grad_raw(paths.reshape(-1, rob.dof).cuda())
    opt = _lib.TrajOpts(0.02, 0.9, 0.999, 1e-8, 1, 10, 10, 10, float(s0.median()), 0.3, 1e9, 0.35)
    outs = []
    knob("nw", 16)   # (clamped to the width's block-size ceiling by both forms)
    knob("ys", ys)
    knob("traj_ys", ys)
    for fused in (0, 1):
        knob("traj_fused", fused)
        st, bufs = _traj_state(model, rob, paths)
        stream = C.c_void_p(torch.cuda.current_stream(model.dev).cuda_stream)
        _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), 1, iters - 7, stream))
        _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), iters - 6, 7, stream))  # resumes mid-run
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in bufs.items() if k not in ("col_score", "col_grad", "limits")})
    a, b = outs
    assert float(b["stats"][:, 7].min()) == 0.0   # no exchange gave up
    assert int(a["steps"].min()) >= 1 and int(a["steps"].max()) == iters
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))
    assert float((a["path"].cpu() - paths).abs().max()) > 1e-3


def test_cluster_form_agrees_with_one_workgroup_per_path(knob):
    """the rule's choice for a 32-restart shard (8 workgroups per path) against one workgroup per path: a different order
    of the support sums, so not bitwise - the loss terms of the first iteration agree to fp32 rounding, the first Adam
    step (lr * sign(g) wherever |g| >> eps) moves the same waypoints the same way"""
    import ctypes as C
    from diffco_amd import _lib, _ops
    rob = make_robot("baxter_left")
    lib = _lib.require_gpu()
    g = torch.Generator().manual_seed(11)
    lim = rob.limits
    S, R, W, lr = 2000, 32, 50, 0.002
    sup_q = torch.rand((S, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
    model = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, (0.02 * torch.randn(S, generator=g)).cuda())
    paths = _random_paths(rob, R, W, seed=3)
    s0, _ = model.score_grad_raw(paths.reshape(-1, rob.dof).cuda())
    srt = s0.reshape(-1).sort().values
    margin = float(0.5 * (srt[len(srt) // 2] + srt[len(srt) // 2 + 1]))   # BETWEEN two scores: no waypoint sits on the hinge
    opt = _lib.TrajOpts(lr, 0.9, 0.999, 1e-8, 1, 10, 10, 10, margin, 0.3, 1e9, 0.0)
    outs = []
    for tys in (1, -1):
        knob("traj_ys", tys)
        st, bufs = _traj_state(model, rob, paths)
        stream = C.c_void_p(torch.cuda.current_stream(model.dev).cuda_stream)
        _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), 1, 1, stream))
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in bufs.items()})
    a, b = outs
    assert float(b["stats"][:, 7].min()) == 0.0
    assert relerr(b["stats"][:, :7].cpu().numpy(), a["stats"][:, :7].cpu().numpy()) < 5e-6
    assert relerr(b["stats"][:, 1].cpu().numpy(), a["stats"][:, 1].cpu().numpy()) < 1e-6   # the path length does not see the sweep
    moved = (a["path"] - b["path"]).abs().cpu()
    assert float(moved.max()) <= 2.0 * lr * 1.001              # a step is at most lr either way
    assert float((moved > 1e-6).float().mean()) < 1e-3         # and all but a few sign-of-a-tiny-gradient entries agree
    assert torch.equal(a["steps"], b["steps"])


@pytest.mark.parametrize("R,W,expect_cluster", [(256, 12, False), (129, 12, False), (128, 12, True), (37, 64, True), (3, 2 + 1, True), (1, 50, True)])
def test_cluster_rule_edges(R, W, expect_cluster, knob):
    """the rule that picks the workgroups per path (largest power of two <= min(8, CUs / R)): path counts around the
    boundaries, a path that fills its 64-lane tile, three-waypoint paths, a single path - each against the two-launch loop
    split the same way (bit-identical), and the rule's choice against one workgroup per path (same step counts, loss
    terms to rounding)"""
    import ctypes as C
    from diffco_amd import _lib, _ops
    rob = make_robot("baxter_left")
    lib = _lib.require_gpu()
    g = torch.Generator().manual_seed(R + W)
    lim = rob.limits
    S = 1200
    sup_q = torch.rand((S, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
    model = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, (0.02 * torch.randn(S, generator=g)).cuda())
    paths = _random_paths(rob, R, W, seed=R * W)
    s0, _ = model.score_grad_raw(paths.reshape(-1, rob.dof).cuda())
    srt = s0.reshape(-1).sort().values
    margin = float(0.5 * (srt[len(srt) // 2] + srt[len(srt) // 2 + 1])) if len(srt) > 2 else float(srt.mean())
    opt = _lib.TrajOpts(0.02, 0.9, 0.999, 1e-8, 1, 10, 10, 10, margin, 0.3, 1e9, 0.0)
    ys_rule = 1
    while 2 * ys_rule <= min(8, 256 // R) and S // (2 * ys_rule * 16) >= 15:
        ys_rule *= 2
    assert (ys_rule > 1) == expect_cluster
    outs = {}
    for label, fused, tys, ys in (("rule", 1, -1, None), ("two-launch", 0, 1, ys_rule), ("one", 1, 1, None)):
        knob("traj_fused", fused)
        knob("traj_ys", tys)
        knob("nw", 16)
        knob("ys", ys if ys is not None else -1)
        st, bufs = _traj_state(model, rob, paths)
        stream = C.c_void_p(torch.cuda.current_stream(model.dev).cuda_stream)
        _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), 1, 12, stream))
        torch.cuda.synchronize()
        outs[label] = {k: v.clone() for k, v in bufs.items() if k not in ("col_score", "col_grad", "limits")}
    a, b, c = outs["rule"], outs["two-launch"], outs["one"]
    assert float(a["stats"][:, 7].min()) == 0.0
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))
    assert torch.equal(a["steps"], c["steps"]) and int(a["steps"].min()) == 12
    assert torch.equal(a["path"][:, 0].cpu(), paths[:, 0]) and torch.equal(a["path"][:, -1].cpu(), paths[:, -1])
