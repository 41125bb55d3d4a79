"""GPU tests of the optimisers' collision term on a MULTI-CLASS checker (round 6; SURVEY.md §8 rows f2 / f4 for config #3's
model): `dcx_score_hinge_grad_mc`, `dcx_traj_adam_run_mc` (persistent two-sweep kernel and three-launch loop), the drop-in and
fused Adam optimisers and the scipy drivers' constraint terms, pinned to `tests/golden/optim_multi_baxter.npz` - the record of
the REFERENCE's adam_traj_optimize on a reference MultiDiffCo.rbf_score with a [C] safety margin, and the reference's
con_collision_free / Jacobian / Hessian on the same checker (tools/make_golden.py gen_optim_multi)."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import load, make_robot, relerr

pytestmark = pytest.mark.gpu


def _model(d, rob):
    from diffco_amd import _ops
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, torch.from_numpy(d["sup_q"]).cuda()).reshape(len(d["sup_q"]), -1)
    return _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, torch.from_numpy(d["weights"]).cuda())


def _multi(d, rob):
    """old-API MultiDiffCo over the fixture's supports, nodes set directly (what fit_poly would leave behind)"""
    from diffco_amd import kernel
    from diffco_amd.deprecated import MultiDiffCo
    md = MultiDiffCo(None, kernel_func=kernel.FKKernel(rob.fkine, kernel.RQKernel(10.0)))
    md.fkine = rob.fkine
    md.num_class = d["weights"].shape[1]
    md.support_points = torch.from_numpy(d["sup_q"])
    md.support_fkine = rob.fkine(md.support_points).reshape(len(md.support_points), -1)
    md.rbf_kernel, md.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["weights"])
    return md


def test_multiclass_hinge_gradient_is_the_masked_upstream_sweep():
    d = load("optim_multi_baxter")
    rob = make_robot("baxter_left")
    m = _model(d, rob)
    g = torch.Generator().manual_seed(5)
    lim = rob.limits
    for B in (50, 777, 5000):     # split launch, a few tiles, many tiles
        q = (torch.rand((B, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
        s = m.score_raw(q)
        margin = s.median(dim=0).values
        for weight in (10.0, -1.0):
            sh, gh = m.score_hinge_grad_raw(q, margin, weight)
            assert torch.equal(sh, s)
            up = ((s - margin) > 0).float() * weight
            _, gu = m.score_grad_raw(q, up, want_score=False)
            assert torch.equal(gh, gu)                                   # the same sweep with the same upstream: identical bits
            # and a float64 restatement of d/dq [weight * sum_c clamp(score_c - margin_c, 0)] through the full Jacobian
            _, jac = m.score_jac_raw(q)
            ref = (up.double()[:, :, None] * jac.double()).sum(dim=1)
            assert relerr(gh.cpu().numpy(), ref.cpu().numpy()) < 2e-6
            idle = (up == 0).all(dim=1)
            assert bool(idle.any()) and float(gh[idle].abs().max()) == 0.0   # no class over its margin: exactly zero
    # one margin for all classes (the reference's broadcast), and the wrong number of margins
    sh, gh = m.score_hinge_grad_raw(q, 0.0, 1.0)
    _, gu = m.score_grad_raw(q, (s > 0).float(), want_score=False)
    assert torch.equal(gh, gu)
    with pytest.raises(ValueError):
        m.score_hinge_grad_raw(q, [0.0, 0.1], 1.0)


def test_multiclass_loss_terms_and_first_step_against_the_reference():
    """one iteration through dcx_traj_adam_run_mc at the fixture's initial path: the reference's loss terms (optim.py:88-99 on the
    reference MultiDiffCo in float64), and Adam's first step = lr * sign(reference gradient) wherever it is not tiny"""
    from diffco_amd import _lib
    from test_gpu_traj import _traj_state
    d = load("optim_multi_baxter")
    rob = make_robot("baxter_left")
    m = _model(d, rob)
    lib = _lib.require_gpu()
    init = torch.from_numpy(d["init"])
    s = m.score_raw(init.float().cuda())
    assert relerr(s.cpu().numpy(), d["score_init"]) < 1e-5
    lr = float(d["lr"])
    for fused in (1, 0):
        lib.dcx_debug_set(b"traj_fused", fused)
        try:
            st, bufs = _traj_state(m, rob, init[None].float())
            bufs["col_score"] = torch.zeros(init.shape[0] * m.C, device=m.dev)   # [R*W, C] for several classes
            st.col_score = C.c_void_p(bufs["col_score"].data_ptr())
            opt = _lib.TrajOpts(lr, 0.9, 0.999, 1e-8, 1, 10, 10, 10, 0.0, float(d["max_speed"]), 1e-2, 1e-4)
            _lib.check(lib.dcx_traj_adam_run_mc(m._h, C.byref(st), C.byref(opt), m.margins(d["margin"]), 1, 1,
                                                C.c_void_p(torch.cuda.current_stream(m.dev).cuda_stream)))
            torch.cuda.synchronize()
        finally:
            lib.dcx_debug_set(b"traj_fused", -1)
        diff, col, mm, jl = (float(v) for v in d["loss0_terms"])
        stats = bufs["stats"][0].cpu().double().numpy()
        assert abs(stats[0] - float(d["loss0"])) < 2e-5 * abs(float(d["loss0"]))
        assert relerr(stats[[1, 4, 5, 6]], np.array([diff, col, mm, jl])) < 2e-5 and col > 0
        g = torch.from_numpy(d["grad0"]).clone()
        g[[0, -1]] = 0
        big = g.abs() > 1e-3 * g.abs().max()
        moved = bufs["path"][0].cpu().double() - init
        assert float((moved + lr * torch.sign(g))[big].abs().max()) < 1e-5
        assert abs(stats[3] - float(g.norm())) < 1e-4 * float(g.norm())


def test_multiclass_fused_and_dropin_optimisers_reproduce_the_reference_record():
    from diffco_amd import fused_adam_traj_optimize, optim
    d = load("optim_multi_baxter")
    rob = make_robot("baxter_left")
    md = _multi(d, rob)
    start, target = torch.from_numpy(d["start"]), torch.from_numpy(d["target"])
    options = {"N_WAYPOINTS": len(d["init"]), "NUM_RE_TRIALS": 1, "MAXITER": int(d["maxiter"]),
               "safety_margin": torch.from_numpy(d["margin"]), "max_speed": float(d["max_speed"]), "seed": int(d["seed"]),
               "history": False, "extra_optimizer_options": {"lr": float(d["lr"])},
               "init_solution": torch.from_numpy(d["init"]).clone()}
    rec = fused_adam_traj_optimize(rob, md.rbf_score, start, target, dict(options))
    assert rec["success"] == bool(d["success"]) and rec["cnt_check"] == int(d["cnt_check"])
    assert abs(rec["cost"] - float(d["cost"])) < 5e-3 * float(d["cost"])
    assert relerr(np.array(rec["solution"]), d["solution"]) < 5e-3
    # the drop-in loop (torch's autograd around the HIP score: forward = the class scores, backward = the upstream sweep)
    rec2 = optim.adam_traj_optimize(rob, md.rbf_score, start, target, dict(options))
    assert rec2["success"] == bool(d["success"]) and rec2["cnt_check"] == int(d["cnt_check"])
    assert abs(rec2["cost"] - float(d["cost"])) < 5e-3 * float(d["cost"])
    assert relerr(np.array(rec2["solution"]), d["solution"]) < 5e-3
    assert relerr(np.array(rec["solution"]), np.array(rec2["solution"])) < 2e-3


@pytest.mark.parametrize("quantile", [0.6, 0.97, 0.999, 2.0])
@pytest.mark.parametrize("kind,C_,R,W,iters,S,ys", [(1, 5, 7, 20, 40, 500, 1), (1, 5, 256, 50, 30, 2000, 1), (0, 5, 9, 33, 25, 800, 1),
                                                    (1, 3, 6, 50, 25, 600, 1), (1, 2, 5, 64, 20, 400, 1), (0, 8, 4, 30, 20, 640, 1),
                                                    (1, 5, 32, 50, 40, 2000, 8), (0, 5, 20, 40, 25, 1000, 4), (1, 3, 6, 50, 25, 600, 2)])
def test_multiclass_persistent_launch_is_bit_identical_to_the_three_launch_loop(kind, C_, R, W, iters, S, ys, quantile, knob):
    """the two-sweep persistent kernel (traj_fused.h, CC > 1; cluster form for ys > 1) against the loop of {class scores,
    hinge-gradient sweep, step} launches sliced the same way: every output bit-identical - paths, moments, loss terms, records,
    stop flags - for Polyharmonic(1) (expanded form) and RQKernel(p = 2) models of 2, 3 (both run as 4), 5 and 8 classes.  The margins
    sit at a quantile of the initial scores: 0.6 (indicators flip every iteration: two sweeps), 0.97 / 0.999 (few entries over their
    margin: the kernel's speculation on the last iteration's indicator mostly holds, sometimes fails; whole paths without an active
    class skip the gradient sweep), 2.0 = above every score (free space: score sweeps only) - the results never depend on which
    of those routes an iteration took"""
    from diffco_amd import _lib, _ops
    from test_gpu_traj import _random_paths, _traj_state
    rob = make_robot("baxter_left")
    lib = _lib.require_gpu()
    g = torch.Generator().manual_seed(S + C_ + ys)
    lim = rob.limits
    sup_q = torch.rand((S, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
    Wn = 0.02 * torch.randn((S, C_), generator=g) * (torch.rand((S, C_), generator=g) >= 0.4)
    model = _ops.ScoreModel(desc, kind, (1.0 if kind == 1 else 10.0), (1.0 if kind == 1 else 2.0), sup, Wn.cuda())
    paths = _random_paths(rob, R, W, seed=R * W + C_)
    s0 = model.score_raw(paths.reshape(-1, rob.dof).cuda())
    margin = s0.quantile(quantile, dim=0) if quantile <= 1.0 else s0.max(dim=0).values + 10.0
    opt = _lib.TrajOpts(0.02, 0.9, 0.999, 1e-8, 1, 10, 10, 10, 0.0, 0.3, 1e9, 0.35)
    outs = []
    # (equal slicing: eight classes = 20 accumulators per lane, whose 16 partial rows do not fit a sweep block's 64 KB - the sweep
    # kernel would pick its own block size there; 8 waves fit both forms)
    knob("nw", 8 if C_ == 8 else 16)
    knob("ys", ys)
    knob("traj_ys", ys)
    if kind == 0:
        knob("xf", 0)    # (the persistent kernel keeps RQ models in the direct form; the sweep kernel's rule may expand them)
    for fused in (0, 1):
        knob("traj_fused", fused)
        st, bufs = _traj_state(model, rob, paths)
        bufs["col_score"] = torch.zeros(R * W * C_, device=model.dev)
        st.col_score = C.c_void_p(bufs["col_score"].data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(model.dev).cuda_stream)
        mg = model.margins(margin)
        _lib.check(lib.dcx_traj_adam_run_mc(model._h, C.byref(st), C.byref(opt), mg, 1, iters - 7, stream))
        _lib.check(lib.dcx_traj_adam_run_mc(model._h, C.byref(st), C.byref(opt), mg, iters - 6, 7, stream))  # resumes mid-run
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in bufs.items() if k not in ("col_score", "col_grad", "limits")})
    a, b = outs
    assert float(b["stats"][:, 7].min()) == 0.0
    assert int(a["steps"].min()) >= 1 and int(a["steps"].max()) == iters
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))
    assert float((a["path"].cpu() - paths).abs().max()) > 1e-3
    if quantile <= 0.6:
        assert float(a["stats"][:, 4].max()) > 0      # the collision term was active somewhere at the last step
    if quantile > 1.0:
        assert float(a["stats"][:, 4].max()) == 0.0
    assert torch.equal(a["path"][:, 0].cpu(), paths[:, 0]) and torch.equal(a["path"][:, -1].cpu(), paths[:, -1])


def test_multiclass_loop_on_widths_and_kernels_without_a_persistent_form():
    """a generic kernel function (MultiQuadratic) and a 27-wide URDF tree have no two-sweep persistent kernel: the call runs
    the three-launch loop by itself; one step against a float64 torch restatement built on the (verified) HIP score ops"""
    from diffco_amd import _lib, _ops
    from helpers import urdf_robot
    from test_gpu_traj import _random_paths, _traj_state
    lib = _lib.require_gpu()
    for rob, kind, p0, p1 in ((make_robot("baxter_left"), 2, 0.7, 0.0), (urdf_robot("urdf_panda"), 1, 1.0, 1.0)):
        g = torch.Generator().manual_seed(3)
        lim = rob.limits
        S, C_, R, W = 300, 3, 4, 20
        sup_q = torch.rand((S, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
        desc = rob.fk_desc()
        sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
        model = _ops.ScoreModel(desc, kind, p0, p1, sup, (0.02 * torch.randn((S, C_), generator=g)).cuda())
        paths = _random_paths(rob, R, W, seed=17)
        flat = paths.reshape(-1, rob.dof).cuda()
        s0 = model.score_raw(flat)
        margin = s0.quantile(0.5, dim=0)
        st, bufs = _traj_state(model, rob, paths)
        bufs["col_score"] = torch.zeros(R * W * C_, device=model.dev)
        st.col_score = C.c_void_p(bufs["col_score"].data_ptr())
        lr = 0.02
        opt = _lib.TrajOpts(lr, 0.9, 0.999, 1e-8, 1, 10, 10, 10, 0.0, 0.3, 1e9, 0.0)
        _lib.check(lib.dcx_traj_adam_run_mc(model._h, C.byref(st), C.byref(opt), model.margins(margin), 1, 1,
                                            C.c_void_p(torch.cuda.current_stream(model.dev).cuda_stream)))
        torch.cuda.synchronize()
        col = torch.clamp(s0.double() - margin.double(), min=0).reshape(R, W * C_).sum(dim=1)
        assert relerr(bufs["stats"][:, 4].cpu().double().numpy(), col.cpu().numpy()) < 1e-5
        # (the loop's score sweep is sliced like its gradient sweep - bit-identical to the persistent kernel where there is one -,
        # dcx_score has its own slices: the same sums in another order)
        assert relerr(bufs["col_score"].reshape(R * W, C_).cpu().numpy(), s0.cpu().numpy()) < 1e-6


def test_multiclass_scipy_constraint_terms_against_the_reference_fixture():
    """row f4 on a multi-class checker: constraint values, the Jacobian (one dcx_score_jac launch, masked by the hinge, entries
    grouped as the reference's flat reshape groups them) and the Hessian of v . c (one dcx_score_hess launch with the per-entry
    upstream) against the reference's con_collision_free and its autograd derivatives (optim.py:190-218, 380-391)"""
    from diffco_amd import optim
    d = load("optim_multi_baxter")
    rob = make_robot("baxter_left")
    md = _multi(d, rob)
    start, target, init = (torch.from_numpy(d[k]).double() for k in ("start", "target", "init2"))
    opts = {"N_WAYPOINTS": len(init), "NUM_RE_TRIALS": 1, "MAXITER": 5, "safety_margin": torch.from_numpy(d["margin2"]),
            "max_speed": float(d["max_speed2"]), "seed": 1, "history": False, "init_solution": init.clone()}
    prob = optim._PathProblem(rob, start, target, dict(opts))
    prob.make_init(0)
    terms = optim._ScipyTerms(prob, md.rbf_score)
    assert terms._fused_model() is not None and terms._fused_model().C == 5
    x = prob.init_path[1:-1].reshape(-1).numpy()
    n_dense = int(d["n_dense2"])
    c = terms.collision(x)
    assert relerr(c, d["con0"]) < 1e-5 and prob.cnt_check == n_dense and float(np.abs(d["con0"]).max()) > 0
    J = terms.jac_collision(x)
    assert relerr(J, d["jac0"]) < 1e-5 and prob.cnt_check == 2 * n_dense
    H = terms.hess_collision(x, d["v2"])
    assert relerr(H, d["hess0"]) < 5e-5 and prob.cnt_check == 3 * n_dense
    # the autograd route (a foreign callable with the same values) gives the same Jacobian
    terms2 = optim._ScipyTerms(prob, lambda q: md.rbf_score(q))
    assert terms2._fused_model() is None
    assert relerr(terms2.jac_collision(x), J) < 1e-5
    # a dense path whose point count does not divide: the reference's reshape does not exist there either
    prob3 = optim._PathProblem(rob, start, target, dict(opts, max_speed=0.2))
    prob3.make_init(0)
    n_pt = len(optim.utils.dense_path(prob3.init_path, 0.2)) - 2
    if (n_pt * 5 + ((-n_pt) % (len(init) - 1))) % (len(init) - 1):
        with pytest.raises(RuntimeError):
            optim._ScipyTerms(prob3, md.rbf_score).jac_collision(x)
