"""GPU tests of the drop-in Python surface (DiffCo / MultiDiffCo / kernels / robots / optimisers): the
reference's call signatures and shape/dtype rules, with values checked against the golden vectors
generated from the reference.  Everything numeric here runs in libdcx.so (HIP)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load, make_robot, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _np(t):
    return t.detach().cpu().numpy()


def _new_diffco(rob, d, which, device="cpu"):
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    kinds = {"rq": lambda p: kernel.RQKernel(p[0], int(p[1])), "poly": lambda p: kernel.Polyharmonic(int(p[0]), p[1]),
             "mq": lambda p: kernel.MultiQuadratic(p[0])}
    kf = kinds[str(d["kind"])](d["kparams"])
    dc = DiffCo(kernel_func=kf if which == "score" else "rq", transform=None if rob is None else rob.fkine)
    dc.support_points = torch.from_numpy(d["sup_q"]).to(device)
    dc.support_transformed = torch.from_numpy(d["sup_x32"]).to(device)
    w = torch.from_numpy(d["weights"][:, 0].copy()).to(device)
    if which == "score":
        dc.gains = w
    else:
        dc.rbf_kernel, dc.rbf_nodes = kf, w
    return dc


@pytest.mark.parametrize("device", ["cpu", "cuda"])
def test_poly_score_and_autograd_match_reference(device):
    d = load("cfg2_baxter_poly1")
    dc = _new_diffco(make_robot("baxter_left"), d, "poly", device)
    q = torch.from_numpy(d["q"]).to(device).requires_grad_(True)
    s = dc.poly_score(q)
    assert s.shape == (4096, 1) and s.device.type == device and s.dtype == torch.float32
    (g,) = torch.autograd.grad(s.sum(), q)
    assert g.device.type == device
    assert relerr(_np(s), d["score64"]) < TOL and relerr(_np(g), d["grad64"]) < TOL
    assert relerr(_np(s), d["score32"]) < TOL + relerr(d["score32"], d["score64"])
    # a non-trivial upstream through autograd (clamp + weights), as the optimisers use it
    wts = torch.linspace(0.5, 2.0, 4096, device=device).reshape(-1, 1)
    (g2,) = torch.autograd.grad((torch.clamp(dc.poly_score(q) - 0.1, min=0) * wts).sum(), q)
    mask = (torch.from_numpy(d["score64"]).to(device) - 0.1 > 0).float() * wts
    assert relerr(_np(g2), _np(mask) * d["grad64"]) < 5e-5  # entries right at the clamp may flip
    with torch.no_grad():
        assert torch.allclose(dc.poly_score(q), s, atol=1e-6 * float(s.abs().max()))


def test_score_rq_and_is_collision():
    d = load("cfg2_baxter_rq")
    dc = _new_diffco(make_robot("baxter_left"), d, "score", "cuda")
    q = torch.from_numpy(d["q"]).cuda().requires_grad_(True)
    s = dc.score(q)
    assert s.shape == (512,)
    (g,) = torch.autograd.grad(s.sum(), q)
    assert relerr(_np(s), d["score64"].reshape(-1)) < TOL and relerr(_np(g), d["grad64"]) < TOL
    assert torch.equal(dc.is_collision(q.detach()), dc.score(q.detach()) > 0)
    assert torch.equal(dc(q.detach()), dc.score(q.detach()) > 0)


def test_line_queries_are_one_batch():
    """Perceptron.line_predict (kernel_perceptrons.py:22-24) and the old API's line_collision (deprecated/DiffCo.py:20-22):
    `any(is_collision(start + (target - start) i / res))`, asked as one batch - equal to the per-point loop the reference runs"""
    from diffco_amd import deprecated, kernel
    d = load("cfg2_baxter_rq")
    rob = make_robot("baxter_left")
    dc = _new_diffco(rob, d, "score", "cuda")
    q = torch.from_numpy(d["q"])
    s = dc.score(q)
    free, hit = q[int(torch.argmin(s))], q[int(torch.argmax(s))]
    for a, b, res in ((free, hit, 50), (hit, free, 7), (free, free, 3), (q[0], q[1], 50), (q[2], q[3], 50)):
        want = any(bool(dc.is_collision(a + (b - a) / res * i)) for i in range(res))
        assert dc.line_predict(a, b, res=res) is want
        assert dc.line_predict(a.cuda(), b.cuda(), res=res) is want
    assert dc.line_predict(hit, free, 7) is True
    old = deprecated.DiffCo(None, kernel_func=kernel.FKKernel(rob.fkine, kernel.RQKernel(10.0)), beta=1.0)
    old.train_method = "original"
    old.support_points, old.gains = torch.from_numpy(d["sup_q"]), torch.from_numpy(d["weights"]).reshape(-1)
    for a, b in ((free, hit), (q[0], q[1])):
        want = any(bool(old.is_collision(a + (b - a) / 50 * i)) for i in range(50))
        assert old.line_collision(a, b) is want


def test_planar_config1_on_cpu_tensors():
    """BASELINE config #1: 2-DoF planar arm, RQ(10), 200 supports, batch 256 — CPU tensors in and out,
    computed by the HIP path (the reference runs this case on PyTorch CPU)."""
    d = load("cfg1_planar2_rq")
    dc = _new_diffco(make_robot("planar2"), d, "score", "cpu")
    q = torch.from_numpy(d["q"]).requires_grad_(True)
    s = dc.score(q)
    (g,) = torch.autograd.grad(s.sum(), q)
    assert s.device.type == "cpu" and s.shape == (256,)
    assert relerr(_np(s), d["score64"].reshape(-1)) < TOL and relerr(_np(g), d["grad64"]) < TOL
    assert relerr(_np(s), d["score32"].reshape(-1)) < TOL + relerr(d["score32"].reshape(-1), d["score64"].reshape(-1))


def test_shape_and_dtype_quirks():
    """single-query squeeze, fp64 input to poly_score, transformed_point, zero padding"""
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    e = load("edges")
    rob = make_robot("baxter_left")
    sup_q, w = torch.from_numpy(e["sup_q"]), torch.from_numpy(e["weights"])
    dc = DiffCo(kernel_func=kernel.RQKernel(10.0), transform=rob.fkine)
    dc.support_points, dc.support_transformed, dc.gains = sup_q, rob.fkine(sup_q), w
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), w
    s1 = dc.score(torch.from_numpy(e["q1"]))
    assert tuple(s1.shape) == tuple(e["score_b1_shape"]) == ()
    assert relerr(_np(s1).reshape(-1), e["score_b1"]) < 2e-5
    p1 = dc.poly_score(torch.from_numpy(e["q1"]))
    assert tuple(p1.shape) == tuple(e["poly_b1_shape"]) == (1, 1)
    assert relerr(_np(p1).reshape(-1), e["poly_b1"]) < 2e-5
    qd = torch.from_numpy(e["q_f64"]).requires_grad_(True)
    s = dc.poly_score(qd)
    (g,) = torch.autograd.grad(s.sum(), qd)
    assert str(s.dtype) == str(e["poly_f64in_dtype"]) == "torch.float32"   # cast to the nodes' dtype
    assert str(g.dtype) == str(e["grad_f64in_dtype"]) == "torch.float64"   # gradient comes back in fp64
    assert relerr(_np(s), e["poly_f64in"]) < 2e-5 and relerr(_np(g), e["grad_f64in"]) < 2e-5
    tp = dc.poly_score(transformed_point=rob.fkine(torch.from_numpy(e["q_tp"])))
    assert tp.shape == (8, 1) and relerr(_np(tp), e["poly_tp"]) < 2e-5
    # gradient w.r.t. the transformed point flows too (collision_checkers.py:493 differentiates this form)
    X = rob.fkine(torch.from_numpy(e["q_tp"])).detach().requires_grad_(True)
    (gx,) = torch.autograd.grad(dc.poly_score(transformed_point=X).sum(), X)
    assert gx.shape == X.shape and float(gx.abs().max()) > 0
    # zero padding (max_num_supports)
    dz = DiffCo(transform=rob.fkine)
    dz.support_points = torch.cat([sup_q, torch.zeros(14, 7)])
    dz.support_transformed = torch.cat([rob.fkine(sup_q), torch.zeros(14, 4, 3)])
    dz.rbf_kernel, dz.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.cat([w, torch.zeros(14)])
    assert relerr(_np(dz.poly_score(torch.from_numpy(e["q_tp"]))), e["poly_padded"]) < 2e-5


def test_to_cuda_pickle_and_state_updates():
    d = load("cfg2_baxter_rq")
    rob = make_robot("baxter_left")
    dc = _new_diffco(rob, d, "score", "cpu")
    q = torch.from_numpy(d["q"][:100])
    s_cpu = dc.score(q)
    dc.to(torch.device("cuda"))
    assert dc.gains.device.type == "cuda" and dc.support_transformed.device.type == "cuda"  # gains move too (H3)
    s_gpu = dc.score(q.cuda())
    assert torch.allclose(s_gpu.cpu(), s_cpu, atol=1e-6 * float(s_cpu.abs().max()))
    dc.to("cpu")
    blob = pickle.dumps(dc)          # no device handle inside
    dc2 = pickle.loads(blob)
    assert torch.allclose(dc2.score(q), s_cpu, atol=1e-6 * float(s_cpu.abs().max()))
    dc2.gains = dc2.gains * 2        # state change -> the cached device model is rebuilt
    assert torch.allclose(dc2.score(q), 2 * s_cpu, atol=2e-6 * float(s_cpu.abs().max()))
    dc2.gains.mul_(0.5)              # in-place change is noticed as well
    assert torch.allclose(dc2.score(q), s_cpu, atol=2e-6 * float(s_cpu.abs().max()))


def test_training_on_the_hip_kernel_rows_reproduces_the_reference_model():
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    d = load("trained_baxter")
    rob = make_robot("baxter_left")
    dc = DiffCo(kernel_func=kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine)
    X, y, dist = (torch.from_numpy(d[k]) for k in ("X", "y", "dist"))
    dc.train(X, y, max_iteration=3000, distance=dist)
    np.testing.assert_array_equal(_np(dc.support_points), d["support_points"])
    assert relerr(_np(dc.gains), d["gains"]) < 1e-3 and relerr(_np(dc.hypothesis), d["hypothesis"]) < 1e-3
    dc.fit_poly(kernel.Polyharmonic(1, 1.0), target="label")
    assert relerr(_np(dc.rbf_nodes), d["rbf_nodes_label"]) < 2e-2  # ill-conditioned solve, see test_host_logic
    qt = torch.from_numpy(d["q_test"]).requires_grad_(True)
    s = dc.poly_score(qt)
    (g,) = torch.autograd.grad(s.sum(), qt)
    assert relerr(_np(s), d["poly_test"]) < 2e-3 and relerr(_np(g), d["poly_grad_test"]) < 5e-3
    assert relerr(_np(dc.score(qt.detach())), d["score_test"]) < 1e-3
    # with the reference's own nodes the spline agrees to kernel accuracy
    dc.rbf_nodes = torch.from_numpy(d["rbf_nodes_label"])
    dc.support_transformed = torch.from_numpy(d["support_transformed"])
    assert relerr(_np(dc.poly_score(qt.detach())), d["poly_test"]) < 5e-5
    # active-learning update with jump start (collision_checkers.py:220-252 call pattern)
    dc.fit_poly(kernel.Polyharmonic(1, 1.0), target="label")
    dc.train(torch.from_numpy(d["Xu"]), torch.from_numpy(d["yu"]), update=True,
             exist_mask=torch.from_numpy(d["exist_mask"]), max_iteration=2000, distance=torch.from_numpy(d["du"]))
    np.testing.assert_array_equal(_np(dc.support_points), d["upd_support_points"])
    assert relerr(_np(dc.gains), d["upd_gains"]) < 2e-3
    # fixed-size variant
    dm = DiffCo(kernel_func=kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine, max_num_supports=300)
    dm.train(X, y, max_iteration=3000, distance=dist)
    dm.fit_poly(kernel.Polyharmonic(1, 1.0), target="label")
    assert dm.valid_supports == int(d["mns_valid"])
    assert relerr(_np(dm.poly_score(qt.detach())), d["mns_poly_test"]) < 2e-3


def test_old_api_multidiffco():
    from diffco_amd import MultiDiffCo, kernel
    d = load("trained_multi_planar2")
    rob = make_robot("planar2")
    md = MultiDiffCo(None, kernel_func=kernel.FKKernel(rob.fkine, kernel.RQKernel(10.0)), beta=1.0)
    md.train(torch.from_numpy(d["X"]), torch.from_numpy(d["y"]), max_iteration=1500, distance=torch.from_numpy(d["dist"]))
    np.testing.assert_array_equal(_np(md.support_points), d["support_points"])
    assert relerr(_np(md.gains), d["gains"]) < 1e-3
    md.fit_poly(kernel_func=kernel.Polyharmonic(1, 1.0), target="label", fkine=rob.fkine, reg=0.0)
    qt = torch.from_numpy(d["q_test"]).requires_grad_(True)
    s = md.rbf_score(qt)
    assert s.shape == (256, 2)
    (g,) = torch.autograd.grad(s.sum(), qt)
    assert relerr(_np(s), d["rbf_test"]) < 5e-3 and relerr(_np(g), d["rbf_grad_test"]) < 1e-2
    assert relerr(_np(md.score(qt.detach())), d["score_test"]) < 1e-3
    # per-class margins broadcast against [N, C] scores (scripts/active.py:65 call pattern)
    margin = torch.tensor([0.1, -0.2])
    loss = torch.clamp(md.rbf_score(qt) - margin, min=0).sum()
    loss.backward()
    assert qt.grad is not None and torch.isfinite(qt.grad).all()
    # exact values with the reference's nodes: C=5 golden case through the old-API attributes
    c = load("cfg3_baxter_rq_c5")
    rb = make_robot("baxter_left")
    m5 = MultiDiffCo(None, kernel_func=kernel.FKKernel(rb.fkine, kernel.RQKernel(10.0)))
    m5.fkine, m5.support_points = rb.fkine, torch.from_numpy(c["sup_q"])
    m5.support_fkine = torch.from_numpy(c["sup_x32"]).reshape(2000, -1)
    m5.rbf_kernel, m5.rbf_nodes, m5.num_class = kernel.RQKernel(10.0), torch.from_numpy(c["weights"]), 5
    q5 = torch.from_numpy(c["q"]).requires_grad_(True)
    s5 = m5.rbf_score(q5)
    (gv,) = torch.autograd.grad((s5 * torch.from_numpy(c["upstream"])).sum(), q5)
    assert relerr(_np(s5), c["score64"]) < TOL and relerr(_np(gv), c["vjp64"]) < TOL


def test_full_jacobian_through_vmap():
    """torch.autograd.functional.jacobian(vectorize=True) — what SLSQP/trust-constr call (optim.py:211-216)"""
    d = load("cfg3_baxter_poly1_c5")
    from diffco_amd import MultiDiffCo, kernel
    rb = make_robot("baxter_left")
    m = MultiDiffCo(None)
    m.fkine, m.support_points = rb.fkine, torch.from_numpy(d["sup_q"])
    m.support_fkine = torch.from_numpy(d["sup_x32"]).reshape(2000, -1)
    m.rbf_kernel, m.rbf_nodes, m.num_class = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["weights"]), 5
    q = torch.from_numpy(d["q"][:8]).double()
    jac = torch.autograd.functional.jacobian(lambda x: m.rbf_score(x).sum(0), q, vectorize=True, strategy="reverse-mode")
    assert jac.shape == (5, 8, 7)
    assert relerr(_np(jac).transpose(1, 0, 2), d["jac32"][:8]) < 2e-5


def test_adam_and_slsqp_on_the_hip_path():
    from diffco_amd import kernel, optim
    from diffco_amd.kernel_perceptrons import DiffCo
    d = load("optim_adam_baxter")
    options = json.load(open(os.path.join(GOLDEN, "optim_adam_baxter_options.json")))
    rob = make_robot("baxter_left")
    dc = DiffCo(transform=rob.fkine)
    dc.support_points = torch.from_numpy(d["sup_q"])
    dc.support_transformed = rob.fkine(dc.support_points)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["weights"])
    # one evaluation of the Adam loss at the initial path: value and gradient vs the reference
    p = torch.from_numpy(d["init"]).clone().requires_grad_(True)
    col = torch.clamp(dc.poly_score(p), min=0).sum()
    cp = rob.fkine(p)
    mm = torch.clamp((cp[1:] - cp[:-1]).square().sum(dim=2) - 0.3 ** 2, min=0).sum()
    lim = rob.limits.double()
    jl = (torch.clamp(lim[:, 0] - p, min=0) + torch.clamp(p - lim[:, 1], min=0)).sum()
    diff = (cp[1:] - cp[:-1]).square().sum()
    loss = diff + 10 * col + 10 * mm + 10 * jl
    terms = torch.stack([diff, col, mm, jl]).detach().numpy()
    assert relerr(terms, d["loss0_terms"]) < 2e-5
    (g,) = torch.autograd.grad(loss, p)
    assert relerr(g.numpy(), d["grad0"]) < 2e-5
    assert float(g[:, 6].abs().max()) == 0.0  # Baxter's last joint moves no control point: EXACT zero, as autograd gives
    options["init_solution"] = torch.from_numpy(d["init"]).clone()
    rec = optim.adam_traj_optimize(rob, dc.poly_score, torch.from_numpy(d["start"]), torch.from_numpy(d["target"]),
                                   dict(options))
    assert rec["success"] == bool(d["success"]) and rec["cnt_check"] == int(d["cnt_check"])
    assert abs(rec["cost"] - float(d["cost"])) < 5e-3 * float(d["cost"])
    assert relerr(np.array(rec["solution"]), d["solution"]) < 5e-3
    # f4: the fused analytic constraint Jacobian (one hinge-gradient launch) equals autograd's vectorised Jacobian
    prob = optim._PathProblem(rob, torch.from_numpy(d["start"]), torch.from_numpy(d["target"]), dict(options, safety_margin=-50.0))
    prob.init_path = torch.from_numpy(d["init"]).clone()
    terms = optim._ScipyTerms(prob, dc.poly_score)
    x = prob.init_path[1:-1].reshape(-1).numpy()
    assert terms._fused_model() is not None
    Ja = terms.jac_collision(x)
    terms._model = None
    Jb = terms.jac_collision(x)
    assert Ja.shape == Jb.shape == (19, 18 * 7) and np.abs(Jb).max() > 0
    assert relerr(Ja, Jb) < 2e-5
    opts = dict(options, MAXITER=8, extra_optimizer_options={})
    rec2 = optim.givengrad_traj_optimize(rob, dc.poly_score, torch.from_numpy(d["start"]), torch.from_numpy(d["target"]), opts)
    assert np.isfinite(rec2["cost"]) and len(rec2["solution"]) == 20 and rec2["cnt_check"] > 0
    # trust-constr's constraint Hessian: the analytic per-point Hessians of dcx_score_hess chained through the dense
    # path vs a double backward through an fp64 torch restatement of the same score (the reference's route,
    # optim.py:380-391)
    from helpers import TorchDHRobot, TorchKernel
    trob, tk = TorchDHRobot(rob), TorchKernel("poly1", 1, 1.0)
    sup, w = trob.fkine(dc.support_points.double()), dc.rbf_nodes.double().reshape(-1, 1)
    ref_terms = optim._ScipyTerms(prob, lambda q: tk(trob.fkine(q), sup) @ w)
    ref_terms._model = None
    assert relerr(ref_terms.collision(x), terms.collision(x)) < 2e-5
    v = np.random.default_rng(0).standard_normal(19)
    del terms._model  # back to the fused route
    Ha, Hb = terms.hess_collision(x, v), ref_terms.hess_collision(x, v)
    assert Ha.shape == Hb.shape == (18 * 7, 18 * 7) and np.abs(Hb).max() > 0
    assert relerr(Ha, Hb) < 5e-5
    rec3 = optim.trustconstr_traj_optimize(rob, dc.poly_score, torch.from_numpy(d["start"]), torch.from_numpy(d["target"]),
                                           dict(opts, MAXITER=5))
    assert np.isfinite(rec3["cost"]) and len(rec3["solution"]) == 20
    with pytest.raises(ValueError):
        optim.trustconstr_traj_optimize(rob, lambda q: dc.poly_score(q), torch.from_numpy(d["start"]),
                                        torch.from_numpy(d["target"]), dict(opts, constraint_hessian="fused"))


def test_scipy_constraint_and_drivers_against_the_reference_fixture():
    """row f4 pinned to the REFERENCE (tools/make_golden.py gen_optim_scipy): the fused route of `_ScipyTerms` - constraint
    values from the HIP score, the Jacobian assembled from one hinge-gradient launch, the Hessian from dcx_score_hess -
    against the reference's con_collision_free / jac_con_collision_free / hess_con_collision_free at the initial path
    (optim.py:190-218, 380-391), and both scipy drivers against the records the reference's drivers produced
    (optim.py:166-321, 324-516: cost, solution, cnt_check)"""
    from diffco_amd import kernel, optim
    from diffco_amd.kernel_perceptrons import DiffCo
    d = load("optim_scipy_baxter")
    rob = make_robot("baxter_left")
    dc = DiffCo(transform=rob.fkine)
    dc.support_points = torch.from_numpy(d["sup_q"])
    dc.support_transformed = rob.fkine(dc.support_points)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["weights"])
    start, target, init = (torch.from_numpy(d[k]).double() for k in ("start", "target", "init"))
    opts = {"N_WAYPOINTS": len(init), "NUM_RE_TRIALS": 1, "MAXITER": int(d["slsqp_maxiter"]), "safety_margin": float(d["margin"]),
            "max_speed": float(d["max_speed"]), "seed": 4321, "history": False, "extra_optimizer_options": {"disp": False},
            "init_solution": init.clone()}
    prob = optim._PathProblem(rob, start, target, dict(opts))
    prob.make_init(0)
    terms = optim._ScipyTerms(prob, dc.poly_score)
    assert terms._fused_model() is not None
    x = prob.init_path[1:-1].reshape(-1).numpy()
    n_dense = int(d["n_dense"])
    c = terms.collision(x)
    assert relerr(c, d["con0_f64"]) < 1e-5 and relerr(c, d["con0_ref"]) < 2e-5 and prob.cnt_check == n_dense
    J = terms.jac_collision(x)
    assert relerr(J, d["jac0_f64"]) < 1e-5 and relerr(J, d["jac0_ref"]) < 2e-5 and prob.cnt_check == 2 * n_dense
    H = terms.hess_collision(x, d["v"])
    assert relerr(H, d["hess0_f64"]) < 5e-5 and relerr(H, d["hess0_ref"]) < 5e-5 and prob.cnt_check == 3 * n_dense
    rec = optim.givengrad_traj_optimize(rob, dc.poly_score, start, target, dict(opts))
    assert rec["success"] == bool(d["slsqp_success"]) and rec["cnt_check"] == int(d["slsqp_cnt_check"])
    assert abs(rec["cost"] - float(d["slsqp_cost"])) < 1e-3 * float(d["slsqp_cost"])
    assert relerr(np.array(rec["solution"]), d["slsqp_solution"]) < 1e-3
    rec = optim.trustconstr_traj_optimize(rob, dc.poly_score, start, target, dict(opts, MAXITER=int(d["tc_maxiter"])))
    assert rec["success"] == bool(d["tc_success"]) and rec["cnt_check"] == int(d["tc_cnt_check"])
    assert abs(rec["cost"] - float(d["tc_cost"])) < 1e-3 * float(d["tc_cost"])
    assert relerr(np.array(rec["solution"]), d["tc_solution"]) < 1e-3


@pytest.mark.parametrize("rob_name,kspec,C,zero_frac", [("baxter_left", (1, 1.0, 1.0), 1, 0.0), ("baxter_left", (0, 10.0, 2.0), 5, 0.4),
                                                         ("panda", (0, 5.0, 2.0), 1, 0.3), (None, (0, 10.0, 2.0), 1, 0.0),
                                                         ("se3", (1, 3.0, 2.0), 2, 0.5), ("planar3", (2, 0.7, 0.0), 1, 0.2)])
def test_rows_packed_on_the_device_equal_the_host_packing(rob_name, kspec, C, zero_frac, knob):
    """dcx_model_create_ex from device tensors (one packing kernel: zero rows dropped in order, constants folded, centred
    copy, 16 bytes read back) against the host packing of dcx_model_create (CPU tensors in): every output bit-identical in
    both sweep forms; then dcx_model_update into the same storage - fewer rows, more rows than the capacity, all weights
    zero - against fresh models"""
    from diffco_amd import _fkdesc, _ops
    g = torch.Generator().manual_seed(17)
    S, B = 700, 300
    if rob_name is None:
        desc, dof = _fkdesc.none_desc(6), 6
        sup = torch.rand((S, 6), generator=g) * 4 - 2
        q = (torch.rand((B, 6), generator=g) * 4 - 2).cuda()
    else:
        rob = make_robot(rob_name)
        desc, lim = rob.fk_desc(), rob.limits
        sq = torch.rand((S, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
        sup = _ops.fkine(desc, sq.cuda()).reshape(S, -1).cpu()
        q = (torch.rand((B, rob.dof), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    W = torch.randn((S, C), generator=g)
    W[torch.rand(S, generator=g) < zero_frac] = 0.0           # rows that are dropped (max_num_supports padding)
    knob("nw", 4)
    up = torch.randn((B, C), generator=g).cuda() if C > 1 else None

    def outputs(m):
        res = []
        for form in (1, 0):
            knob("xf", form)
            s, gr = m.score_grad_raw(q, up)
            res += [s, gr, m.score_jac_raw(q)[1]]
        knob("xf", -1)
        return res

    host = _ops.ScoreModel(desc, *kspec, sup, W)                 # CPU tensors: staged and packed on the host
    devm = _ops.ScoreModel(desc, *kspec, sup.cuda(), W.cuda())   # device tensors: packed by pack_rows_kernel
    for a, b in zip(outputs(host), outputs(devm)):
        assert torch.equal(a, b)
    # refill in place: a smaller support set, then one beyond the capacity, then nothing active
    for n, scale in ((250, 1.0), (S, -0.5), (40, 0.0)):
        sup2, W2 = sup[:n].cuda().contiguous(), (scale * W[:n]).cuda().contiguous()
        devm.update(sup2, W2)
        fresh = _ops.ScoreModel(desc, *kspec, sup2.cpu(), W2.cpu())
        for a, b in zip(outputs(fresh), outputs(devm)):
            assert torch.equal(a, b), (n, scale)
    assert devm.capacity >= S and float(devm.score_raw(q).abs().max()) == 0.0


@pytest.mark.parametrize("S,C,pattern", [(2500, 1, "random"), (3100, 5, "blocks"), (1024, 2, "last"), (1025, 8, "first"), (4097, 1, "none")])
def test_device_packing_across_scan_chunks(S, C, pattern):
    """the packing kernel compacts 1024 rows per pass: support sets that span several passes, with the dropped rows at
    random, in whole blocks, only the last / first row kept, and nothing dropped - bit-identical to the host packing"""
    from diffco_amd import _ops
    rob = make_robot("baxter_left")
    desc, lim = rob.fk_desc(), rob.limits
    g = torch.Generator().manual_seed(S + C)
    sq = torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    sup = _ops.fkine(desc, sq.cuda()).reshape(S, -1).cpu()
    W = torch.randn((S, C), generator=g)
    keep = torch.ones(S, dtype=torch.bool)
    if pattern == "random":
        keep = torch.rand(S, generator=g) > 0.37
    elif pattern == "blocks":
        keep[500:1600] = False
        keep[2049:2050] = False
    elif pattern == "last":
        keep[:-1] = False
    elif pattern == "first":
        keep[1:] = False
    W[~keep] = 0.0
    q = (torch.rand((257, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    up = torch.randn((257, C), generator=g).cuda() if C > 1 else None
    host = _ops.ScoreModel(desc, 0, 10.0, 2.0, sup, W)
    devm = _ops.ScoreModel(desc, 0, 10.0, 2.0, sup.cuda(), W.cuda())
    import ctypes as Ct
    n_host, n_dev = Ct.c_int64(), Ct.c_int64()
    host._lib.dcx_model_info(host._h, Ct.byref(n_host), None, None, None, None)
    devm._lib.dcx_model_info(devm._h, Ct.byref(n_dev), None, None, None, None)
    assert n_host.value == n_dev.value == int(keep.sum())
    for a, b in zip(host.score_grad_raw(q, up), devm.score_grad_raw(q, up)):
        assert torch.equal(a, b)
    assert torch.equal(host.score_jac_raw(q)[1], devm.score_jac_raw(q)[1])


def test_rq_takes_the_expanded_form_only_inside_its_rule(knob):
    """RQKernel(2) behind an FK transform: gamma * max |s - c|^2 <= 32 -> the expanded sweep (2e-6 from float64), beyond it
    the direct one (bit-identical to knob xf = 0); raw-input models never take it; both stay inside 1e-5"""
    from diffco_amd import _fkdesc, _ops
    from oracle import oracle
    rob = make_robot("baxter_left")
    desc, lim = rob.fk_desc(), rob.limits
    g = torch.Generator().manual_seed(9)
    S, B = 1500, 1024
    sq = torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    sup = _ops.fkine(desc, sq.cuda()).reshape(S, -1)
    W = torch.randn((S, 1), generator=g).cuda()
    q = (torch.rand((B, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    ss_max = float(((sup - sup.mean(0)) ** 2).sum(1).max())
    for gamma, expanded in ((0.9 * 32.0 / ss_max, True), (1.15 * 32.0 / ss_max, False)):
        m = _ops.ScoreModel(desc, 0, gamma, 2.0, sup, W)
        s, gr = m.score_grad_raw(q)
        knob("xf", 0)
        s0, g0 = m.score_grad_raw(q)
        knob("xf", -1)
        assert (not torch.equal(s, s0)) == expanded, gamma           # the rule's side of the line
        so, go, _ = oracle.score_grad(desc, 0, gamma, 2.0, sup.cpu().numpy().astype(np.float64), W.cpu().numpy().astype(np.float64),
                                      q.cpu().numpy().astype(np.float64), dtype=np.float64)
        assert relerr(s.cpu().numpy(), so) < 1e-5 and relerr(gr.cpu().numpy(), go) < 1e-5
        assert relerr(s0.cpu().numpy(), so) < 3e-6 and relerr(g0.cpu().numpy(), go) < 3e-6
    raw = _fkdesc.none_desc(12)
    m = _ops.ScoreModel(raw, 0, 1.0, 2.0, sup, W)                        # the same numbers as raw inputs: always direct
    x = sup[:B].contiguous() + 0.01
    knob("qt", 0)   # (this batch would otherwise run as 16-configuration tiles - direct too, in another summation order)
    s, gr = m.score_grad_raw(x)
    knob("xf", 0)
    s0, g0 = m.score_grad_raw(x)
    knob("xf", -1)
    knob("qt", -1)
    assert torch.equal(s, s0) and torch.equal(gr, g0)


def test_model_build_is_refused_while_the_stream_is_captured():
    """dcx_model_create_ex / dcx_model_update allocate and read 16 bytes back: on a capturing stream they return
    DCX_ERR_UNSUPPORTED before touching anything (the capture stays valid), instead of invalidating the graph"""
    from diffco_amd import _lib, _ops
    rob = make_robot("baxter_left")
    desc = rob.fk_desc()
    g = torch.Generator().manual_seed(5)
    sup = torch.randn((50, 12), generator=g).cuda()
    w = torch.randn((50, 1), generator=g).cuda()
    q = torch.zeros((8, 7), device="cuda")
    m = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, w)
    s0 = m.score_raw(q)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        m.score_raw(q)                       # (this stream's scratch exists before the capture)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
        with pytest.raises(_lib.DcxUnsupported):
            m.update(sup, 2 * w)
        with pytest.raises(_lib.DcxUnsupported):
            _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, w)
        s1 = m.score_raw(q)                  # launches themselves can be captured
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(s1, s0)


def test_checker_refills_its_model_in_place():
    """the FusedScorer behind a checker keeps ONE dcx_model across train / fit_poly style state changes (same transform,
    kernel, class count): new weights or supports are packed into it (dcx_model_update); a model someone else still
    holds is left alone"""
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    rob = make_robot("baxter_left")
    lim = rob.limits
    g = torch.Generator().manual_seed(3)
    S = 400
    sq = (torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    q = (torch.rand((64, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    dc = DiffCo(transform=rob.fkine)
    dc.support_points, dc.support_transformed = sq, rob.fkine(sq)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.randn(S, generator=g).cuda()
    s1 = dc.poly_score(q)
    h1 = dc._poly_fused._model._h.value
    dc.rbf_nodes = torch.randn(S, generator=g).cuda()                       # fit_poly's output of the next round
    s2 = dc.poly_score(q)
    assert dc._poly_fused._model._h.value == h1 and not torch.equal(s1, s2)    # the same dcx_model, refilled
    ref = DiffCo(transform=rob.fkine)
    ref.support_points, ref.support_transformed = sq, rob.fkine(sq)
    ref.rbf_kernel, ref.rbf_nodes = kernel.Polyharmonic(1, 1.0), dc.rbf_nodes.clone()
    assert torch.equal(ref.poly_score(q), s2)
    dc.support_points, dc.support_transformed = sq[:300], rob.fkine(sq[:300])  # supports retired
    dc.rbf_nodes = dc.rbf_nodes[:300].clone()
    dc._poly_fused.invalidate()
    s3 = dc.poly_score(q)
    assert dc._poly_fused._model._h.value == h1 and s3.shape == s2.shape
    held = dc._poly_fused._model.acquire()                                  # a lease: e.g. an optimiser's terms object
    dc.rbf_nodes = torch.randn(300, generator=g).cuda()
    s4 = dc.poly_score(q)
    assert dc._poly_fused._model is not held and torch.equal(held.score(q), s3) and not torch.equal(s4, s3)
    held.release()
    # the holders inside the package take that lease themselves: a sharded Adam run, the scipy drivers' constraint terms
    from diffco_amd.traj import ShardedAdamRun
    m5 = dc._poly_fused._model
    run = ShardedAdamRun(m5, rob.limits, q[:40].reshape(2, 20, 7).clone(), 0.05, 0.0, 0.3)
    assert m5.leases == 1
    dc.rbf_nodes = torch.randn(300, generator=g).cuda()
    dc.poly_score(q)
    assert dc._poly_fused._model is not m5                                  # not refilled under the run
    run.close()
    run.close()
    assert m5.leases == 0
    # a refit on features of ANOTHER width is a rebuild, not an error: transform=None, raw 7-wide rows
    raw = DiffCo(transform=None)
    raw.support_points = raw.support_transformed = sq[:100].clone()
    raw.rbf_kernel, raw.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.randn(100, generator=g).cuda()
    a = raw.poly_score(q)
    wide = torch.cat([sq[:100], sq[100:200]], dim=1)                        # 14-wide rows, same count
    raw.support_points = raw.support_transformed = wide
    raw._poly_fused.invalidate()
    b = raw.poly_score(torch.cat([q, q], dim=1))
    assert a.shape == b.shape == (64, 1) and raw._poly_fused._model.D == 14


def test_poly_score_and_grad_equals_the_autograd_route():
    """the one-launch (score, gradient) entry of the new-API checker = poly_score + torch.autograd.grad, bit for bit (C = 1: the
    backward of the autograd route multiplies the same Jacobian row by an upstream of ones), on GPU and on CPU inputs"""
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    rob = make_robot("baxter_left")
    lim = rob.limits
    g = torch.Generator().manual_seed(21)
    S = 300
    sq = (torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    dc = DiffCo(transform=rob.fkine)
    dc.support_points, dc.support_transformed = sq, rob.fkine(sq)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.randn(S, generator=g).cuda()
    for dev in ("cuda", "cpu"):
        q = (torch.rand((77, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev)
        s, gr = dc.poly_score_and_grad(q)
        qg = q.clone().requires_grad_(True)
        s2 = dc.poly_score(qg)
        (g2,) = torch.autograd.grad(s2.sum(), qg)
        # (like poly_score, the point goes where the nodes live - kernel_perceptrons.py:313 - and the results come back there)
        assert s.shape == (77, 1) and gr.shape == (77, 7) and s.device == s2.device and gr.dtype == q.dtype
        assert torch.equal(s, s2.detach()) and torch.equal(gr, g2.to(gr.device))
    one = dc.poly_score_and_grad(q[0])
    assert one[0].shape == (1, 1) and one[1].shape == (1, 7)
    foreign = DiffCo(transform=lambda x: rob.fkine(x))
    foreign.support_points, foreign.support_transformed = sq, rob.fkine(sq)
    foreign.rbf_kernel, foreign.rbf_nodes = dc.rbf_kernel, dc.rbf_nodes
    with pytest.raises(TypeError, match="fusable transform"):
        foreign.poly_score_and_grad(q)


def test_refill_waits_for_launches_on_other_streams():
    """dcx_model_update runs on the current stream; a sweep of the same model still running on ANOTHER torch stream
    must have read its rows before they are repacked.  Many long sweeps on a side stream, then the refill on the default
    stream: every side-stream result equals the first (the old rows), the next default-stream result is the new model's."""
    from diffco_amd import _ops
    rob = make_robot("baxter_left")
    lim = rob.limits
    g = torch.Generator().manual_seed(8)
    S, B = 2000, 65536
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, (torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()).reshape(S, -1)
    w0, w1 = torch.randn(S, generator=g).cuda(), torch.randn(S, generator=g).cuda()
    q = (torch.rand((B, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    m = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, w0)
    want0 = m.score_grad_raw(q)[0].clone()
    want1 = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, w1).score_raw(q).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        outs = [m.score_grad_raw(q)[0] for _ in range(40)]   # ~3.5 ms of sweeps queued on the side stream
    m.update(sup, w1)                                         # default stream: must wait for them
    got1 = m.score_raw(q)
    torch.cuda.synchronize()
    assert all(torch.equal(o, want0) for o in outs)
    assert torch.equal(got1, want1)


def test_device_trainer_equals_host_trainer_and_reference(monkeypatch):
    """f1: the persistent-workgroup trainer (dcx_train_perceptron) walks the same sequence as the host loop — same
    supports in the same order as the reference's model — for the single- and the multi-class perceptron."""
    import time
    from diffco_amd import MultiDiffCo, kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    d = load("trained_baxter")
    rob = make_robot("baxter_left")
    X, y, dist = (torch.from_numpy(d[k]) for k in ("X", "y", "dist"))
    runs = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("DCX_HOST_TRAINER", "1")
        dc = DiffCo(kernel_func=kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine)
        t0 = time.perf_counter()
        dc.train(X, y, max_iteration=3000, distance=dist)
        runs[mode] = (dc, time.perf_counter() - t0)
    monkeypatch.delenv("DCX_HOST_TRAINER")
    dev, host = runs["device"][0], runs["host"][0]
    np.testing.assert_array_equal(_np(dev.support_points), d["support_points"])
    np.testing.assert_array_equal(_np(dev.support_points), _np(host.support_points))
    assert relerr(_np(dev.gains), _np(host.gains)) < 1e-4 and relerr(_np(dev.gains), d["gains"]) < 1e-3
    assert relerr(_np(dev.kernel_matrix), _np(host.kernel_matrix)) < 1e-6
    assert torch.allclose(dev.kernel_matrix @ dev.gains, dev.hypothesis, atol=1e-4)
    print(f"trainer: device {runs['device'][1] * 1e3:.1f} ms, host loop {runs['host'][1] * 1e3:.1f} ms")
    # multi-class, old API
    m = load("trained_multi_planar2")
    r2 = make_robot("planar2")
    md = MultiDiffCo(None, kernel_func=kernel.FKKernel(r2.fkine, kernel.RQKernel(10.0)), beta=1.0)
    md.train(torch.from_numpy(m["X"]), torch.from_numpy(m["y"]), max_iteration=1500, distance=torch.from_numpy(m["dist"]))
    np.testing.assert_array_equal(_np(md.support_points), m["support_points"])
    assert relerr(_np(md.gains), m["gains"]) < 1e-3 and relerr(_np(md.hypothesis), m["hypothesis"]) < 1e-3
    # max_iteration is honoured and a 20k-sample problem (beyond the reference's CPU threshold) trains on the device
    g = torch.Generator().manual_seed(1)
    lim = rob.limits
    Xb = torch.rand((20000, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    P = rob.fkine(Xb)
    yb = torch.where(((P - torch.tensor([0.7, 0.3, 0.3])).norm(dim=-1) < 0.3).any(dim=1), 1.0, -1.0)
    big = DiffCo(kernel_func=kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine)
    t0 = time.perf_counter()
    big.train(Xb, yb, max_iteration=20000)
    print(f"trainer: 20000 samples -> {big.valid_supports} supports in {(time.perf_counter() - t0) * 1e3:.0f} ms")
    assert torch.all((big.hypothesis > 0) == (big.y > 0)) and 10 < big.valid_supports < 20000
    s = big.score(Xb[:4096])
    assert float(((s > 0) == (yb[:4096] > 0)).float().mean()) > 0.99


@pytest.mark.parametrize("n", [900, 6000])
def test_device_trainer_with_labels_other_than_plus_minus_one(monkeypatch, n):
    """the register-resident trainer keeps a label as its sign, which is only the reference's arithmetic for
    labels in {-1, +1}.  Labels 0 / 0.5 / 2 (margin y*h, target beta^((1+y)/2)*y evaluated on y itself,
    kernel_perceptrons.py:115-124) must give what the host loop gives, whatever N is (both register-resident sizes)."""
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    d = load("trained_baxter")
    rob = make_robot("baxter_left")
    X = torch.from_numpy(d["X"])
    X = X.repeat(-(-n // len(X)), 1)[:n] + 0.001 * torch.arange(n)[:, None] / n
    g = torch.Generator().manual_seed(7)
    y = torch.tensor([-1.0, 0.0, 0.5, 1.0, 2.0])[torch.randint(0, 5, (n,), generator=g)]
    runs = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("DCX_HOST_TRAINER", "1")
        dc = DiffCo(kernel_func=kernel.RQKernel(10.0), beta=0.7, transform=rob.fkine)
        dc.train(X, y, max_iteration=400)
        runs[mode] = dc
    monkeypatch.delenv("DCX_HOST_TRAINER")
    dev, host = runs["device"], runs["host"]
    np.testing.assert_array_equal(_np(dev.support_points), _np(host.support_points))
    assert relerr(_np(dev.gains), _np(host.gains)) < 1e-4 and relerr(_np(dev.hypothesis), _np(host.hypothesis)) < 1e-4


@pytest.mark.parametrize("mns", [None, 16])
@pytest.mark.parametrize("n", [60, 5000])
def test_device_trainer_all_equal_labels_then_update(mns, n):
    """an all-free sample set leaves ONE support; the force-kept second one was never selected, so its
    kernel-matrix entries exist only because the trainer mirrors row i into column i like the reference
    (kernel_perceptrons.py:117-119).  Both asserts the reference passes must pass here: `hypothesis == K @ gains`
    inside train() with max_num_supports, and inside jump_start_initialize on the next train(update=True).
    n = 60 runs the register-resident kernel, n = 5000 its 20-per-thread form."""
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    rob = make_robot("baxter_left")
    g = torch.Generator().manual_seed(5)
    lim = rob.limits
    X = torch.rand((n, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    dc = DiffCo(kernel_func=kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine, max_num_supports=mns)
    dc.train(X, -torch.ones(n), max_iteration=500)
    v = dc.valid_supports
    assert v == 2 and int((dc.gains != 0).sum()) == 1
    K = dc.kernel_matrix[:v, :v]
    assert float(K[0, 1]) == float(K[1, 0]) != 0.0
    assert torch.allclose(dc.kernel_matrix @ dc.gains, dc.hypothesis, atol=1e-5)
    # active-learning round: new samples with both labels, the two supports as the warm start
    Xn = torch.rand((200, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    yn = torch.where(rob.fkine(Xn)[:, -1, 2] > 0.3, 1.0, -1.0)
    Xa = torch.cat([Xn, dc.support_points[:v]])
    ya = torch.cat([yn, dc.y[:v]])
    mask = torch.cat([torch.zeros(200, dtype=torch.bool), torch.ones(v, dtype=torch.bool)])
    if mns is not None:
        dc.max_num_supports = 202
    dc.train(Xa, ya, update=True, exist_mask=mask, max_iteration=2000)
    v = dc.valid_supports
    assert torch.all((dc.hypothesis[:v] > 0) == (dc.y[:v] > 0))
    assert torch.allclose(dc.kernel_matrix @ dc.gains, dc.hypothesis, atol=1e-4)


@pytest.mark.parametrize("where", ["cpu", "cuda"])
def test_update_round_through_the_host_loop(where, monkeypatch):
    """jump_start_initialize assembled the n x n matrix on the GPU whenever the kernel was ours or X was a CUDA
    tensor, while gains / hypothesis went `home` (the CPU for n <= 10000) - fine for the device trainer, a device mismatch in
    the host loop (DCX_HOST_TRAINER=1; a foreign kernel callable never gets this far: the score path that seeds the new
    samples' hypothesis is HIP-only and rejects it).  The host loop must train the update round, from CPU and from CUDA
    inputs, and agree with the device trainer on the same data."""
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    rob = make_robot("baxter_left")
    g = torch.Generator().manual_seed(11)
    lim = rob.limits
    label = lambda q: torch.where(rob.fkine(q)[:, -1, 2] > 0.3, 1.0, -1.0)
    X = torch.rand((400, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    Xn = torch.rand((150, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    yX, yXn = label(X).cpu(), label(Xn).cpu()

    def two_rounds(dev):
        dc = DiffCo(kernel_func=kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine)
        dc.train(X.to(dev), yX.to(dev), max_iteration=1500)
        v = dc.valid_supports
        Xa = torch.cat([Xn.to(dev), dc.support_points[:v].to(dev)])
        ya = torch.cat([yXn.to(dev), dc.y[:v].to(dev)])
        mask = torch.cat([torch.zeros(len(Xn), dtype=torch.bool), torch.ones(v, dtype=torch.bool)]).to(dev)
        dc.train(Xa, ya, update=True, exist_mask=mask, max_iteration=1500)
        assert torch.allclose(dc.kernel_matrix @ dc.gains, dc.hypothesis, atol=1e-4)
        return dc

    ref = two_rounds("cpu")   # the device trainer
    monkeypatch.setenv("DCX_HOST_TRAINER", "1")
    got = two_rounds(where)
    monkeypatch.delenv("DCX_HOST_TRAINER")
    np.testing.assert_array_equal(_np(got.support_points), _np(ref.support_points))
    assert relerr(_np(got.gains), _np(ref.gains)) < 1e-3 and relerr(_np(got.hypothesis), _np(ref.hypothesis)) < 1e-3


def test_device_trainer_labels_outside_plus_minus_one():
    """0/1 labels (and y = 0, which the reference pulls towards 0): the register-resident kernel's margin form needs
    +-1 labels, so such a label set takes the generic loop — same result as the host loop with the actual y"""
    from diffco_amd import _ops
    from diffco_amd import _perceptron as P
    from diffco_amd import kernel
    g = torch.Generator().manual_seed(9)
    feats = torch.rand((300, 6), generator=g)
    y = (feats[:, 0] > 0.5).float()                    # labels in {0, 1}
    y[:7] = 0.5
    kf = kernel.RQKernel(10.0)
    z = torch.zeros(300)
    gd, hd, Kd, it_d, _ = _ops.train_perceptron_device(0, 10.0, 2.0, 2.0, feats, y, z, z, None, 400)
    gh, hh, Kh = z.clone(), z.clone(), torch.zeros(300, 300)
    it_h = P.train_perceptron(y.clone(), hh, gh, Kh, P.RowFiller(kf, feats, torch.device("cpu")), 2.0, 400)
    assert it_d in (it_h, it_h + 1)
    np.testing.assert_array_equal(_np(gd) != 0, gh.numpy() != 0)
    assert relerr(_np(gd), gh.numpy()) < 1e-4 and relerr(_np(hd), hh.numpy()) < 1e-4


@pytest.mark.parametrize("n,kind", [(3000, 0), (7000, 0), (20000, 0), (5000, 3)])
def test_trainer_on_several_workgroups_equals_one_workgroup_bitwise(n, kind, knob):
    """the register-resident perceptron loop on G workgroups with a grid-wide barrier per
    iteration (train_kernels.hip perceptron_grid_kernel) takes the same argmin at every iteration as one workgroup:
    identical gains, hypothesis, kernel matrix and iteration count — cold start, then a jump start from that state
    with flipped labels.  n = 20000 has no register-resident one-workgroup form: the generic kernel is the referee."""
    from diffco_amd import _ops
    rob = make_robot("baxter_left")
    g = torch.Generator().manual_seed(n)
    lim = rob.limits
    q = torch.rand((n, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    feats = rob.fkine(q.cuda()).reshape(n, -1)
    y = torch.where(feats[:, -1] + 0.3 * feats[:, -2] > 0.2, 1.0, -1.0)
    p0, p1 = (10.0, 2.0) if kind == 0 else (3.0, 3.0)  # kind 3 here: RQ with p = 3, the generic kernel function
    kind = 0
    zeros = torch.zeros(n, device="cuda")
    runs = {}
    for mode in (0, 1, 2) if n <= 7000 else (0, 1):  # 2 = the generic one-workgroup kernel
        knob("train_grid", mode)
        g1, h1, K1, it1, c1 = _ops.train_perceptron_device(kind, p0, p1, 0.8, feats, y, zeros, zeros, None, 300)
        y2 = y.clone()
        y2[::97] *= -1
        g2, h2, K2, it2, c2 = _ops.train_perceptron_device(kind, p0, p1, 0.8, feats, y2, g1, h1, K1, 200)
        runs[mode] = (g1, h1, K1, it1, c1, g2, h2, K2, it2, c2)
    a = runs[0]
    assert a[3] > 50 and int((a[0] != 0).sum()) > 5  # the loop did run
    for b in list(runs.values())[1:]:
        assert a[3] == b[3] and a[4] == b[4] and a[8] == b[8] and a[9] == b[9], (a[3], b[3], a[8], b[8])
        for k in (0, 1, 2, 5, 6, 7):
            assert torch.equal(a[k], b[k]), k


def test_trainer_decides_about_the_labels_on_the_device_and_never_waits(knob):
    """`dcx_train_perceptron` used to read the labels back (a stream synchronisation) to decide whether
    the sign-bit kernels apply.  The decision is the kernels' own now: 0 / 1 labels on the several-workgroup path (one
    grid-wide agreement, then workgroup 0 runs the generic loop) and on the one-workgroup register kernels give exactly
    what the generic kernel gives, and the call can be captured in a HIP graph and replayed."""
    from diffco_amd import _ops
    rob = make_robot("baxter_left")
    g = torch.Generator().manual_seed(5)
    lim = rob.limits
    for n in (6000, 3000):
        q = torch.rand((n, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
        feats = rob.fkine(q.cuda()).reshape(n, -1)
        y01 = torch.where(feats[:, -1] + 0.3 * feats[:, -2] > 0.2, 1.0, 0.0)   # not sign labels
        zeros = torch.zeros(n, device="cuda")
        runs = {}
        for mode in (2, 1, 0):   # the generic kernel (referee), several workgroups, one workgroup
            knob("train_grid", mode)
            runs[mode] = _ops.train_perceptron_device(0, 10.0, 2.0, 0.8, feats, y01, zeros, zeros, None, 120)
        for mode in (1, 0):
            assert runs[mode][3] == runs[2][3] and runs[mode][4] == runs[2][4]
            for k in (0, 1, 2):
                assert torch.equal(runs[mode][k], runs[2][k]), (n, mode, k)
    # capture + replay of the C ABI call (one-workgroup path: no allocation inside the call)
    knob("train_grid", 0)
    import ctypes as C
    from diffco_amd import _lib
    lib = _lib.load()
    n = 2000
    q = torch.rand((n, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    feats = rob.fkine(q.cuda()).reshape(n, -1).contiguous()
    y = torch.where(feats[:, -1] > 0.2, 1.0, -1.0).contiguous()
    want = _ops.train_perceptron_device(0, 10.0, 2.0, 0.8, feats, y, torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), None, 80)
    gains, hypo = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    K = torch.zeros((n, n), device="cuda")
    info = torch.zeros(2, device="cuda", dtype=torch.int32)
    kp = (C.c_float * 2)(10.0, 2.0)
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=s):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.dcx_train_perceptron(0, 0, kp, C.c_float(0.8), C.c_void_p(feats.data_ptr()), n, feats.shape[1],
                                            C.c_void_p(y.data_ptr()), 1, C.c_void_p(gains.data_ptr()), C.c_void_p(hypo.data_ptr()),
                                            C.c_void_p(K.data_ptr()), 80, C.c_void_p(info.data_ptr()), st))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(gains, want[0]) and torch.equal(hypo, want[1]) and int(info[0]) == want[3]


def test_trainer_survives_a_diverging_run(knob):
    """MultiQuadratic is not positive definite: the perceptron's margins overflow to inf / NaN.  The reference keeps
    walking (torch.min returns a NaN's index); so must every device kernel — a NaN wins the argmin instead of leaving
    it without a valid index (which used to send the row pointer out of bounds)."""
    from diffco_amd import _ops
    rob = make_robot("baxter_left")
    n = 5000
    g = torch.Generator().manual_seed(2)
    lim = rob.limits
    q = torch.rand((n, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    feats = rob.fkine(q.cuda()).reshape(n, -1)
    y = torch.where(feats[:, -1] + 0.3 * feats[:, -2] > 0.2, 1.0, -1.0)
    zeros = torch.zeros(n, device="cuda")
    its = []
    for mode in (0, 1, 2):
        knob("train_grid", mode)
        out = _ops.train_perceptron_device(2, 0.7, 0.0, 0.8, feats, y, zeros, zeros, None, 300)
        torch.cuda.synchronize()
        its.append(out[3])
    assert its[0] == its[1] == its[2]


def test_multiclass_backward_refuses_rows_refilled_after_its_forward():
    """a multi-class score's backward pass sweeps the model again; if train / fit_poly refilled the model's rows in
    place between forward and backward the gradient would be the new model's - it raises instead (ScoreModel.revision)"""
    from diffco_amd import _ops
    rob = make_robot("baxter_left")
    g = torch.Generator().manual_seed(4)
    lim = rob.limits
    S = 120
    sq = torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    sup = _ops.fkine(rob.fk_desc(), sq.cuda()).reshape(S, -1)
    w = torch.randn((S, 3), generator=g).cuda()
    m = _ops.ScoreModel(rob.fk_desc(), 1, 1.0, 1.0, sup, w, capacity=2 * S)
    q = (torch.rand((40, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda().requires_grad_(True)
    s = m.score(q)
    (g1,) = torch.autograd.grad(s.sum(), q, retain_graph=True)       # fine: same rows
    m.update(sup, 2 * w)
    with pytest.raises(RuntimeError, match="changed between the forward"):
        torch.autograd.grad(s.sum(), q)
    s2 = m.score(q)
    (g2,) = torch.autograd.grad(s2.sum(), q)
    assert relerr(g2.cpu().numpy(), 2 * g1.cpu().numpy()) < 1e-6
