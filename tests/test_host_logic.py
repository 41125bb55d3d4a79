"""Host-side logic that needs no GPU: the sequential perceptron trainer (driven with a test-local torch
kernel), fit_poly's linear solve, state bookkeeping (max_num_supports padding, pickling, filtering),
path utilities and the optimisers' plumbing.  Golden data comes from the reference (tools/make_golden.py)."""
import pickle

import numpy as np
import pytest
import torch

from helpers import TorchDHRobot, TorchKernel, desc_for, load, make_robot, relerr
from oracle import oracle


def oracle_transform(desc):
    return lambda q: torch.from_numpy(oracle.fkine(desc, q.detach().numpy().astype(np.float32)))


def test_new_api_trainer_matches_reference_supports():
    from diffco_amd.kernel_perceptrons import DiffCo
    d = load("trained_baxter")
    desc = desc_for("baxter_left")
    dc = DiffCo(kernel_func=TorchKernel("rq", 10.0, 2.0), beta=1.0, transform=oracle_transform(desc))
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(d["y"])
    dc.train(X, y, max_iteration=3000, distance=torch.from_numpy(d["dist"]))
    assert dc.valid_supports == len(d["gains"]) == len(dc.gains)
    np.testing.assert_array_equal(dc.support_points.numpy(), d["support_points"])  # same samples, same order
    assert relerr(dc.gains.numpy(), d["gains"]) < 1e-3
    assert relerr(dc.hypothesis.numpy(), d["hypothesis"]) < 1e-3
    np.testing.assert_array_equal(dc.y.numpy(), d["sup_y"])
    np.testing.assert_allclose(dc.distance.numpy(), d["sup_dist"])
    assert relerr(dc.support_transformed.numpy(), d["support_transformed"]) < 1e-6
    # perfect separation of the training supports and K g == hypothesis
    assert torch.all((dc.hypothesis > 0) == (dc.y > 0))
    assert torch.allclose(dc.kernel_matrix @ dc.gains, dc.hypothesis, atol=1e-4)
    # fit_poly for the three targets
    # (the S x S polyharmonic system is ill-conditioned: the reference's ~1e-5 cdist error in K moves its
    #  nodes by ~3e-3, so nodes are compared loosely and the interpolation property tightly)
    pk = TorchKernel("poly1", 1, 1.0)
    for tgt, vals in (("label", dc.y), ("hypo", dc.hypothesis), ("dist", dc.distance)):
        dc.fit_poly(pk, target=tgt)
        assert relerr(dc.rbf_nodes.numpy(), d[f"rbf_nodes_{tgt}"]) < 2e-2, tgt
        fit = pk(dc.support_transformed, dc.support_transformed) @ dc.rbf_nodes
        assert float((fit - vals).abs().max()) < 2e-3 * float(vals.abs().max()), tgt
    # the state pickles without device handles and comes back usable
    dc2 = pickle.loads(pickle.dumps({k: v for k, v in dc.__getstate__().items() if k != "transform"}))
    assert torch.equal(dc2["gains"], dc.gains)


def test_oracle_solve_against_the_reference_nodes():
    """oracle.solve (LAPACK gesv, what torch.linalg.solve calls) on the reference's own supports and targets reproduces the
    nodes the reference's fit_poly stored (trained_baxter.npz), in its fp32 arithmetic and - within the conditioning of the
    polyharmonic system, see above - as the fp64 referee the HIP solve is held to (tests/test_gpu_solve.py)"""
    from helpers import KIND
    d = load("trained_baxter")
    S = d["support_transformed"].reshape(len(d["support_transformed"]), -1)
    for tgt, vals in (("label", d["sup_y"]), ("hypo", d["hypothesis"]), ("dist", d["sup_dist"])):
        K32 = oracle.kernel_matrix(KIND["poly"], 1, 1.0, S, S)
        n32 = oracle.solve(K32, vals, np.float32)
        n64 = oracle.solve(oracle.kernel_matrix(KIND["poly"], 1, 1.0, S, S, dtype=np.float64), vals)
        assert n32.shape == vals.shape
        assert relerr(n32, d[f"rbf_nodes_{tgt}"]) < 2e-2 and relerr(n64, d[f"rbf_nodes_{tgt}"]) < 2e-2, tgt
        # both interpolate the targets
        assert np.abs(K32.astype(np.float64) @ n64 - vals).max() < 2e-3 * np.abs(vals).max(), tgt
    # several right-hand sides at once, and a 1 x 1 system
    rhs = np.stack([d["sup_y"], d["hypothesis"]], axis=1)
    both = oracle.solve(K32, rhs)
    assert both.shape == rhs.shape and relerr(both[:, 0], oracle.solve(K32, d["sup_y"])) < 1e-9
    assert oracle.solve(np.array([[4.0]]), np.array([2.0]))[0] == 0.5


def test_max_num_supports_padding():
    from diffco_amd.kernel_perceptrons import DiffCo
    d = load("trained_baxter")
    desc = desc_for("baxter_left")
    dm = DiffCo(kernel_func=TorchKernel("rq", 10.0, 2.0), beta=1.0, transform=oracle_transform(desc),
                max_num_supports=300)
    dm.train(torch.from_numpy(d["X"]), torch.from_numpy(d["y"]), max_iteration=3000, distance=torch.from_numpy(d["dist"]))
    v = int(d["mns_valid"])
    assert dm.valid_supports == v and len(dm.gains) == 300
    np.testing.assert_array_equal(dm.support_points.numpy(), d["mns_support_points"])
    assert relerr(dm.gains.numpy(), d["mns_gains"]) < 1e-3
    assert float(dm.gains[v:].abs().max()) == 0.0 and float(dm.support_transformed[v:].abs().max()) == 0.0
    dm.fit_poly(TorchKernel("poly1", 1, 1.0), target="label")
    assert relerr(dm.rbf_nodes.numpy(), d["mns_rbf_nodes"]) < 2e-2
    assert float(dm.rbf_nodes[v:].abs().max()) == 0.0


@pytest.mark.parametrize("mns", [None, 8])
def test_all_equal_labels_keep_a_second_support(mns):
    """an all-free sample set trains to ONE support; the 'keep at least two' rule (kernel_perceptrons.py:139-141)
    then keeps a sample that was never selected, whose kernel-matrix entries against the real support must be there
    (the reference fills row AND column on selection) for `hypothesis == K @ gains` — asserted inside train() with
    max_num_supports and inside jump_start_initialize on the next train(update=True) (tests/test_gpu_api.py)"""
    from diffco_amd.kernel_perceptrons import DiffCo
    g = torch.Generator().manual_seed(3)
    X = torch.rand((40, 3), generator=g)
    y = -torch.ones(40)
    dc = DiffCo(kernel_func=TorchKernel("rq", 10.0, 2.0), beta=1.0, transform=None, max_num_supports=mns)
    dc.train(X, y, max_iteration=200)
    v = dc.valid_supports
    assert v == 2 and int((dc.gains != 0).sum()) == 1
    assert torch.allclose(dc.kernel_matrix @ dc.gains, dc.hypothesis, atol=1e-5)
    assert float(dc.kernel_matrix[:v, :v].abs().min()) > 0 or float(dc.kernel_matrix[1, 1]) == 0.0
    assert float(dc.kernel_matrix[0, 1]) == float(dc.kernel_matrix[1, 0]) != 0.0


def test_old_api_multiclass_trainer_matches_reference():
    from diffco_amd import deprecated, kernel
    d = load("trained_multi_planar2")
    desc = desc_for("planar2")
    fk = oracle_transform(desc)
    md = deprecated.MultiDiffCo(None, kernel_func=kernel.FKKernel(fk, TorchKernel("rq", 10.0, 2.0)), beta=1.0)
    md.train(torch.from_numpy(d["X"]), torch.from_numpy(d["y"]), max_iteration=1500, distance=torch.from_numpy(d["dist"]))
    assert md.num_class == 2
    np.testing.assert_array_equal(md.support_points.numpy(), d["support_points"])
    assert relerr(md.gains.numpy(), d["gains"]) < 1e-3
    assert relerr(md.hypothesis.numpy(), d["hypothesis"]) < 1e-3
    md.fit_poly(kernel_func=TorchKernel("poly1", 1, 1.0), target="label", fkine=fk, reg=0.0)
    assert md.rbf_nodes.shape == d["rbf_nodes"].shape
    assert relerr(md.rbf_nodes.numpy(), d["rbf_nodes"]) < 2e-2
    assert torch.all(md.rbf_nodes[md.gains == 0] == 0)  # per-class sparsity (deprecated/MultiDiffCo.py:152-153)


def test_score_path_rejects_foreign_kernels_instead_of_falling_back():
    from diffco_amd.kernel_perceptrons import DiffCo
    dc = DiffCo(kernel_func=TorchKernel("rq", 10.0, 2.0))
    dc.support_points = dc.support_transformed = torch.zeros(3, 4)
    dc.gains = torch.ones(3)
    with pytest.raises(TypeError, match="HIP-only"):
        dc.score(torch.zeros(2, 4))


def test_dense_path_matches_reference():
    from diffco_amd import utils
    d = load("dense_path")
    p = torch.from_numpy(d["path"])
    np.testing.assert_allclose(utils.dense_path(p, 0.3).numpy(), d["dense_0p3"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(utils.dense_path(p, 2.0).numpy(), d["dense_2p0"], rtol=0, atol=1e-12)
    few = utils.dense_path(p, 0.05, max_step_num=20)
    assert torch.equal(few[0], p[0]) and torch.equal(few[-1], p[-1]) and len(few) <= 20 + len(p)


def test_angle_utils():
    from diffco_amd import utils
    assert abs(float(utils.wrap2pi(torch.tensor(3 * np.pi / 2))) + np.pi / 2) < 1e-6
    a = utils.anglin([-0.8 * np.pi], [0.9 * np.pi], 3, endpoint=True)
    assert a.shape == (3, 1) and abs(float(a[1, 0]) - (-0.95 * np.pi)) < 1e-6  # goes the short way round
    r = utils.rot_2d(torch.tensor([0.3]))
    assert torch.allclose(r[0] @ r[0].T, torch.eye(2), atol=1e-6)
    assert torch.allclose(utils.rotz(torch.tensor([0.3]))[0, :2, :2], r[0])
    mc = utils.make_continue(torch.tensor([[3.0], [-3.1], [-3.0]]))
    assert float((mc[1:] - mc[:-1]).abs().max()) < 1.0


def test_shard_bounds_cover_exactly():
    from diffco_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_robot_descriptions():
    """parameters of the drop-in robot classes (the FK itself is checked against golden FK vectors)"""
    from diffco_amd import model
    b = model.BaxterLeftArmFK()
    assert b.dof == 7 and tuple(b.limits.shape) == (7, 2) and b.fk_desc().feature_dim == 12
    assert model.BaxterFK is model.BaxterLeftArmFK
    assert model.BaxterDualArmFK().fk_desc().feature_dim == 24 and model.BaxterDualArmFK().dof == 14
    assert model.PandaFK().fk_desc().feature_dim == 21 and model.PandaFK(fingers=False).fk_desc().feature_dim == 15
    dp = model.DualPandaFK()
    assert dp.dof == 14 and dp.fk_desc().feature_dim == 42 and tuple(dp.limits.shape) == (14, 2)
    assert float(dp.limits[6, 1]) == pytest.approx(-0.0698) and float(dp.limits[7, 1]) == pytest.approx(-0.0698)
    pl = model.RevolutePlanarRobot(1.0, 0.3, dof=2)
    assert pl.fk_desc().feature_dim == 4 and torch.allclose(pl.limits, torch.tensor([[-np.pi, np.pi]] * 2))
    assert torch.allclose(pl.wrap(torch.tensor([4.0])), torch.tensor([4.0 - 2 * np.pi]))
    with pytest.raises(ValueError):
        model.RigidBody()


def test_adam_optimizer_plumbing_against_reference_record():
    """adam_traj_optimize on a torch-only dist_est/robot reproduces the reference's record (same init path,
    options and support set; the reference's fp32 cdist vs direct fp64 differences explains the tolerance)"""
    from diffco_amd import optim
    d = load("optim_adam_baxter")
    import json, os
    from helpers import GOLDEN
    options = json.load(open(os.path.join(GOLDEN, "optim_adam_baxter_options.json")))
    rob = TorchDHRobot(make_robot("baxter_left"))
    sup = rob.fkine(torch.from_numpy(d["sup_q"]).double()).reshape(len(d["sup_q"]), -1)
    w = torch.from_numpy(d["weights"]).double()
    kern = TorchKernel("poly1", 1, 1.0)

    def dist_est(p):
        return kern(rob.fkine(p).reshape(len(p), -1), sup) @ w[:, None]
    # the fused loss at the initial path (what the fused optimiser will have to reproduce too)
    p = torch.from_numpy(d["init"]).clone().requires_grad_(True)
    col = torch.clamp(dist_est(p), min=0).sum()
    cp = rob.fkine(p)
    mm = torch.clamp((cp[1:] - cp[:-1]).square().sum(dim=2) - 0.3 ** 2, min=0).sum()
    lim = rob.limits.double()
    jl = (torch.clamp(lim[:, 0] - p, min=0) + torch.clamp(p - lim[:, 1], min=0)).sum()
    diff = (cp[1:] - cp[:-1]).square().sum()
    loss = diff + 10 * col + 10 * mm + 10 * jl
    assert abs(loss.item() - float(d["loss0"])) < 1e-4 * abs(float(d["loss0"]))
    (g,) = torch.autograd.grad(loss, p)
    assert relerr(g.numpy(), d["grad0"]) < 1e-4
    options["init_solution"] = torch.from_numpy(d["init"]).clone()
    rec = optim.adam_traj_optimize(rob, dist_est, torch.from_numpy(d["start"]), torch.from_numpy(d["target"]), options)
    assert rec["success"] == bool(d["success"]) and rec["cnt_check"] == int(d["cnt_check"])
    assert abs(rec["cost"] - float(d["cost"])) < 2e-3 * float(d["cost"])
    assert relerr(np.array(rec["solution"]), d["solution"]) < 2e-3
    assert set(rec) == {"start_cfg", "target_cfg", "cnt_check", "cost", "time", "success", "seed", "solution"}


def test_scipy_optimizers_run_on_a_toy_problem():
    """SLSQP / trust-constr / gradient-free drivers: a 2-DoF point 'robot' steering round a bump"""
    from diffco_amd import optim

    class Point2D:
        dof = 2
        limits = torch.tensor([[-3.0, 3.0], [-3.0, 3.0]])

        def fkine(self, q, reuse=False):
            return q.reshape(-1, 1, 2)

    def bump(q):  # positive inside a disc of radius 0.7 around the origin
        return (0.49 - (q.reshape(-1, 2) ** 2).sum(-1)).reshape(-1, 1)
    opts = {"N_WAYPOINTS": 8, "NUM_RE_TRIALS": 1, "MAXITER": 60, "safety_margin": -0.05, "max_speed": 0.4, "seed": 3,
            "history": False, "extra_optimizer_options": {}}
    start, target = torch.tensor([-1.5, 0.1]), torch.tensor([1.5, 0.1])
    for fn in (optim.givengrad_traj_optimize, optim.trustconstr_traj_optimize):
        rec = fn(Point2D(), bump, start, target, dict(opts))
        sol = torch.tensor(rec["solution"], dtype=torch.float64)
        assert sol.shape == (8, 2) and torch.allclose(sol[0], start.double()) and torch.allclose(sol[-1], target.double())
        assert float(bump(sol).max()) < 0.0 + 1e-3, fn.__name__  # the path leaves the bump
        assert rec["cnt_check"] > 0 and np.isfinite(rec["cost"])
    # trust-constr's constraint Hessian: a foreign dist_est gets a BFGS model by default ('auto'); 'autograd' is the
    # reference's double backward (optim.py:380-391); 'fused' needs a diffco_amd score
    for mode in ("bfgs", "autograd"):
        rec = optim.trustconstr_traj_optimize(Point2D(), bump, start, target, dict(opts, constraint_hessian=mode))
        sol = torch.tensor(rec["solution"], dtype=torch.float64)
        assert float(bump(sol).max()) < 1e-3, mode
    with pytest.raises(ValueError):
        optim.trustconstr_traj_optimize(Point2D(), bump, start, target, dict(opts, constraint_hessian="fused"))
    with pytest.raises(ValueError):
        optim.trustconstr_traj_optimize(Point2D(), bump, start, target, dict(opts, constraint_hessian="exact"))
    rec = optim.gradient_free_traj_optimize(Point2D(), bump, start, target, dict(opts, MAXITER=15))
    assert len(rec["solution"]) == 8 and np.isfinite(rec["cost"])
    two = dict(opts, init_solution=torch.stack([start, target]).double())
    rec = optim.adam_traj_optimize(Point2D(), bump, start, target, two)
    assert rec["success"] and rec["cnt_check"] == 0


def test_fused_optimizer_selection_policy():
    from diffco_amd.traj import select_trial
    inf = float("inf")
    steps = torch.tensor([200., 200., 57., 200.])
    # trial 2 is the first valid one: it wins even though trial 3 has a better objective; checks stop there
    t, ok, cost, cnt = select_trial(torch.tensor([inf, inf, 0.9, 0.4]), torch.tensor([5., 3., 1., .5]),
                                    torch.tensor([4., 2., .9, .4]), steps, 20)
    assert (t, ok, cost, cnt) == (2, True, pytest.approx(0.9), (200 + 200 + 57) * 20)
    # nothing valid: lowest loss over all trials, every step counted
    t, ok, cost, cnt = select_trial(torch.full((4,), inf), torch.tensor([5., 3., 7., 4.]), torch.tensor([4., 2., 6., 3.]),
                                    steps, 20)
    assert (t, ok, cost, cnt) == (1, False, pytest.approx(2.0), 657 * 20)


def test_analytic_constraint_jacobian_matches_autograd():
    """f4: the segment-collision Jacobian assembled from per-point hinge gradients and the dense-path geometry
    equals autograd's Jacobian of the same constraint (a stand-in 'model' supplies score and hinge gradient)"""
    from diffco_amd import optim, utils
    g = torch.Generator().manual_seed(0)
    p = torch.randn((6, 3), generator=g, dtype=torch.float64)
    d1 = utils.dense_path(p, 0.3)
    d2, seg, st = utils.dense_path_indexed(p, 0.3)
    assert torch.equal(d1, d2) and len(seg) == len(d1) and int(seg[-1]) == 5 and float(st[-1]) == 0.0

    class FakeModel:
        C, dev = 1, torch.device("cpu")

        def score_hinge_grad_raw(self, q, margin, weight):
            s = torch.sin(q.double()).sum(1, keepdim=True)
            return s, torch.cos(q.double()) * ((s - margin) > 0).double() * weight

    class Rob:
        dof, limits = 3, torch.tensor([[-3.0, 3.0]] * 3)
    prob = optim._PathProblem(Rob(), p[0], p[-1], {"N_WAYPOINTS": 6, "max_speed": 0.3, "safety_margin": 0.1})
    prob.init_path = p.clone()
    terms = optim._ScipyTerms(prob, lambda q: torch.sin(q).sum(1, keepdim=True))
    x = p[1:-1].reshape(-1).numpy()
    Ja = terms._jac_collision_fused(x, FakeModel())
    terms._model = None  # force the autograd route
    Jb = terms.jac_collision(x)
    assert Ja.shape == Jb.shape == (5, 12) and np.abs(Jb).max() > 1.0
    assert np.abs(Ja - Jb).max() < 1e-6 * np.abs(Jb).max()  # the fused route evaluates the points in fp32


def test_constraint_hessian_from_point_hessians_matches_double_backward():
    """trust-constr's `hess`: the (stand-in) analytic per-point Hessians at the dense points, chained through the
    dense-path geometry by autograd on the Taylor surrogate, equal the reference's double backward through dist_est
    (optim.py:380-391)"""
    from diffco_amd import optim
    g = torch.Generator().manual_seed(1)
    p = torch.randn((6, 3), generator=g, dtype=torch.float64)

    class FakeModel:
        C, dev = 1, torch.device("cpu")

        def score_grad_raw(self, q, upstream=None, want_score=True):
            return torch.sin(q.double()).sum(1, keepdim=True), torch.cos(q.double())

        def score_hess_raw(self, q, upstream=None):
            return torch.cos(q.double()), torch.diag_embed(-torch.sin(q.double()))

    class Rob:
        dof, limits = 3, torch.tensor([[-3.0, 3.0]] * 3)
    prob = optim._PathProblem(Rob(), p[0], p[-1], {"N_WAYPOINTS": 6, "max_speed": 0.3, "safety_margin": 0.1})
    prob.init_path = p.clone()
    terms = optim._ScipyTerms(prob, lambda q: torch.sin(q).sum(1, keepdim=True))
    x = p[1:-1].reshape(-1).numpy()
    v = np.random.default_rng(0).standard_normal(5)
    Ha = terms._hess_collision_fused(x, torch.from_numpy(v), FakeModel())
    terms._model = None  # the reference's route
    Hb = terms.hess_collision(x, v)
    assert Ha.shape == Hb.shape == (12, 12) and np.abs(Hb).max() > 0.5
    assert np.abs(Ha - Ha.T).max() < 1e-12
    assert np.abs(Ha - Hb).max() < 1e-6 * np.abs(Hb).max()  # the dense points go through fp32 on the fused route

    # a transform dcx_score_hess cannot hold: differences of the analytic gradient instead
    from diffco_amd import _lib

    class NoHess(FakeModel):
        def score_hess_raw(self, q, upstream=None):
            raise _lib.DcxUnsupported("frames do not fit")
    Hc = terms._hess_collision_fused(x, torch.from_numpy(v), NoHess())
    assert np.abs(Hc - Hb).max() < 1e-5 * np.abs(Hb).max()  # step^2 / 6 truncation of the central difference


def _scipy_fixture():
    """the SLSQP / trust-constr fixture (tools/make_golden.py gen_optim_scipy: the reference's drivers and their collision
    constraint, optim.py:166-516) with a float64 torch restatement of its model"""
    from diffco_amd import optim
    d = load("optim_scipy_baxter")
    rob = TorchDHRobot(make_robot("baxter_left"))
    sup = rob.fkine(torch.from_numpy(d["sup_q"]).double()).reshape(len(d["sup_q"]), -1)
    w = torch.from_numpy(d["weights"]).double()
    kern = TorchKernel("poly1", 1, 1.0)
    dist_est = lambda p: kern(rob.fkine(p).reshape(len(p), -1), sup) @ w[:, None]
    start, target, init = (torch.from_numpy(d[k]).double() for k in ("start", "target", "init"))
    opts = {"N_WAYPOINTS": len(init), "NUM_RE_TRIALS": 1, "MAXITER": int(d["slsqp_maxiter"]), "safety_margin": float(d["margin"]),
            "max_speed": float(d["max_speed"]), "seed": 4321, "history": False, "extra_optimizer_options": {"disp": False},
            "init_solution": init.clone()}
    prob = optim._PathProblem(rob, start, target, dict(opts))
    prob.make_init(0)
    return d, rob, dist_est, start, target, opts, prob


def test_scipy_constraint_terms_against_the_reference_fixture():
    """row f4: `_ScipyTerms.collision / jac_collision / hess_collision` (autograd route, float64 dist_est) against the
    reference's con_collision_free, its Jacobian and the Hessian of v . con at the initial path (optim.py:190-218,
    380-391); `cnt_check` advances by len(dense path) per evaluation as the reference's counter does (:197)"""
    from diffco_amd import optim
    d, rob, dist_est, start, target, opts, prob = _scipy_fixture()
    terms = optim._ScipyTerms(prob, dist_est)
    x = prob.init_path[1:-1].reshape(-1).numpy()
    n_dense = int(d["n_dense"])
    c = terms.collision(x)
    assert prob.cnt_check == n_dense
    assert (c < 0).sum() == (d["con0_f64"] < 0).sum() >= 3          # an active constraint
    assert relerr(c, d["con0_f64"]) < 1e-12 and relerr(c, d["con0_ref"]) < 2e-5
    J = terms.jac_collision(x)
    assert prob.cnt_check == 2 * n_dense
    assert J.shape == d["jac0_f64"].shape and relerr(J, d["jac0_f64"]) < 1e-12 and relerr(J, d["jac0_ref"]) < 2e-5
    H = terms.hess_collision(x, d["v"])
    assert prob.cnt_check == 3 * n_dense
    assert H.shape == d["hess0_f64"].shape and relerr(H, d["hess0_f64"]) < 1e-11 and relerr(H, d["hess0_ref"]) < 5e-5


def test_scipy_drivers_reproduce_the_reference_records():
    """row f4: givengrad_traj_optimize (SLSQP) and trustconstr_traj_optimize on the float64 restatement of the model walk
    the iterations the reference's drivers walked (optim.py:166-321, 324-516): same cnt_check, cost, solution"""
    from diffco_amd import optim
    d, rob, dist_est, start, target, opts, _ = _scipy_fixture()
    rec = optim.givengrad_traj_optimize(rob, dist_est, start, target, dict(opts))
    assert rec["success"] == bool(d["slsqp_success"]) and rec["cnt_check"] == int(d["slsqp_cnt_check"])
    assert abs(rec["cost"] - float(d["slsqp_cost"])) < 1e-5 * float(d["slsqp_cost"])
    assert relerr(np.array(rec["solution"]), d["slsqp_solution"]) < 1e-5
    rec = optim.trustconstr_traj_optimize(rob, dist_est, start, target,
                                          dict(opts, MAXITER=int(d["tc_maxiter"]), constraint_hessian="autograd"))
    assert rec["success"] == bool(d["tc_success"]) and rec["cnt_check"] == int(d["tc_cnt_check"])
    assert abs(rec["cost"] - float(d["tc_cost"])) < 1e-5 * float(d["tc_cost"])
    assert relerr(np.array(rec["solution"]), d["tc_solution"]) < 1e-5


def test_fused_constraint_jacobian_chain_rule_on_the_oracle():
    """row f4: `_jac_collision_fused` - the constraint Jacobian assembled from ONE hinge-gradient evaluation over the dense
    path, chained through dense_n = p_i + k * max_step * unit(p_{i+1} - p_i) - against the reference's Jacobian, with the
    C oracle standing in for the HIP launch (the GPU test runs the same comparison through libdcx)"""
    from diffco_amd import optim
    from oracle import oracle
    d, rob, dist_est, start, target, opts, prob = _scipy_fixture()
    desc = make_robot("baxter_left").fk_desc()
    sup32 = oracle.fkine(desc, d["sup_q"].astype(np.float64), dtype=np.float64).reshape(len(d["sup_q"]), -1)

    class OracleModel:
        dev, C = torch.device("cpu"), 1

        def score_hinge_grad_raw(self, q, margin, weight):
            s, g, _ = oracle.score_grad(desc, 1, 1.0, 1.0, sup32, d["weights"].astype(np.float64).reshape(-1, 1),
                                        q.double().numpy(), dtype=np.float64)
            mask = ((s[:, 0] - margin) > 0) * weight
            return torch.from_numpy(s), torch.from_numpy(g * mask[:, None])
    terms = optim._ScipyTerms(prob, dist_est)
    x = prob.init_path[1:-1].reshape(-1).numpy()
    J = terms._jac_collision_fused(x, OracleModel())
    assert relerr(J, d["jac0_f64"]) < 2e-6   # (the dense points travel as fp32 to the launch)


def test_escape_host_loop_against_the_reference_record():
    """OptimSampler's host loop (the route a foreign dist_est takes) on a torch-only spline score reproduces the records the
    REFERENCE's OptimSampler made (tests/golden/escape.npz: scripts/escape.py:19-38 run on reference checkers), evaluation
    and record counts exactly; and the rules that decide whether a loop can run as dcx_escape_adam"""
    from diffco_amd import utils
    from diffco_amd.escape import OptimSampler, resampling_escape
    d = load("escape")
    rob = TorchDHRobot(make_robot("baxter_left"))
    sup = rob.fkine(torch.from_numpy(d["bx_sup_q"]).double()).reshape(len(d["bx_sup_q"]), -1)
    w = torch.from_numpy(d["bx_w"]).double()
    kern = TorchKernel("poly1", 1, 1.0)

    def dist_est(p):
        s = kern(rob.fkine(p.double()).reshape(-1, sup.shape[1]), sup) @ w
        return s.to(p.dtype)
    starts = torch.from_numpy(d["bx_starts"])
    assert relerr(dist_est(starts).numpy(), d["bx_score0"].reshape(-1)) < 1e-5
    m1, m3 = float(d["bx_margin1"]), float(d["bx_margin3"])
    for tag, start, args in (
            ("bx_single", starts[:1], {"N_WAYPOINTS": 20, "safety_margin": m1, "lr": 5e-2, "record_freq": 1}),
            ("bx_flat", starts[0], {"N_WAYPOINTS": 20, "safety_margin": m1, "lr": 5e-2, "record_freq": 3}),
            ("bx_three", starts[:3], {"N_WAYPOINTS": 12, "safety_margin": m3, "lr": 2e-2, "record_freq": 2,
                                      "opt_args": {"lr": 2e-2, "betas": (0.8, 0.99), "eps": 1e-6}})):
        sampler = OptimSampler(rob, dist_est, args)
        hist, checks = sampler.optim_escape(start)
        assert sampler.last_route == "host" and checks == int(d[tag + "_checks"])
        assert tuple(hist.shape) == d[tag + "_hist"].shape and relerr(hist.numpy(), d[tag + "_hist"]) < 1e-4
        assert torch.equal(hist[0], start)          # the first record is the start itself (escape.py:29-30)
    # the batch entry point has no host form
    with pytest.raises(TypeError):
        OptimSampler(rob, dist_est, {}).optim_escape_batch(starts)
    # what the fused form accepts
    s = OptimSampler(rob, dist_est, {"lr": 0.2})
    assert s._adam() == (0.2, 0.9, 0.999, 1e-8) and s._wrap_mask(7) == 0
    assert OptimSampler(rob, dist_est, {"opt_args": {"lr": 0.1, "betas": (0.5, 0.9), "eps": 1e-6}})._adam() == (0.1, 0.5, 0.9, 1e-6)
    assert OptimSampler(rob, dist_est, {"opt_args": {"lr": 0.1, "weight_decay": 0.1}})._adam() is None
    assert OptimSampler(rob, dist_est, {"opt_args": {"lr": 0.1, "amsgrad": True}})._adam() is None
    assert OptimSampler(rob, dist_est, {"optimizer": torch.optim.SGD})._adam() is None
    assert OptimSampler(rob, dist_est, {"post_transform": utils.wrap2pi})._wrap_mask(7) == 127
    assert OptimSampler(rob, dist_est, {"post_transform": utils.se2_wrap2pi})._wrap_mask(3) == 4
    assert OptimSampler(rob, dist_est, {"post_transform": lambda x: x})._wrap_mask(3) is None
    cfg = resampling_escape(rob)
    assert cfg.shape == (1, 7) and bool(((cfg >= rob.limits[:, 0]) & (cfg <= rob.limits[:, 1])).all())


def test_multiclass_adam_plumbing_and_oracle_against_the_reference_record():
    """tests/golden/optim_multi_baxter.npz (the reference's adam_traj_optimize on a reference MultiDiffCo with a [C] safety margin,
    tools/make_golden.py gen_optim_multi) on the CPU: the fixture's class scores through the C oracle; the loss terms and gradient
    through a float64 torch restatement of the multi-class collision term sum_c clamp(score_c - margin_c, 0) (optim.py:88-89); the
    host optimiser with a vector margin reproducing the record; and the scipy terms' flat-reshape bookkeeping for several
    classes (optim.py:199-207)"""
    from diffco_amd import optim
    d = load("optim_multi_baxter")
    desc = desc_for("baxter_left")
    rob = TorchDHRobot(make_robot("baxter_left"))
    S, C = d["weights"].shape
    sup = rob.fkine(torch.from_numpy(d["sup_q"]).double()).reshape(S, -1)
    W = torch.from_numpy(d["weights"]).double()
    margin = torch.from_numpy(d["margin"]).double()
    kern = TorchKernel("poly1", 1, 1.0)

    def dist_est(p):
        return kern(rob.fkine(p).reshape(len(p), -1), sup) @ W
    init = torch.from_numpy(d["init"])
    # the oracle (fp64) on the same state agrees with the reference's scores at the initial path
    so, _, _ = oracle.score_grad(desc, 1, 1.0, 1.0, sup.numpy(), W.numpy(), init.numpy(), dtype=np.float64)
    assert relerr(so, d["score_init"]) < 1e-9
    # and its gradient with the hinge's upstream is the collision part of the reference's gradient: check the whole loss
    p = init.clone().requires_grad_(True)
    col = torch.clamp(dist_est(p) - margin, min=0).sum()
    cp = rob.fkine(p)
    mm = torch.clamp((cp[1:] - cp[:-1]).square().sum(dim=2) - float(d["max_speed"]) ** 2, min=0).sum()
    lim = rob.limits.double()
    jl = (torch.clamp(lim[:, 0] - p, min=0) + torch.clamp(p - lim[:, 1], min=0)).sum()
    diff = (cp[1:] - cp[:-1]).square().sum()
    loss = diff + 10 * col + 10 * mm + 10 * jl
    assert relerr(torch.stack([diff, col, mm, jl]).detach().numpy(), d["loss0_terms"]) < 1e-9
    (g,) = torch.autograd.grad(loss, p, retain_graph=True)
    assert relerr(g.numpy(), d["grad0"]) < 1e-8
    up = ((torch.from_numpy(so) - margin) > 0).double().numpy() * 10.0
    _, go, _ = oracle.score_grad(desc, 1, 1.0, 1.0, sup.numpy(), W.numpy(), init.numpy(), upstream=up, dtype=np.float64)
    (g_path,) = torch.autograd.grad(diff + 10 * mm + 10 * jl, p, allow_unused=True)
    assert relerr(go + g_path.numpy(), d["grad0"]) < 1e-7
    options = {"N_WAYPOINTS": len(init), "NUM_RE_TRIALS": 1, "MAXITER": int(d["maxiter"]), "safety_margin": margin,
               "max_speed": float(d["max_speed"]), "seed": int(d["seed"]), "history": False,
               "extra_optimizer_options": {"lr": float(d["lr"])}, "init_solution": init.clone()}
    rec = optim.adam_traj_optimize(rob, dist_est, torch.from_numpy(d["start"]), torch.from_numpy(d["target"]), options)
    assert rec["success"] == bool(d["success"]) and rec["cnt_check"] == int(d["cnt_check"])
    assert abs(rec["cost"] - float(d["cost"])) < 1e-6 * float(d["cost"])
    assert relerr(np.array(rec["solution"]), d["solution"]) < 1e-6
    # the scipy terms on the same checker (autograd route on the CPU): the reference's constraint, Jacobian and Hessian
    start, target, init2 = (torch.from_numpy(d[k]).double() for k in ("start", "target", "init2"))
    opts = {"N_WAYPOINTS": len(init2), "NUM_RE_TRIALS": 1, "MAXITER": 5, "safety_margin": torch.from_numpy(d["margin2"]).double(),
            "max_speed": float(d["max_speed2"]), "seed": 1, "history": False, "init_solution": init2.clone()}
    prob = optim._PathProblem(rob, start, target, opts)
    prob.make_init(0)
    terms = optim._ScipyTerms(prob, dist_est)
    x = prob.init_path[1:-1].reshape(-1).numpy()
    assert relerr(terms.collision(x), d["con0"]) < 1e-9 and prob.cnt_check == int(d["n_dense2"])
    assert relerr(terms.jac_collision(x), d["jac0"]) < 1e-8
    n_seg, n_pt = len(init2) - 1, int(d["n_dense2"]) - 2
    assert optim._ScipyTerms._flat_per(n_pt, n_seg, C) == n_pt * C // n_seg
    with pytest.raises(RuntimeError):
        optim._ScipyTerms._flat_per(n_pt + 1, n_seg, C)     # 23 points, 5 classes, 11 rows: the reference's reshape fails too
