"""GPU parity of the URDF kinematic-tree feed: dcx_fkine / dcx_fkine_vjp with DCX_FK_TREE and the
tree fused into the score kernel, against the reference-generated golden vectors (tools/make_golden_urdf.py) and
the CPU oracle.  Tolerances as in test_gpu_parity.py (metric max|a - ref| / max|ref|)."""
import numpy as np
import pytest
import torch

from helpers import URDF_NAMES, dual_panda_robot, load, relerr, urdf_robot

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32, device="cuda")


def _n(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from diffco_amd import _lib, _ops
    _lib.require_gpu()
    return _ops


@pytest.mark.parametrize("name", URDF_NAMES)
def test_tree_fkine_and_vjp(ops, name):
    from oracle import oracle
    d, rob = load("fk_" + name), urdf_robot(name)
    q = _t(d["q"]).requires_grad_(True)
    X = rob.fkine(q)
    assert X.shape == d["x64"].shape  # [B, 3, L]
    assert relerr(_n(X), d["x64"]) < 2e-6
    assert relerr(_n(X), d["x32"]) < 2e-6      # the reference's own fp32 output
    assert relerr(_n(X), oracle.fkine(rob.fk_desc(), d["q"])) < 1e-6
    (gq,) = torch.autograd.grad((X * _t(d["gx"])).sum(), q)
    assert relerr(_n(gq), d["gq64"]) < 5e-6
    assert relerr(_n(gq), d["gq32"]) < 5e-6
    for n in (1, 47):  # ragged tiles; a 1-D configuration comes back without the batch axis, like the reference
        assert relerr(_n(rob.fkine(_t(d["q"][:n]))), d["x64"][:n]) < 2e-6
    assert rob.fkine(_t(d["q"][3])).shape == d["x64"].shape[1:]
    lp = rob.link_positions(_t(d["q"]))
    k = len(rob.unique_position_link_names) - 1
    assert relerr(_n(lp[rob.unique_position_link_names[k]]), d["x64"][:, :, k]) < 2e-6


def test_multi_robot_fkine_and_fused_score(ops):
    from oracle import oracle
    d, rob = load("fk_urdf_dual_panda"), dual_panda_robot()
    q = _t(d["q"]).requires_grad_(True)
    X = rob.fkine(q)
    assert X.shape == d["x64"].shape == (48, 3, 18)
    assert relerr(_n(X), d["x64"]) < 2e-6 and relerr(_n(X), d["x32"]) < 2e-6
    (gq,) = torch.autograd.grad((X * _t(d["gx"])).sum(), q)
    assert relerr(_n(gq), d["gq64"]) < 5e-6 and relerr(_n(gq), d["gq32"]) < 5e-6
    # D = 54 features: the fused sweep runs on the next compiled width (64) with zero padding
    # (supports from other configurations than the queries: at a coincidence the fp64 oracle sees the fp32 rounding
    # of the support as a displacement, which is all that is left of a gradient whose other terms are tiny)
    rng = np.random.default_rng(11)
    lim = d["limits"]
    sq = (rng.random((40, rob.dof)) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).astype(np.float32)
    sup = _n(rob.fkine(_t(sq))).reshape(40, -1)
    W = rng.standard_normal((40, 1)).astype(np.float32)
    m = ops.ScoreModel(rob.fk_desc(), 0, 10.0, 2.0, _t(sup), _t(W))
    s, g = m.score_grad_raw(_t(d["q"]), None)
    rs, rg, _ = oracle.score_grad(rob.fk_desc(), 0, 10.0, 2.0, sup, W, d["q"], dtype=np.float64)
    assert relerr(_n(s), rs) < 1e-5 and relerr(_n(g), rg) < 1e-5


def test_point_major_layout(ops):
    d = load("fk_urdf_allegro")
    rob = urdf_robot("urdf_allegro", coord_major=False)
    q = _t(d["q"]).requires_grad_(True)
    X = rob.fkine(q)
    assert relerr(_n(X), d["x64"].transpose(0, 2, 1)) < 2e-6
    (gq,) = torch.autograd.grad((X * _t(d["gx"].transpose(0, 2, 1))).sum(), q)
    assert relerr(_n(gq), d["gq64"]) < 5e-6


def test_structural_zeros_survive_the_reverse_sweep(ops):
    """a joint that moves no feature (iiwa7's last joint: the flange origin lies on its axis) gets an EXACT zero
    gradient — the property Adam relies on (DH reverse sweep)"""
    d, rob = load("fk_urdf_iiwa7"), urdf_robot("urdf_iiwa7")
    q = _t(d["q"]).requires_grad_(True)
    (gq,) = torch.autograd.grad((rob.fkine(q) * _t(d["gx"])).sum(), q)
    exact_zero = np.where(np.abs(d["gq64"]).max(axis=0) == 0)[0]
    assert len(exact_zero) >= 1
    assert np.all(_n(gq)[:, exact_zero] == 0)


@pytest.mark.parametrize("name,kspec,C", [("urdf_panda", (1, 1.0, 1.0), 1), ("urdf_fetch_arm", (0, 10.0, 2.0), 1),
                                          ("urdf_allegro", (1, 1.0, 1.0), 3), ("urdf_trifinger", (0, 3.0, 3.0), 2),
                                          ("urdf_jaco", (2, 0.7, 0.0), 1), ("urdf_fetch", (1, 1.0, 1.0), 1),
                                          ("urdf_iiwa7_allegro", (1, 1.0, 1.0), 1), ("urdf_iiwa7_allegro", (0, 10.0, 2.0), 2)])
def test_fused_score_grad_on_urdf_trees(ops, name, kspec, C):
    """K(T(q), supports) @ W and its gradient with the tree fused into the sweep kernel, vs the fp64 oracle"""
    from oracle import oracle
    d, rob = load("fk_" + name), urdf_robot(name)
    desc = rob.fk_desc()
    rng = np.random.default_rng(7)
    lim = d["limits"]
    S, B = 300, 200
    sq = (rng.random((S, rob.dof)) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).astype(np.float32)
    q = (rng.random((B, rob.dof)) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).astype(np.float32)
    W = rng.standard_normal((S, C)).astype(np.float32)
    up = rng.standard_normal((B, C)).astype(np.float32)
    sup = _n(rob.fkine(_t(sq))).reshape(S, -1)           # supports through the same fp32 FK as the queries
    kind, p0, p1 = kspec
    m = ops.ScoreModel(desc, kind, p0, p1, _t(sup), _t(W))
    s, g = m.score_grad_raw(_t(q), _t(up) if C > 1 else None)
    rs, rg, rj = oracle.score_grad(desc, kind, p0, p1, sup, W, q, up if C > 1 else None, want_jac=True, dtype=np.float64)
    assert relerr(_n(s), rs) < 1e-5
    assert relerr(_n(g), rg) < 1e-5
    sj, jac = m.score_jac_raw(_t(q))
    assert relerr(_n(sj), rs) < 1e-5 and relerr(_n(jac), rj) < 1e-5
    assert relerr(_n(m.score_raw(_t(q))), rs) < 1e-5


def test_diffco_on_a_urdf_robot_end_to_end(ops):
    """the reference's flow (collision_checkers.py:163-200 fit -> poly_score) with robot.fkine as the transform:
    train on HIP kernel rows, fit the polyharmonic model, score with autograd, and match the oracle"""
    from oracle import oracle
    from diffco_amd import DiffCo, kernel
    rob = urdf_robot("urdf_panda")
    torch.manual_seed(3)
    q = rob.rand_configs(400)
    X = rob.fkine(q.cuda()).cpu()
    centre = X[:, :, -1].mean(0)
    dist = (X[:, :, -1] - centre).norm(dim=1) - 0.25      # synthetic ground truth: fingertip near a sphere
    labels = torch.where(dist < 0, 1.0, -1.0)
    dc = DiffCo(kernel_func=kernel.RQKernel(gamma=10), transform=rob.fkine)
    dc.train(q, labels, max_iteration=len(q), distance=dist)
    dc.fit_poly(kernel_func=kernel.Polyharmonic(k=1, epsilon=1), target="label")
    assert (torch.sign(dc.poly_score(q)[:, 0]) == labels).float().mean() > 0.97
    qt = rob.rand_configs(64).requires_grad_(True)
    s = dc.poly_score(qt)
    (g,) = torch.autograd.grad(s.sum(), qt)
    sup = dc.support_transformed.reshape(len(dc.support_transformed), -1).numpy()
    rs, rg, _ = oracle.score_grad(rob.fk_desc(), 1, 1.0, 1.0, sup, dc.rbf_nodes.reshape(len(sup), -1).numpy(),
                                  qt.detach().numpy(), dtype=np.float64)
    assert relerr(s.detach().numpy(), rs) < 1e-5 and relerr(g.numpy(), rg) < 1e-5
    assert dc.support_transformed.shape[1:] == (3, len(rob.unique_position_link_names))  # the reference's layout


def test_fused_adam_trajectory_on_a_urdf_robot(ops):
    """the fused Adam loop (dcx_traj_adam_run) drives a URDF tree as well as a DH arm: collision cost drops and
    the endpoints stay fixed"""
    from diffco_amd import DiffCo, fused_adam_traj_optimize, kernel
    rob = urdf_robot("urdf_iiwa7")
    torch.manual_seed(5)
    q = rob.rand_configs(300)
    X = rob.fkine(q.cuda()).cpu()
    dist = (X[:, :, -1] - torch.tensor([0.4, 0.0, 0.6])).norm(dim=1) - 0.3
    labels = torch.where(dist < 0, 1.0, -1.0)
    dc = DiffCo(kernel_func=kernel.RQKernel(gamma=10), transform=rob.fkine)
    dc.train(q, labels, max_iteration=len(q), distance=dist)
    dc.fit_poly(kernel_func=kernel.Polyharmonic(k=1, epsilon=1), target="label")
    start, target = q[labels < 0][0], q[labels < 0][1]
    opts = dict(N_WAYPOINTS=20, NUM_RE_TRIALS=3, MAXITER=60, safety_margin=-0.1, max_speed=0.3, seed=11, history=False)
    rec = fused_adam_traj_optimize(rob, dc.poly_score, start, target, opts)
    sol = torch.as_tensor(rec["solution"])
    assert sol.shape == (20, rob.dof)
    assert torch.allclose(sol[0], start, atol=1e-6) and torch.allclose(sol[-1], target, atol=1e-6)
    assert torch.isfinite(sol).all()


def test_forward_kinematics_diffco_facade(ops):
    """the reference's recommended facade (collision_checkers.py:318-509) on the HIP path: fit with a user ground
    truth, verify, biased collision_score with leading batch dims, q_link_pos bypass, active-learning update"""
    from diffco_amd.collision_checkers import ForwardKinematicsDiffCo, RBFDiffCo
    rob = urdf_robot("urdf_panda")
    k_tip = rob.unique_position_link_names.index("panda_virtual_ee_link")
    centre = torch.tensor([0.35, 0.0, 0.55])

    def ground_truth(q):  # 1 = in collision: the hand origin is inside a sphere
        return ((rob.fkine(q.cuda()).cpu()[:, :, k_tip] - centre).norm(dim=1) < 0.35).float()

    torch.manual_seed(0)
    fkdc = ForwardKinematicsDiffCo(robot=rob, gamma=10, gt_check_func=ground_truth)
    assert fkdc.unique_position_link_names == rob.unique_position_link_names
    acc, tpr, tnr = fkdc.fit(num_samples=1500, verify_ratio=0.2, fix_joints=[7], fix_joint_values=[0.04])
    assert acc > 0.85 and fkdc.perceptron_trained and float(fkdc.safety_bias) > 0
    n_sup = len(fkdc.perceptron.gains)
    assert 0 < n_sup < 1200
    # collision_score: leading dims are kept, the bias is added, autograd reaches q
    q = rob.rand_configs(12).reshape(3, 4, 8).requires_grad_(True)
    s = fkdc.collision_score(q)
    assert s.shape == (3, 4, 1)
    raw = fkdc.perceptron.poly_score(q.detach().reshape(-1, 8)).reshape(3, 4, 1)
    assert torch.allclose(s.detach(), raw + fkdc.safety_bias, atol=1e-6)
    (g,) = torch.autograd.grad(s.sum(), q)
    assert g.shape == q.shape and torch.isfinite(g).all() and g.abs().max() > 0
    # link positions instead of configurations (transformed_point route)
    X = rob.fkine(q.detach().reshape(-1, 8))
    s2 = fkdc.collision_score(q_link_pos=X.reshape(3, 4, *X.shape[1:]))
    assert torch.allclose(s2, s.detach(), atol=2e-5)
    assert fkdc.collision(q.detach()).dtype == torch.bool
    assert torch.allclose(fkdc.unnormalizer(fkdc.normalizer(q.detach())), q.detach(), atol=1e-5)
    # active learning: supports survive as a jump start, the model stays accurate
    acc2, _, _ = fkdc.update(num_samples=200, verify=0.2)
    assert acc2 > 0.8
    acc3, _, _ = fkdc.verify(num_samples=400)
    assert acc3 > 0.85
    # manifold-uniform sampling goes through the HIP vjp
    qs, _, _ = fkdc._generate_dataset(None, None, None, 64, sample_transform='fkine')
    assert qs.shape == (64, 8)
    # configuration-space variant and the loud failure without a ground truth
    rbf = RBFDiffCo(robot=rob, gamma=5, gt_check_func=ground_truth)
    assert rbf.fit(num_samples=600, verify_ratio=0.2)[0] > 0.7
    with pytest.raises(ValueError, match="no ground truth"):
        ForwardKinematicsDiffCo(robot=rob).fit(num_samples=10)
    with pytest.raises(NotImplementedError):
        ForwardKinematicsDiffCo(robot=rob, environment={"box": {}})


@pytest.mark.parametrize("seed", range(8))
def test_random_trees_hip_vs_oracle(ops, seed):
    """random kinematic trees (tests/helpers.random_urdf_model): HIP fkine / vjp / fused score+grad vs the fp64 oracle"""
    from oracle import oracle
    from helpers import random_urdf_model, urdf_xml
    from diffco_amd.urdf import URDFRobotFK
    rob = URDFRobotFK(urdf_xml(random_urdf_model(1000 + seed, n_links=5 + 2 * seed)))
    if rob.dof == 0 or not rob.unique_position_link_names:
        pytest.skip("degenerate tree")
    desc = rob.fk_desc()
    rng = np.random.default_rng(seed)
    q = rng.uniform(-1.5, 1.5, (130, rob.dof)).astype(np.float32)
    qt = _t(q).requires_grad_(True)
    X = rob.fkine(qt)
    ref = oracle.fkine(desc, q, np.float64)
    assert relerr(_n(X), ref) < 2e-6
    g = rng.standard_normal(ref.shape).astype(np.float32)
    (gq,) = torch.autograd.grad((X * _t(g)).sum(), qt)
    assert relerr(_n(gq), oracle.fkine_vjp(desc, q, g, np.float64)) < 5e-6
    S = 150
    sq = rng.uniform(-1.5, 1.5, (S, rob.dof)).astype(np.float32)
    sup = _n(rob.fkine(_t(sq))).reshape(S, -1)
    W = rng.standard_normal((S, 1)).astype(np.float32)
    m = ops.ScoreModel(desc, 1, 1.0, 1.0, _t(sup), _t(W))
    s, gr = m.score_grad_raw(_t(q), None)
    rs, rg, _ = oracle.score_grad(desc, 1, 1.0, 1.0, sup, W, q, dtype=np.float64)
    assert relerr(_n(s), rs) < 1e-5 and relerr(_n(gr), rg) < 1e-5


def test_host_optimisers_accept_a_urdf_robot(ops):
    """adam / SLSQP (givengrad) / trust-constr drive a URDF robot through the same entry points as a DH arm: the path
    terms use per-link norms of the [W, 3, L] features, the collision constraint its analytic fused Jacobian"""
    from diffco_amd import DiffCo, kernel, optim
    rob = urdf_robot("urdf_iiwa7")
    torch.manual_seed(7)
    q = rob.rand_configs(300)
    X = rob.fkine(q.cuda()).cpu()
    dist = (X[:, :, -1] - torch.tensor([0.45, 0.0, 0.55])).norm(dim=1) - 0.25
    labels = torch.where(dist < 0, 1.0, -1.0)
    dc = DiffCo(kernel_func=kernel.RQKernel(gamma=10), transform=rob.fkine)
    dc.train(q, labels, max_iteration=len(q), distance=dist)
    dc.fit_poly(kernel_func=kernel.Polyharmonic(k=1, epsilon=1), target="label")
    free = q[labels < 0]
    start, target = free[0], free[1]
    init = torch.from_numpy(np.linspace(start.numpy(), target.numpy(), 10)).double()
    base = dict(N_WAYPOINTS=10, NUM_RE_TRIALS=1, safety_margin=-0.2, max_speed=0.4, seed=3, history=False,
                init_solution=init, extra_optimizer_options={"disp": False})
    for fn, iters in ((optim.adam_traj_optimize, 25), (optim.givengrad_traj_optimize, 8), (optim.trustconstr_traj_optimize, 6)):
        rec = fn(rob, dc.poly_score, start, target, dict(base, MAXITER=iters))
        sol = torch.as_tensor(rec["solution"])
        assert sol.shape == (10, rob.dof) and torch.isfinite(sol).all()
        assert torch.allclose(sol[0].float(), start, atol=1e-5) and torch.allclose(sol[-1].float(), target, atol=1e-5)
        assert np.isfinite(rec["cost"])
