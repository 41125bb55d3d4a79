"""GPU parity of the analytic second-derivative kernel (dcx_score_hess, hess_kernel.hip) — what the reference obtains
with torch.autograd.functional.hessian through dist_est for trust-constr's constraint Hessian (optim.py:380-391).

Pins: (a) tests/golden/hess_points.npz — the reference's own fp32 double backward and a float64 referee at 6
configurations of 14 score fixtures (tools/make_golden.py gen_hess); (b) central differences of the float64 CPU
oracle's analytic gradient (step 1e-5: ~1e-9 of the Hessian) at more configurations and on URDF trees.
Metric max|a - ref| / max|ref|, as everywhere.

Every case runs in both forms of the kernel (knob hess_form): 0 = one lane per (configuration, direction) sweeps the
supports; 1 = the moments form (hess_moments_kernel: one lane per configuration sweeps gX, sum c and the symmetric D x D
matrix, the direction lanes take M dx from it) wherever it is compiled - widths 2 .. 16; the rule picks it from B = 1024."""
import numpy as np
import pytest
import torch

from helpers import CASE_ROBOT, case_kernel, desc_for, load, relerr, urdf_robot

pytestmark = pytest.mark.gpu
TOL_H = 2e-5  # fp32 second derivatives (the reference's own fp32 Hessians sit 2e-7 .. 4e-5 from the float64 referee)


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32, device="cuda")


def _n(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from diffco_amd import _lib, _ops
    _lib.require_gpu()
    return _ops


@pytest.fixture(autouse=True, params=[0, 1], ids=["lanes", "moments"])
def form(request, knob):
    knob("hess_form", request.param)
    return request.param


def _model(ops, name, d):
    kind, p0, p1 = case_kernel(d)
    desc = desc_for(CASE_ROBOT[name], dof=d["q"].shape[1])
    sup = _t(d["sup_x32"])
    return ops.ScoreModel(desc, kind, p0, p1, sup.reshape(len(sup), -1), _t(d["weights"])), desc, (kind, p0, p1)


def _oracle_hess(desc, kspec, sup, W, q, up, eps=1e-5):
    """central differences of the float64 oracle's analytic gradient: H[b, i, :] = (g(q + eps e_i) - g(q - eps e_i)) / 2 eps"""
    from oracle import oracle
    q = np.asarray(q, dtype=np.float64)
    B, dof = q.shape
    H = np.empty((B, dof, dof))
    for i in range(dof):
        e = np.zeros(dof)
        e[i] = eps
        _, gp, _ = oracle.score_grad(desc, *kspec, sup, W, q + e, upstream=up, dtype=np.float64)
        _, gm, _ = oracle.score_grad(desc, *kspec, sup, W, q - e, upstream=up, dtype=np.float64)
        H[:, i, :] = (gp - gm) / (2 * eps)
    return H


GOLDEN = sorted(k.split("/")[0] for k in load("hess_points").files if k.endswith("/hess64"))


@pytest.mark.parametrize("name", GOLDEN)
def test_hessian_vs_reference_double_backward(ops, name):
    h = load("hess_points")
    d = load(name)
    n = int(h["n"])
    m, _, _ = _model(ops, name, d)
    q = _t(d["q"][:n])
    up = _t(d["upstream"][:n]) if "upstream" in d.files else None
    g, H = m.score_hess_raw(q, up)
    H = _n(H)
    H64 = h[name + "/hess64"]
    assert H.shape == H64.shape == (n, q.shape[1], q.shape[1]) and np.abs(H64).max() > 0
    assert relerr(H, H64) < TOL_H
    if name + "/hess32" in h.files:  # the reference's own fp32 route, where its graph is twice differentiable
        H32 = h[name + "/hess32"]
        assert relerr(H, H32) < TOL_H + relerr(H32, H64)
    assert relerr(H, H.transpose(0, 2, 1)) < TOL_H  # symmetric up to round-off (rows are independent lanes)
    # the value part of the duals is the gradient of the same function
    _, g1 = m.score_grad_raw(q, up)
    assert relerr(_n(g), _n(g1)) < 5e-6


@pytest.mark.parametrize("name", sorted(n for n in CASE_ROBOT if not n.startswith("edge_r0")))
def test_hessian_vs_oracle_gradient_differences(ops, name):
    """every transform kind x kernel function of the score fixtures, a ragged batch (B * dof not a multiple of 64),
    random upstream weights for C > 1"""
    d = load(name)
    m, desc, kspec = _model(ops, name, d)
    B = min(len(d["q"]), 37)
    q = d["q"][:B]
    rng = np.random.default_rng(5)
    up = rng.standard_normal((B, m.C)).astype(np.float32) if m.C > 1 else None
    _, H = m.score_hess_raw(_t(q), None if up is None else _t(up))
    sup = d["sup_x32"].reshape(len(d["sup_x32"]), -1).astype(np.float64)
    Ho = _oracle_hess(desc, kspec, sup, d["weights"].astype(np.float64), q, None if up is None else up.astype(np.float64))
    assert relerr(_n(H), Ho) < TOL_H
    if m.C > 1:  # NULL upstream = all ones
        _, H1 = m.score_hess_raw(_t(q))
        _, H2 = m.score_hess_raw(_t(q), torch.ones((B, m.C), device="cuda"))
        assert relerr(_n(H1), _n(H2)) < 2e-6


@pytest.mark.parametrize("name,kspec,C", [("urdf_panda", (1, 1.0, 1.0), 1), ("urdf_fetch_arm", (0, 10.0, 2.0), 1),
                                          ("urdf_allegro", (1, 3.0, 1.0), 3), ("urdf_trifinger", (0, 3.0, 3.0), 2),
                                          ("urdf_jaco", (2, 0.7, 0.0), 1), ("urdf_fetch", (1, 2.0, 1.5), 1),
                                          ("urdf_iiwa7", (0, 10.0, 2.0), 2), ("urdf_iiwa7_allegro", (1, 1.0, 1.0), 1),
                                          ("urdf_iiwa7_allegro", (0, 5.0, 2.0), 3)])
def test_hessian_on_urdf_trees(ops, name, kspec, C):
    """branching trees, mimic and prismatic joints: the tangents go through fk_tree_chain / fk_tree_vjp.  The 23-joint
    iiwa7 + Allegro tree is the one whose frames do not fit the LDS as (value, tangent) pairs: the kernel keeps them in
    global memory for it (hess_kernel.hip, paged frames)"""
    d, rob = load("fk_" + name), urdf_robot(name)
    desc = rob.fk_desc()
    rng = np.random.default_rng(11)
    lim = d["limits"]
    S, B = 150, 9
    sq = (rng.random((S, rob.dof)) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).astype(np.float32)
    q = (rng.random((B, rob.dof)) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).astype(np.float32)
    W = rng.standard_normal((S, C)).astype(np.float32)
    up = rng.standard_normal((B, C)).astype(np.float32) if C > 1 else None
    sup = _n(rob.fkine(_t(sq))).reshape(S, -1)
    m = ops.ScoreModel(desc, *kspec, _t(sup), _t(W))
    _, H = m.score_hess_raw(_t(q), None if up is None else _t(up))
    Ho = _oracle_hess(desc, kspec, sup.astype(np.float64), W.astype(np.float64), q,
                      None if up is None else up.astype(np.float64))
    assert np.abs(Ho).max() > 0
    assert relerr(_n(H), Ho) < TOL_H


def test_hessian_on_a_support_is_finite(ops):
    """Polyharmonic(1) is not twice differentiable where a configuration coincides with a support: that pair is left
    out (include/dcx.h) and everything else stays finite"""
    d = load("edge_r0_baxter_poly1")
    kind, p0, p1 = case_kernel(d)
    desc = desc_for(CASE_ROBOT["edge_r0_baxter_poly1"], dof=d["q"].shape[1])
    sup = ops.fkine(desc, _t(d["sup_q"]))
    m = ops.ScoreModel(desc, kind, p0, p1, sup.reshape(len(sup), -1), _t(d["weights"]))
    g, H = m.score_hess_raw(_t(d["q"]))
    assert torch.isfinite(H).all() and torch.isfinite(g).all()


def test_hessian_empty_batch(ops):
    d = load("cfg1_planar2_rq")
    m, _, _ = _model(ops, "cfg1_planar2_rq", d)
    g, H = m.score_hess_raw(_t(d["q"][:0]))
    assert H.shape == (0, 2, 2) and g.shape == (0, 2)


@pytest.mark.parametrize("name", ["cfg2_baxter_poly1", "cfg3_baxter_rq_c5", "misc_dualpanda_rq", "misc_planar3_poly2"])
@pytest.mark.parametrize("B", [1, 40, 256, 700])
def test_hessian_split_across_blocks_equals_the_unsplit_launch(ops, knob, name, B):
    """small batches split the supports across blocks (the last block to arrive folds the partial rows in a fixed
    order): same Hessian as the one-block-per-tile launch up to the order of the fp32 sums - every batch size, one and
    several classes with an explicit upstream, narrow and wide feature rows, forced block counts - and identical from
    call to call"""
    d = load(name)
    m, desc, kspec = _model(ops, name, d)
    rng = np.random.default_rng(5)
    q = _t(np.repeat(d["q"], -(-B // len(d["q"])), axis=0)[:B] + 0.05 * rng.standard_normal((B, d["q"].shape[1])).astype(np.float32))
    up = _t(rng.standard_normal((B, m.C)).astype(np.float32)) if m.C > 1 else None
    knob("hess_ys", 1)
    g0, H0 = m.score_hess_raw(q, up)
    for ys in (-1, 2, 5, 12):
        knob("hess_ys", ys)
        g1, H1 = m.score_hess_raw(q, up)
        g1b, H1b = m.score_hess_raw(q, up)
        assert torch.equal(H1, H1b) and torch.equal(g1, g1b), ys
        scale = max(float(H0.abs().max()), 1e-30)
        assert float((H1 - H0).abs().max()) / scale < 3e-6 and relerr(_n(g1), _n(g0)) < 3e-6, ys


@pytest.mark.parametrize("name,B", [("cfg2_baxter_poly1", 1024), ("cfg3_baxter_rq_c5", 3000), ("misc_planar3_poly2", 70000),
                                    ("cfg1_planar2_rq", 40000)])
def test_hessian_forms_agree_at_the_batches_the_rule_switches(ops, knob, name, B):
    """the rule (knob -1) takes the moments form from B = 1024 where it is compiled; both forms and the rule agree to fp32
    round-off at those batches - one and several chunks of 32768 configurations, ragged last tiles - and call to call the
    result is identical (the sums are folded in a fixed order)"""
    d = load(name)
    m, desc, kspec = _model(ops, name, d)
    rng = np.random.default_rng(B)
    q = _t(np.repeat(d["q"], -(-B // len(d["q"])), axis=0)[:B] + 0.05 * rng.standard_normal((B, d["q"].shape[1])).astype(np.float32))
    up = _t(rng.standard_normal((B, m.C)).astype(np.float32)) if m.C > 1 else None
    knob("hess_form", 0)
    g0, H0 = m.score_hess_raw(q, up)
    knob("hess_form", 1)
    g1, H1 = m.score_hess_raw(q, up)
    knob("hess_form", -1)
    g2, H2 = m.score_hess_raw(q, up)
    assert torch.equal(H1, H2) and torch.equal(g1, g2)          # the rule's choice at these sizes
    scale = max(float(H0.abs().max()), 1e-30)
    assert float((H1 - H0).abs().max()) / scale < 5e-6 and relerr(_n(g1), _n(g0)) < 3e-6
    _, g = m.score_grad_raw(q, up)
    assert relerr(_n(g1), _n(g)) < 5e-6


def test_hessian_moments_form_on_two_streams(ops, knob):
    """the moments form keeps its sums in a buffer per (model, stream): calls of one model enqueued on two streams at once - different
    batches, interleaved - give what each gives alone"""
    d = load("cfg2_baxter_poly1")
    m, desc, kspec = _model(ops, "cfg2_baxter_poly1", d)
    knob("hess_form", 1)
    rng = np.random.default_rng(9)
    qs = [_t(np.repeat(d["q"], -(-B // len(d["q"])), axis=0)[:B] + 0.05 * rng.standard_normal((B, 7)).astype(np.float32)) for B in (3000, 1700)]
    alone = [m.score_hess_raw(q) for q in qs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for _ in range(20):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i].append(m.score_hess_raw(qs[i]))
    torch.cuda.synchronize()
    for i in range(2):
        for g, H in outs[i]:
            assert torch.equal(H, alone[i][1]) and torch.equal(g, alone[i][0])
