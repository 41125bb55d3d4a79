"""Pins the CPU oracle (oracle/) against outputs of the reference itself (tests/golden, generated
by tools/make_golden.py) and against the known answers recorded in SURVEY.md §8a/§8c.  CPU only."""
import numpy as np
import pytest

from helpers import CASE_ROBOT, FK_NAMES, KIND, case_kernel, desc_for, load, relerr
from oracle import oracle


@pytest.mark.parametrize("name", FK_NAMES)
def test_fk_matches_reference(name):
    d = load("fk_" + name)
    desc = desc_for(name)
    x64 = oracle.fkine(desc, d["q"], np.float64)
    x32 = oracle.fkine(desc, d["q"], np.float32)
    assert relerr(x64, d["x64"]) < 1e-12
    assert relerr(x32, d["x64"]) < 2e-6
    assert relerr(d["x32"], d["x64"]) < 2e-6  # the reference's own fp32 FK, for scale
    gq = oracle.fkine_vjp(desc, d["q"], d["gx"], np.float64)
    # 1e-7, not 1e-11: the geometric Jacobian z x (p - o) assumes orthonormal frames, while the
    # reference's fp32 base rotations / sin-cos(alpha) tables are orthonormal only to ~1e-8
    assert relerr(gq, d["gq64"]) < 1e-7
    assert relerr(oracle.fkine_vjp(desc, d["q"], d["gx"], np.float32), d["gq64"]) < 5e-6


def test_fk_known_answers():
    # SURVEY.md §8a rows a8, a9, a11
    z = np.zeros((1, 7))
    b = oracle.fkine(desc_for("baxter_left"), z, np.float64)[0]
    np.testing.assert_allclose(b, [[0.069, 0, 0.27035], [0.43335, 0, 0.20135], [0.80764, 0, 0.19135],
                                   [1.19499, 0, 0.19135]], atol=2e-6)
    q1 = np.array([[.1, -.2, .3, .4, -.5, .6, -.7]])
    b1 = oracle.fkine(desc_for("baxter_left"), q1, np.float64)[0]
    np.testing.assert_allclose(b1[3], [1.0714469, 0.1578894, -0.0682933], atol=2e-6)
    p = oracle.fkine(desc_for("panda"), z, np.float64)[0]
    np.testing.assert_allclose(p, [[0, 0, .333], [.0825, 0, .649], [0, 0, .649], [0, 0, 1.033], [.088, 0, .819],
                                   [.088, -.107, .819], [.088, .107, .819]], atol=2e-6)
    p1 = oracle.fkine(desc_for("panda"), q1, np.float64)[0]
    np.testing.assert_allclose(p1[4], [-0.0237762, -0.072792, 0.9254279], atol=2e-6)
    np.testing.assert_allclose(p1[5], [0.0015334, -0.1669738, 0.9694532], atol=2e-6)
    np.testing.assert_allclose(p1[6], [-0.0490858, 0.0213897, 0.8814025], atol=2e-6)
    from diffco_amd import _fkdesc
    pl = oracle.fkine(_fkdesc.planar_desc([1.0, 1.0]), np.array([[.5, -1.]]), np.float64)[0]
    np.testing.assert_allclose(pl, [[0.87758255, 0.47942555], [1.7551651, 0.0]], atol=1e-7)


def test_kernels_match_reference():
    d = load("kernels")
    kinds, params = d["kernel_kinds"], d["kernel_params"]
    for D in (4, 12, 21, 6):
        x, s = d[f"x_D{D}"], d[f"s_D{D}"]
        for i, (kind, p) in enumerate(zip(kinds, params)):
            k64 = oracle.kernel_matrix(KIND[str(kind)], p[0], p[1], x, s, np.float64)
            assert relerr(k64, d[f"k64_D{D}_{i}"]) < 1e-12, (D, i)
            k32 = oracle.kernel_matrix(KIND[str(kind)], p[0], p[1], x, s, np.float32)
            assert relerr(k32, d[f"k64_D{D}_{i}"]) < 3e-6, (D, i)
            # scale: the reference fp32 values (cdist GEMM form) are 1e-5..5e-5 from truth, worst at r = 0
            assert relerr(d[f"k32_D{D}_{i}"], d[f"k64_D{D}_{i}"]) < 2e-4, (D, i)
    # known answers SURVEY.md §8c
    a, b = np.zeros((1, 1)), np.array([[1.0], [2.0]])
    np.testing.assert_allclose(oracle.kernel_matrix(0, 10, 2, a, b, np.float64)[0], [0.02777778, 0.00226757], rtol=3e-6)
    np.testing.assert_allclose(oracle.kernel_matrix(1, 1, 1, a, b, np.float64)[0], [1, 2])
    np.testing.assert_allclose(oracle.kernel_matrix(1, 3, 2, a, b, np.float64)[0], [0.5, 4])
    np.testing.assert_allclose(oracle.kernel_matrix(1, 2, 1, a, b, np.float64)[0], [0, 2.7725887], rtol=1e-7)
    for key, ref in (("known_rq10", [0.02777778, 0.00226757]), ("known_poly11", [1, 2]), ("known_poly32", [0.5, 4]),
                     ("known_poly21", [0, 2.7725887])):
        np.testing.assert_allclose(d[key], ref, rtol=2e-6)


@pytest.mark.parametrize("name", sorted(CASE_ROBOT))
def test_score_grad_matches_reference(name):
    d = load(name)
    kind, p0, p1 = case_kernel(d)
    desc = desc_for(CASE_ROBOT[name], dof=d["q"].shape[1])
    W = d["weights"]
    # fp64 oracle on fp64-FK supports == the fp64 referee (tight): same algorithm, independent code
    sup64 = oracle.fkine(desc, d["sup_q"], np.float64)
    s64, g64, _ = oracle.score_grad(desc, kind, p0, p1, sup64, W, d["q"], dtype=np.float64)
    assert relerr(s64, d["score64"].reshape(s64.shape)) < 1e-11
    assert relerr(g64, d["grad64"]) < 1e-7  # frames orthonormal only to ~1e-8, see test_fk_matches_reference
    # fp32 oracle (supports = the reference's fp32 support_transformed) vs referee and vs the reference's fp32
    sup32 = d["sup_x32"]
    if name.startswith("edge_r0"):
        # an exact coincidence needs supports and queries pushed through the SAME fp32 FK (as the
        # product does: support_transformed comes from the same device FK as the queries)
        sup32 = oracle.fkine(desc, d["sup_q"], np.float32)
    s32, g32, jac = oracle.score_grad(desc, kind, p0, p1, sup32, W, d["q"], want_jac="jac32" in d.files)
    assert relerr(s32, d["score64"].reshape(s32.shape)) < 1e-5
    assert relerr(g32, d["grad64"]) < 1e-5
    # vs the reference's own fp32 output: within 1e-5 plus the reference's own distance from the
    # truth (its cdist GEMM form is up to ~1e-4 off when coordinates are large, e.g. the SE(3) case)
    ref_s = relerr(d["score32"].reshape(s32.shape), d["score64"].reshape(s32.shape))
    ref_g = relerr(d["grad32"], d["grad64"])
    assert relerr(s32, d["score32"].reshape(s32.shape)) < 1e-5 + ref_s
    assert relerr(g32, d["grad32"]) < 1e-5 + ref_g
    if "upstream" in d.files:
        _, gv, _ = oracle.score_grad(desc, kind, p0, p1, sup64, W, d["q"], upstream=d["upstream"], dtype=np.float64)
        assert relerr(gv, d["vjp64"]) < 1e-7
        _, gv32, _ = oracle.score_grad(desc, kind, p0, p1, d["sup_x32"], W, d["q"], upstream=d["upstream"])
        assert relerr(gv32, d["vjp32"]) < 1e-5 + relerr(d["vjp32"], d["vjp64"])
        nj = d["jac32"].shape[0]
        assert relerr(jac[:nj], d["jac32"]) < 1e-5 + ref_g
        np.testing.assert_allclose(jac.sum(1), g32, rtol=0, atol=1e-4 * np.abs(g32).max())


def test_r0_subgradient_is_zero():
    # query == support under the kinked kernel: that pair contributes 0 to the gradient (cdist backward)
    d = load("edge_r0_baxter_poly1")
    desc = desc_for("baxter_left")
    kind, p0, p1 = case_kernel(d)
    sup = oracle.fkine(desc, d["sup_q"], np.float32)  # same fp32 FK as the query -> r == 0 exactly
    W = d["weights"].copy()
    _, g_all, _ = oracle.score_grad(desc, kind, p0, p1, sup, W, d["q"][3:4])
    W[11] = 0  # drop the coincident support entirely
    _, g_wo, _ = oracle.score_grad(desc, kind, p0, p1, sup, W, d["q"][3:4])
    assert relerr(g_all, g_wo) < 1e-6


def test_oracle_is_sanitizer_clean():
    """ASan + UBSan pass over every oracle entry point (all FK kinds x all kernel kinds, r = 0 pair included)"""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0 and "oracle sanitize run ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
