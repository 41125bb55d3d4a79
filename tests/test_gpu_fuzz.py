"""Shape fuzz of the fused sweep against the fp64 oracle: every compiled feature width (through keypoint bodies,
planar arms and the identity transform), class counts 1..8, all kernel families, supports that do and do not divide
among the waves, batches with ragged tiles.  Exercises the width-dependent scalar pipelines (whole rows / two rows /
half rows, weights straddling the halves) and both cross-wave fold modes."""
import zlib

import numpy as np
import pytest
import torch

from helpers import relerr

pytestmark = pytest.mark.gpu

KERNELS = [(0, 10.0, 2.0), (1, 1.0, 1.0), (0, 3.0, 3.0), (1, 3.0, 2.0), (1, 2.0, 1.0), (2, 0.7, 0.0)]


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32, device="cuda")


def _n(t):
    return t.detach().cpu().numpy()


def _desc(kind, D, rng):
    from diffco_amd import _fkdesc as fd
    if kind == "none":
        return fd.none_desc(D), D
    if kind == "planar":  # D = 2 * dof
        return fd.planar_desc((0.2 + rng.random(D // 2)).tolist()), D // 2
    kp = rng.uniform(-0.5, 0.5, (D // 3, 3))  # SE(3) body with D / 3 keypoints
    return fd.keypoint_desc(kp, 3), 6


CASES = []
_rng = np.random.default_rng(2024)
for D in (2, 3, 5, 7, 9, 12, 13, 14, 16, 17, 18, 19, 21, 23, 24, 25, 27, 30, 31, 32):
    CASES.append(("none", D, int(_rng.integers(1, 9)), int(_rng.integers(len(KERNELS)))))
for D in (4, 8, 18, 22, 26, 36, 38, 42, 44, 48, 54, 60, 62, 64):
    CASES.append(("planar", D, int(_rng.integers(1, 9)), int(_rng.integers(len(KERNELS)))))
for D in (27, 33, 39, 45, 51, 57, 63, 66, 69, 72, 75, 78, 84, 87, 90, 96):
    CASES.append(("se3", D, int(_rng.integers(1, 9)), int(_rng.integers(len(KERNELS)))))
for C in range(1, 9):  # every class count at a half-row width and at a two-row width
    CASES.append(("planar", 40, C, C % len(KERNELS)))
    CASES.append(("none", 20, C, (C + 1) % len(KERNELS)))


@pytest.mark.parametrize("kind,D,C,ki", CASES)
def test_fused_sweep_shapes(kind, D, C, ki):
    from diffco_amd import _ops
    from oracle import oracle
    seed = zlib.crc32(repr((kind, D, C, ki)).encode())  # deterministic across processes
    rng = np.random.default_rng(seed)
    desc, dof = _desc(kind, D, rng)
    kern = KERNELS[ki]
    S = int(rng.choice([37, 150, 333, 1000]))
    B = int(rng.choice([1, 63, 130, 700, 3000]))
    lo, hi = (-1.5, 1.5)
    sq = rng.uniform(lo, hi, (S, dof)).astype(np.float32)
    q = rng.uniform(lo, hi, (B, dof)).astype(np.float32)
    W = rng.standard_normal((S, C)).astype(np.float32)
    W[rng.random((S, C)) < 0.2] = 0.0
    up = rng.standard_normal((B, C)).astype(np.float32)
    sup = _n(_ops.fkine(desc, _t(sq))).reshape(S, -1)
    assert sup.shape[1] == D
    m = _ops.ScoreModel(desc, *kern, _t(sup), _t(W))
    rs, rg, rj = oracle.score_grad(desc, *kern, sup, W, q, up if C > 1 else None, want_jac=True, dtype=np.float64)
    s, g = m.score_grad_raw(_t(q), _t(up) if C > 1 else None)
    assert relerr(_n(s), rs) < 1e-5, (kind, D, C, kern, S, B)
    assert relerr(_n(g), rg) < 2e-5, (kind, D, C, kern, S, B)
    assert relerr(_n(m.score_raw(_t(q))), rs) < 1e-5
    if C > 1:  # row-sum (all-ones upstream) path and the Jacobian
        s1, g1 = m.score_grad_raw(_t(q), None)
        _, rg1, _ = oracle.score_grad(desc, *kern, sup, W, q, None, dtype=np.float64)
        assert relerr(_n(g1), rg1) < 2e-5
    if B <= 700:
        _, jac = m.score_jac_raw(_t(q))
        assert relerr(_n(jac), rj) < 2e-5


HESS_CASES = []
for D in (2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 14, 15, 16):          # widths that land on the compiled 2 .. 16 (odd ones are padded)
    HESS_CASES.append(("none", D, int(_rng.integers(1, 9)), int(_rng.integers(len(KERNELS)))))
for D in (4, 8, 12, 16):
    HESS_CASES.append(("planar", D, int(_rng.integers(1, 9)), int(_rng.integers(len(KERNELS)))))
for D in (6, 9, 12, 15):
    HESS_CASES.append(("se3", D, int(_rng.integers(1, 6)), int(_rng.integers(len(KERNELS)))))


@pytest.mark.parametrize("kind,D,C,ki", HESS_CASES)
def test_hessian_moments_form_shapes(kind, D, C, ki):
    """shape fuzz of dcx_score_hess's moments form (hess_kernel.hip hess_moments_kernel) against the lanes form - which the reference's
    double backward and the float64 oracle pin (tests/test_gpu_hess.py): every width the form is compiled for, class counts 1 .. 8
    with and without an upstream, all kernel families (the generic ones through KF_GEN), supports that do not fill the block's
    waves, ragged batches, forced splits of the supports"""
    from diffco_amd import _lib, _ops
    lib = _lib.load()
    seed = zlib.crc32(repr(("hess", kind, D, C, ki)).encode())
    rng = np.random.default_rng(seed)
    desc, dof = _desc(kind, D, rng)
    kern = KERNELS[ki]
    S = int(rng.choice([5, 37, 150, 333, 1000]))
    B = int(rng.choice([1, 63, 130, 700, 3000]))
    sq = rng.uniform(-1.5, 1.5, (S, dof)).astype(np.float32)
    q = _t(rng.uniform(-1.5, 1.5, (B, dof)).astype(np.float32))
    W = rng.standard_normal((S, C)).astype(np.float32)
    W[rng.random((S, C)) < 0.2] = 0.0
    W[0, :] = 1.0   # (at least one active support)
    sup = _ops.fkine(desc, _t(sq)).reshape(S, -1)
    m = _ops.ScoreModel(desc, *kern, sup, _t(W))
    ups = [None] + ([_t(rng.standard_normal((B, C)).astype(np.float32))] if C > 1 or rng.random() < 0.5 else [])
    try:
        for up in ups:
            _lib.check(lib.dcx_debug_set(b"hess_form", 0))
            g0, H0 = m.score_hess_raw(q, up)
            for ys in (-1, 3):
                _lib.check(lib.dcx_debug_set(b"hess_form", 1))
                _lib.check(lib.dcx_debug_set(b"hess_ys", ys))
                g1, H1 = m.score_hess_raw(q, up)
                lib.dcx_debug_set(b"hess_ys", -1)
                scale = max(float(H0.abs().max()), 1e-30)
                assert float((H1 - H0).abs().max()) / scale < 1e-5, (kind, D, C, kern, S, B, ys, up is None)
                assert relerr(_n(g1), _n(g0)) < 5e-6, (kind, D, C, kern, S, B, ys, up is None)
    finally:
        lib.dcx_debug_set(b"hess_form", -1)
        lib.dcx_debug_set(b"hess_ys", -1)
