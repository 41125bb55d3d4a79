"""CPU emulation of the bf16x3 split-operand matrix-core formulation of the sweep, to size its
rounding error against the float64 truth BEFORE any kernel is written.

  d2[b, j]  = |x_b|^2 + |s_j|^2 - 2 x_b . s_j      GEMM 1, operands split into three bf16 pieces each (exact: 3 x 8 bits),
                                                   six cross products kept (hh, hm, mh, hl, lh, mm), fp32 accumulation
  gX[b, :]  = x_b * sum_j c_bj - sum_j c_bj s_j    GEMM 2, c split into three bf16 pieces per pair, s pieces as above,
                                                   accumulators flushed into a direct accumulator every FLUSH supports
  near pairs (d2 < tau * (|x|^2 + |s|^2) / 2) take the direct difference form.

Usage: python tools/split_numerics.py            (reads tests/golden/headline_baxter_poly1_s2000.npz; numpy only)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
f32 = np.float32


def trunc_bf16(a):
    """the bf16 obtained by dropping the low 16 bits (what `v_and_b32 0xffff0000` gives), as float32"""
    return (np.ascontiguousarray(a, dtype=f32).view(np.uint32) & np.uint32(0xFFFF0000)).view(f32)


def split3(a):
    a = np.asarray(a, dtype=f32)
    h = trunc_bf16(a)
    r1 = (a - h).astype(f32)
    m = trunc_bf16(r1)
    lo = (r1 - m).astype(f32)          # <= 8 significant bits: a bf16 exactly
    assert np.array_equal(trunc_bf16(lo), lo)
    return h, m, lo


def seq_dot_f32(terms):
    """fp32 accumulation of exact products in the given order: terms [..., K] (float64 exact) -> float32"""
    acc = np.zeros(terms.shape[:-1], dtype=f32)
    for k in range(terms.shape[-1]):
        acc = (acc.astype(np.float64) + terms[..., k]).astype(f32)   # one rounding per accumulate (worst case)
    return acc


def emulate(x, s, w, kind, flush=64, tau=0.02, six_terms=True, block_acc=16):
    """x [B, D], s [S, D], w [S] fp32.  Returns score [B], gX [B, D] of the split formulation (fp32)."""
    B, D = x.shape
    S = s.shape[0]
    xh, xm, xl = split3(x)
    sh, sm, sl = split3(s)
    xx = seq_dot_f32(x.astype(np.float64) ** 2)                        # per lane, once (VALU)
    ss = (s.astype(np.float64) ** 2).sum(1).astype(f32)                # host, rounded once
    # GEMM 1: K order = low-order cross terms first, the hh block last.  Inside one MFMA the K products are summed at
    # (at least) fp32; emulate block-wise exact sums of `block_acc` products followed by one fp32 rounding.
    pieces = [(xm, sm), (xh, sl), (xl, sh), (xh, sm), (xm, sh), (xh, sh)] if six_terms else \
             [(xl, sl), (xm, sl), (xl, sm), (xm, sm), (xh, sl), (xl, sh), (xh, sm), (xm, sh), (xh, sh)]
    acc = np.zeros((B, S), dtype=f32)
    for (a, b) in pieces:
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        for k0 in range(0, D, block_acc):
            blk = a64[:, k0:k0 + block_acc] @ b64[:, k0:k0 + block_acc].T     # exact in float64 (8 x 8 bit products)
            acc = (acc.astype(np.float64) + blk).astype(f32)
    d2 = ((xx[:, None].astype(np.float64) + ss[None, :]).astype(f32).astype(np.float64) - 2.0 * acc).astype(f32)
    thr = (f32(tau) * 0.5 * (xx[:, None] + ss[None, :])).astype(f32)
    near = d2 < thr
    # direct form for near pairs
    delta = (x[:, None, :] - s[None, :, :]).astype(f32)                # [B, S, D] fp32
    d2_direct = seq_dot_f32(delta.astype(np.float64) ** 2)
    d2 = np.where(near, d2_direct, d2)
    if kind == "poly1":
        d2c = np.maximum(d2, f32(1e-30))
        ri = (1.0 / np.sqrt(d2c.astype(np.float64))).astype(f32)
        val = (d2c * ri).astype(f32)
        g = ri
    else:  # rq2, gamma = 10
        t = (f32(5.0) * d2 + f32(1.0)).astype(f32)
        u = (1.0 / t.astype(np.float64)).astype(f32)
        val = (u * u).astype(f32)
        g = (f32(-20.0) * (val * u).astype(f32)).astype(f32)
    score = seq_dot_f32(val.astype(np.float64) * w[None, :].astype(np.float64))
    coef = (g * w[None, :]).astype(f32)
    gx = np.zeros((B, D), dtype=f32)
    # near pairs: direct accumulate, zero coefficient for the GEMM
    if near.any():
        gx = (gx.astype(np.float64) + np.einsum("bs,bsd->bd", np.where(near, coef, 0).astype(np.float64),
                                                 delta.astype(np.float64))).astype(f32)
    coef = np.where(near, f32(0), coef)
    ch, cm, cl = split3(coef)
    for j0 in range(0, S, flush):
        sl_ = slice(j0, min(S, j0 + flush))
        P = np.zeros((B, D), dtype=f32)
        A = np.zeros((B,), dtype=f32)
        # per 16-support K block: six products, each block summed exactly then rounded into the fp32 accumulator
        for k0 in range(sl_.start, sl_.stop, 16):
            kk = slice(k0, min(sl_.stop, k0 + 16))
            for (c, sp) in [(cm, sm), (ch, sl), (cl, sh), (ch, sm), (cm, sh), (ch, sh)]:
                P = (P.astype(np.float64) + c[:, kk].astype(np.float64) @ sp[kk].astype(np.float64)).astype(f32)
            for c in (cl, cm, ch):
                A = (A.astype(np.float64) + c[:, kk].astype(np.float64).sum(1)).astype(f32)
        run = ((x.astype(np.float64) * A[:, None]).astype(f32) - P).astype(f32)   # fma(x, A, -P): one rounding more than fma
        gx = (gx + run).astype(f32)
    return score, gx, near.sum()


def direct_f32(x, s, w, kind):
    """what the VALU sweep computes (direct differences, fp32 sequential)"""
    delta = (x[:, None, :] - s[None, :, :]).astype(f32)
    d2 = seq_dot_f32(delta.astype(np.float64) ** 2)
    if kind == "poly1":
        d2c = np.maximum(d2, f32(1e-30))
        ri = (1.0 / np.sqrt(d2c.astype(np.float64))).astype(f32)
        val, g = (d2c * ri).astype(f32), ri
    else:
        t = (f32(5.0) * d2 + f32(1.0)).astype(f32)
        u = (1.0 / t.astype(np.float64)).astype(f32)
        val = (u * u).astype(f32)
        g = (f32(-20.0) * (val * u).astype(f32)).astype(f32)
    score = seq_dot_f32(val.astype(np.float64) * w[None, :].astype(np.float64))
    coef = (g * w[None, :]).astype(f32)
    gx = np.zeros(x.shape, dtype=f32)
    for j in range(s.shape[0]):
        gx = (gx.astype(np.float64) + coef[:, j:j + 1].astype(np.float64) * delta[:, j, :].astype(np.float64)).astype(f32)
    return score, gx


def truth(x, s, w, kind):
    x, s, w = x.astype(np.float64), s.astype(np.float64), w.astype(np.float64)
    delta = x[:, None, :] - s[None, :, :]
    d2 = (delta ** 2).sum(-1)
    if kind == "poly1":
        r = np.sqrt(d2)
        val = r
        g = np.where(r > 0, 1.0 / np.where(r > 0, r, 1.0), 0.0)
    else:
        t = 1.0 + 5.0 * d2
        val = t ** -2
        g = -20.0 * t ** -3
    return val @ w, np.einsum("bs,bsd->bd", g * w[None, :], delta)


def rel(a, r):
    return float(np.abs(a.astype(np.float64) - r).max() / np.abs(r).max())


def main():
    from oracle import oracle
    from tests import helpers
    d = helpers.load("headline_baxter_poly1_s2000")
    desc = helpers.desc_for("baxter_left")
    B = 96
    x = oracle.fkine(desc, d["q"][:B]).reshape(B, -1).astype(f32)
    s = d["sup_x32"].reshape(2000, -1).astype(f32)
    rng = np.random.default_rng(0)
    # a few near and exact pairs
    x[0] = s[17]
    x[1] = s[400] + f32(1e-4)
    x[2] = s[900] + f32(3e-2) * rng.standard_normal(12).astype(f32)
    cases = {"N(0,1) weights": d["weights"][:, 0].astype(f32),
             "all-positive weights": np.abs(d["weights"][:, 0]).astype(f32),
             "perceptron-like (sign by position)": (np.sign(s[:, 0] - s[:, 0].mean()) * np.abs(d["weights"][:, 0])).astype(f32)}
    for kind in ("poly1", "rq2"):
        for name, w in cases.items():
            st, gt = truth(x, s, w, kind)
            sd, gd = direct_f32(x, s, w, kind)
            print(f"[{kind}] {name}:  direct fp32  score {rel(sd, st):.2e} grad {rel(gd, gt):.2e}")
            for flush in (32, 64, 128, 2000):
                for tau in (0.02,):
                    se, ge, nn = emulate(x, s, w, kind, flush=flush, tau=tau)
                    print(f"      split flush={flush:<5d} tau={tau}: score {rel(se, st):.2e} grad {rel(ge, gt):.2e}   "
                          f"(near pairs {nn}; per-row worst grad {np.abs(ge - gt).max(1).max() / np.abs(gt).max():.2e})")




# ---- the VALU-only expanded form (score_kernel.h "XF" sweep): fp32 fma chains, no matrix cores -----------------------
def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def emulate_valu_expanded(x, s, w, kind, flush=64, tau=0.01):
    """d2 = (xx + ss_j) + sum_k (-2 x_k) s_jk as two interleaved fp32 fma chains (even / odd k: one v_pk_fma per pair of
    features), gradient accumulators acc_k += c s_jk and A += c per row, flushed into gx every `flush` rows; rows with
    d2 < tau * xx take the direct-difference form for that lane only."""
    B, D = x.shape
    S = s.shape[0]
    xm2 = (f32(-2.0) * x).astype(f32)
    xx = seq_dot_f32(x.astype(np.float64) ** 2)
    ss = (s.astype(np.float64) ** 2).sum(1).astype(f32)
    thr = (f32(tau) * xx).astype(f32)
    sc = np.zeros(B, dtype=f32)
    gx = np.zeros((B, D), dtype=f32)
    acc = np.zeros((B, D), dtype=f32)
    A = np.zeros(B, dtype=f32)
    nnear = 0
    for j in range(S):
        a0 = (xx + ss[j]).astype(f32)
        a1 = np.zeros(B, dtype=f32)
        for k in range(0, D - 1, 2):
            a0 = fma32(xm2[:, k], np.full(B, s[j, k], f32), a0)
            a1 = fma32(xm2[:, k + 1], np.full(B, s[j, k + 1], f32), a1)
        d2 = (a0 + a1).astype(f32)
        if D & 1:
            d2 = fma32(xm2[:, D - 1], np.full(B, s[j, D - 1], f32), d2)
        near = d2 < thr
        delta = (x - s[j][None, :]).astype(f32)
        if near.any():
            nnear += int(near.sum())
            d2d = seq_dot_f32(delta.astype(np.float64) ** 2)
            d2 = np.where(near, d2d, d2)
        if kind == "poly1":
            d2c = np.maximum(d2, f32(1e-30))
            ri = (1.0 / np.sqrt(d2c.astype(np.float64))).astype(f32)
            val, g = (d2c * ri).astype(f32), ri
        else:
            t = (f32(5.0) * d2 + f32(1.0)).astype(f32)
            u = (1.0 / t.astype(np.float64)).astype(f32)
            val = (u * u).astype(f32)
            g = (f32(-20.0) * (val * u).astype(f32)).astype(f32)
        sc = fma32(np.full(B, w[j], f32), val, sc)
        coef = (g * w[j]).astype(f32)
        if near.any():
            gx = np.where(near[:, None], fma32(coef[:, None], delta, gx), gx)
            coef = np.where(near, f32(0), coef)
        acc = fma32(coef[:, None], np.broadcast_to(s[j][None, :], (B, D)).astype(f32), acc)
        A = (A + coef).astype(f32)
        if (j + 1) % flush == 0 or j == S - 1:
            gx = (fma32(x, A[:, None], gx) - acc).astype(f32)
            acc[:] = 0
            A[:] = 0
    return sc, gx, nnear


def main_valu():
    from oracle import oracle
    from tests import helpers
    d = helpers.load("headline_baxter_poly1_s2000")
    desc = helpers.desc_for("baxter_left")
    B = 128
    x = oracle.fkine(desc, d["q"][:B]).reshape(B, -1).astype(f32)
    s = d["sup_x32"].reshape(2000, -1).astype(f32)
    rng = np.random.default_rng(0)
    x[0] = s[17]
    x[1] = s[400] + f32(1e-4)
    x[2] = s[900] + f32(3e-2) * rng.standard_normal(12).astype(f32)
    x[3] = s[901] + f32(8e-2) * rng.standard_normal(12).astype(f32)
    x[4] = s[902] + f32(5e-2) * rng.standard_normal(12).astype(f32)
    w0 = d["weights"][:, 0].astype(f32)
    cases = {"N(0,1) weights": w0, "all-positive weights": np.abs(w0),
             "sign by position": (np.sign(s[:, 0] - s[:, 0].mean()) * np.abs(w0)).astype(f32)}
    print("\n== VALU-only expanded form ==")
    for kind in ("poly1", "rq2"):
        for name, w in cases.items():
            st, gt = truth(x, s, w, kind)
            sd, gd = direct_f32(x, s, w, kind)
            print(f"[{kind}] {name}:  direct fp32  score {rel(sd, st):.2e} grad {rel(gd, gt):.2e}")
            for flush, tau in ((64, 0.01), (128, 0.01), (64, 0.003), (2000, 0.01)):
                se, ge, nn = emulate_valu_expanded(x, s, w, kind, flush=flush, tau=tau)
                print(f"      expanded flush={flush:<5d} tau={tau}: score {rel(se, st):.2e} grad {rel(ge, gt):.2e}  (near {nn})")
    # few supports: one near pair dominates the output
    for S in (8, 32):
        w = w0[:S]
        st, gt = truth(x[:8], s[900:900 + S], w, "poly1")
        for tau in (0.01, 0.003):
            se, ge, nn = emulate_valu_expanded(x[:8], s[900:900 + S], w, "poly1", tau=tau)
            print(f"  S={S} tau={tau}: score {rel(se, st):.2e} grad {rel(ge, gt):.2e} (near {nn})")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "valu":
    main_valu()

if __name__ == "__main__" and len(sys.argv) == 1:
    main()
