#!/usr/bin/env python3
"""Soak of the escape loop (dcx_escape_adam): random robots / kernels / class counts / batch sizes / options, every call checked
against a float64 numpy restatement of the loop on the C oracle's score and gradient (evaluation counts exact away from ties,
configurations to 1e-4), and the compacted call (compact_every = 1 .. 5) against the un-compacted one.

    python tools/soak_escape.py [rounds]        (GPU box; prints one line per round and `SOAK OK` / the first mismatch)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from helpers import make_robot
    from diffco_amd import _ops, utils
    from diffco_amd.escape import OptimSampler
    from oracle import oracle
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(20250930)
    robots = ["baxter_left", "panda", "planar3", "planar7", "se2", "se3", "baxter_dual"]
    kernels = [(1, 1.0, 1.0), (0, 0.5, 2.0), (0, 2.0, 2.0), (2, 0.8, 0.0)]
    bad = 0
    for r in range(rounds):
        name = robots[int(rng.integers(len(robots)))]
        kspec = kernels[int(rng.integers(len(kernels)))]
        rob = make_robot(name)
        desc = rob.fk_desc()
        lim = rob.limits.numpy().astype(np.float64)
        S, C = int(rng.choice([1, 7, 60, 333, 1200])), int(rng.choice([1, 1, 2, 3, 5]))
        B = int(rng.choice([1, 2, 9, 64, 65, 700, 5000]))
        n, rf = int(rng.integers(1, 16)), int(rng.choice([0, 1, 2, 5]))
        lr = float(rng.choice([0.02, 0.1, 0.3]))
        wrap = bool(rng.integers(2))
        sup_q = rng.uniform(lim[:, 0], lim[:, 1], (S, len(lim))).astype(np.float32)
        W = (rng.standard_normal((S, C)) * 0.2 + 0.05).astype(np.float32)
        q0 = rng.uniform(lim[:, 0], lim[:, 1], (B, len(lim))).astype(np.float32)
        sup = _ops.fkine(desc, torch.from_numpy(sup_q).cuda()).reshape(S, -1)
        model = _ops.ScoreModel(desc, *kspec, sup, torch.from_numpy(W).cuda())
        s0, _ = model.score_grad_raw(torch.from_numpy(q0).cuda())
        margin = (s0.median(dim=0).values - float(rng.choice([0.0, 0.05, 0.5]))).cpu()
        opts = {"N_WAYPOINTS": n, "safety_margin": margin, "lr": lr, "record_freq": rf or None,
                "post_transform": utils.wrap2pi if wrap else None}
        sampler = OptimSampler(rob, model, opts)
        final, checks, hist, n_rec = sampler.optim_escape_batch(torch.from_numpy(q0), history=True, compact_every=0)
        # float64 restatement on the oracle
        sup64 = oracle.fkine(desc, sup_q.astype(np.float64), dtype=np.float64).reshape(S, -1)
        q, m, v = q0.astype(np.float64), np.zeros((B, len(lim))), np.zeros((B, len(lim)))
        alive, evals, near = np.ones(B, bool), np.zeros(B, np.int64), np.zeros(B, bool)
        mg = margin.numpy().astype(np.float64)
        for t in range(1, n + 1):
            s, g, _ = oracle.score_grad(desc, *kspec, sup64, W.astype(np.float64), q, dtype=np.float64)
            ex = (s - mg[None, :]).sum(axis=1)
            near |= alive & (np.abs(ex) < 1e-4 * (1.0 + np.abs(s).sum(axis=1)))
            evals += alive
            alive &= ex > 0
            m = np.where(alive[:, None], 0.9 * m + 0.1 * g, m)
            v = np.where(alive[:, None], 0.999 * v + 0.001 * g * g, v)
            qn = q - lr / (1 - 0.9 ** t) * m / (np.sqrt(v) / np.sqrt(1 - 0.999 ** t) + 1e-8)
            if wrap:
                qn = (np.pi + qn) % (2 * np.pi) - np.pi
            q = np.where(alive[:, None], qn, q)
        ok = ~near
        e_cnt = int((checks.numpy()[ok] != evals[ok]).sum())
        den = max(1.0, float(np.abs(q).max()))
        # (a wrapped coordinate next to +-pi may land on the other side: compare on the circle)
        dq = final.numpy().astype(np.float64) - q
        if wrap:
            dq = (dq + np.pi) % (2 * np.pi) - np.pi
        same = ok & (checks.numpy() == evals)
        e_q = float(np.abs(dq[same]).max() / den) if same.any() else 0.0
        # the compacted call
        k = int(rng.integers(1, 6))
        f2, c2, h2, n2 = sampler.optim_escape_batch(torch.from_numpy(q0), history=True, compact_every=k)
        agree = (c2 == checks)
        e_c = float((~agree).float().mean())
        e_f = float((f2[agree] - final[agree]).abs().max()) if bool(agree.any()) else 0.0
        e_h = float((h2[:, agree] - hist[:, agree]).abs().max()) if bool(agree.any()) else 0.0
        line = (f"round {r:3d} {name:<12} kernel {kspec} S {S:5d} C {C} B {B:5d} steps {n:2d} rf {rf} wrap {int(wrap)} k {k}: counts off {e_cnt} "
                f"(of {int(ok.sum())}), |q - q64| {e_q:.1e}, compacted: counts differ {e_c:.4f}, |dq| {e_f:.1e}, |dhist| {e_h:.1e}")
        fail = e_cnt > 0 or e_q > 1e-4 or e_c > 0.01 or e_f > 1e-4 or e_h > 1e-4
        print(line + ("   <-- MISMATCH" if fail else ""), flush=True)
        bad += fail
    print("SOAK OK" if bad == 0 else f"SOAK FAILED: {bad} rounds")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
