"""GPU tests of the escape loop (SURVEY.md §8f-2 variant, reference scripts/escape.py:19-38): `OptimSampler.optim_escape` as ONE
library call (dcx_escape_adam) against records made by running the REFERENCE's OptimSampler on reference checkers
(tests/golden/escape.npz, tools/make_golden.py gen_escape), against the host loop on the same HIP score, and a float64
restatement of the loop on the C oracle's score and gradient.

Tolerance: the configurations after <= 20 Adam steps, max|a - ref| / max|ref| <= 1e-4 against the reference's fp32 run (Adam
divides by the gradient's running magnitude: a step is ~lr whatever the gradient's size, and a 1e-6 difference in the gradient
moves a configuration by ~lr * 1e-6 per step); evaluation counts and record counts are exact."""
import numpy as np
import pytest
import torch

from helpers import load, make_robot, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _np(t):
    return t.detach().cpu().numpy()


def _baxter(d):
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    rob = make_robot("baxter_left")
    dc = DiffCo(transform=rob.fkine)
    dc.support_points = torch.from_numpy(d["bx_sup_q"])
    dc.support_transformed = rob.fkine(dc.support_points)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["bx_w"])
    return rob, dc


def _planar(d):
    from diffco_amd import MultiDiffCo, kernel
    rob = make_robot("planar3")
    md = MultiDiffCo(None, kernel_func=kernel.FKKernel(rob.fkine, kernel.RQKernel(10.0)))
    md.fkine, md.support_points = rob.fkine, torch.from_numpy(d["pl_sup_q"])
    md.support_fkine = rob.fkine(md.support_points).reshape(len(md.support_points), -1)
    md.rbf_kernel, md.rbf_nodes, md.num_class = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["pl_w"]), 3
    return rob, md


def _se2(d):
    from diffco_amd import kernel
    from diffco_amd.kernel_perceptrons import DiffCo
    rob = make_robot("se2")
    g, p = (float(v) for v in d["se2_kparams"])
    dc = DiffCo(kernel_func=kernel.RQKernel(g, int(p)), transform=rob.fkine)
    dc.support_points = torch.from_numpy(d["se2_sup_q"])
    dc.support_transformed = rob.fkine(dc.support_points)
    dc.gains = torch.from_numpy(d["se2_w"])
    return rob, dc


def test_scores_at_the_starts_match_the_reference():
    d = load("escape")
    _, dc = _baxter(d)
    assert relerr(_np(dc.poly_score(torch.from_numpy(d["bx_starts"]))), d["bx_score0"]) < 1e-5
    _, md = _planar(d)
    assert relerr(_np(md.rbf_score(torch.from_numpy(d["pl_starts"]))), d["pl_score0"]) < 1e-5
    _, dc2 = _se2(d)
    assert relerr(_np(dc2.score(torch.from_numpy(d["se2_starts"]))), d["se2_score0"]) < 1e-5


@pytest.mark.parametrize("tag,rows,args", [
    ("bx_single", slice(0, 1), {"N_WAYPOINTS": 20, "lr": 5e-2, "record_freq": 1}),
    ("bx_flat", 0, {"N_WAYPOINTS": 20, "lr": 5e-2, "record_freq": 3}),
    ("bx_three", slice(0, 3), {"N_WAYPOINTS": 12, "lr": 2e-2, "record_freq": 2,
                               "opt_args": {"lr": 2e-2, "betas": (0.8, 0.99), "eps": 1e-6}}),
])
def test_one_loop_reproduces_the_reference_record(tag, rows, args):
    """escape.py:19-38 as the reference runs it: the whole start_cfg is one loop (2-D and 1-D starts, a loop that stops and
    one that runs out of steps, non-default Adam options)"""
    from diffco_amd.escape import OptimSampler
    d = load("escape")
    rob, dc = _baxter(d)
    margin = float(d["bx_margin3" if tag == "bx_three" else "bx_margin1"])
    start = torch.from_numpy(d["bx_starts"])[rows]
    sampler = OptimSampler(rob, dc.poly_score, dict(args, safety_margin=margin))
    hist, checks = sampler.optim_escape(start)
    assert sampler.last_route == "fused"
    ref = d[tag + "_hist"]
    assert checks == int(d[tag + "_checks"]) and tuple(hist.shape) == ref.shape and hist.dtype == start.dtype
    assert not hist.is_cuda
    assert relerr(_np(hist), ref) < TOL
    # the host loop (a post_transform the library does not know: identity) on the same HIP score agrees with both
    host = OptimSampler(rob, dc.poly_score, dict(args, safety_margin=margin, post_transform=lambda x: x))
    hh, hc = host.optim_escape(start)
    assert host.last_route == "host" and hc == checks and hh.shape == hist.shape
    assert relerr(_np(hh), ref) < TOL and relerr(_np(hh), _np(hist)) < TOL
    # CUDA starts stay on the device
    hist_c, checks_c = sampler.optim_escape(start.cuda())
    assert hist_c.is_cuda and checks_c == checks and torch.equal(hist_c.cpu(), hist)


def _check_batch(sampler, starts, d, tag):
    final, checks, hist, n_rec = sampler.optim_escape_batch(starts, history=True)
    assert sampler.last_route == "fused"
    np.testing.assert_array_equal(_np(checks), d[tag + "_checks"])
    np.testing.assert_array_equal(_np(n_rec), d[tag + "_nrec"])
    assert relerr(_np(final), d[tag + "_final"]) < TOL
    ref_hist = d[tag + "_hist"]
    assert hist.shape[0] >= ref_hist.shape[0]
    assert relerr(_np(hist[:ref_hist.shape[0]]), ref_hist) < TOL
    assert torch.equal(hist[ref_hist.shape[0]:], final[None].expand(hist.shape[0] - ref_hist.shape[0], -1, -1))
    # without the records: the same configurations, bit for bit
    f2, c2 = sampler.optim_escape_batch(starts)
    assert torch.equal(f2, final) and torch.equal(c2, checks)
    return final, checks, hist, n_rec


def test_independent_loops_baxter():
    from diffco_amd.escape import OptimSampler
    d = load("escape")
    rob, dc = _baxter(d)
    starts = torch.from_numpy(d["bx_starts"])
    sampler = OptimSampler(rob, dc.poly_score, {"N_WAYPOINTS": 15, "safety_margin": float(d["bx_marginb"]) - 0.05, "lr": 5e-2,
                                                "record_freq": 4})
    final, checks, hist, n_rec = _check_batch(sampler, starts, d, "bx_batch")
    # a row of the batch is the loop optim_escape runs for that row alone: same launches, same arithmetic
    for b in (0, 5, 9):
        h1, c1 = sampler.optim_escape(starts[b:b + 1])
        assert c1 == int(checks[b]) and len(h1) == int(n_rec[b])
        assert torch.equal(h1[:, 0], hist[:len(h1), b])


def test_independent_loops_three_classes_wrap2pi():
    """scripts/compare_sampling.py:177-195: three Adam steps at lr 0.2 with wrap2pi on random configurations, last
    configuration only; per-class margins (scripts/2d_escape.py:98-107)"""
    from diffco_amd import utils
    from diffco_amd.escape import OptimSampler
    d = load("escape")
    rob, md = _planar(d)
    starts, margin = torch.from_numpy(d["pl_starts"]), torch.from_numpy(d["pl_margin"])
    opts = {"N_WAYPOINTS": 3, "safety_margin": margin, "lr": 0.2, "record_freq": None, "post_transform": utils.wrap2pi,
            "optimizer": torch.optim.Adam}
    final, *_ = _check_batch(OptimSampler(rob, md.rbf_score, opts), starts, d, "pl_batch")
    moved = _np(final) != d["pl_starts"]
    assert np.all(np.abs(_np(final)[moved]) <= np.pi + 1e-6)          # every configuration that moved was wrapped
    _check_batch(OptimSampler(rob, md.rbf_score, dict(opts, N_WAYPOINTS=20, record_freq=1, lr=0.1)), starts[:6], d, "pl_long")


def test_five_classes_on_baxter():
    """config #3's shape - old MultiDiffCo.rbf_score, five classes with zeroed weights, per-class margins - on the DH arm, wrap2pi,
    records every second step: B independent loops and one loop over four configurations, against the reference's records"""
    from diffco_amd import MultiDiffCo, kernel, utils
    from diffco_amd.escape import OptimSampler
    d = load("escape")
    rob = make_robot("baxter_left")
    md = MultiDiffCo(None, kernel_func=kernel.FKKernel(rob.fkine, kernel.RQKernel(10.0)))
    md.fkine, md.support_points = rob.fkine, torch.from_numpy(d["b5_sup_q"])
    md.support_fkine = rob.fkine(md.support_points).reshape(len(md.support_points), -1)
    md.rbf_kernel, md.rbf_nodes, md.num_class = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["b5_w"]), 5
    starts, margin = torch.from_numpy(d["b5_starts"]), torch.from_numpy(d["b5_margin"])
    assert relerr(_np(md.rbf_score(starts)), d["b5_score0"]) < 1e-5
    opts = {"N_WAYPOINTS": 10, "safety_margin": margin, "lr": 5e-2, "record_freq": 2, "post_transform": utils.wrap2pi}
    _check_batch(OptimSampler(rob, md.rbf_score, opts), starts, d, "b5_batch")
    joint = OptimSampler(rob, md.rbf_score, dict(opts, N_WAYPOINTS=6, record_freq=1))
    hist, checks = joint.optim_escape(starts[:4])
    assert joint.last_route == "fused" and checks == int(d["b5_joint_checks"])
    assert tuple(hist.shape) == d["b5_joint_hist"].shape and relerr(_np(hist), d["b5_joint_hist"]) < TOL


def test_independent_loops_se2_wrap():
    from diffco_amd import utils
    from diffco_amd.escape import OptimSampler
    d = load("escape")
    rob, dc = _se2(d)
    starts = torch.from_numpy(d["se2_starts"])
    sampler = OptimSampler(rob, dc.score, {"N_WAYPOINTS": 10, "safety_margin": float(d["se2_margin"]), "lr": 0.1,
                                           "record_freq": 3, "post_transform": utils.se2_wrap2pi})
    final, checks, *_ = _check_batch(sampler, starts, d, "se2_batch")
    went = _np(checks) > 1
    assert np.all(np.abs(_np(final)[went, 2]) <= np.pi + 1e-6) and np.abs(_np(final)[:, :2]).max() > np.pi


def test_float64_restatement_on_the_oracle_score():
    """300 independent loops on a model the golden file does not hold: the loop restated in float64 numpy on the C oracle's
    fp64 score and gradient (the checker of every parity test), evaluation counts exact where the oracle's excess is not within
    1e-5 of zero at any step"""
    from diffco_amd import kernel
    from diffco_amd.escape import OptimSampler
    from diffco_amd.kernel_perceptrons import DiffCo
    from oracle import oracle
    rob = make_robot("panda")
    desc = rob.fk_desc()
    g = torch.Generator().manual_seed(41)
    lim = rob.limits
    sup_q = torch.rand((500, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    w = torch.randn(500, generator=g) * 0.03 + 0.002
    dc = DiffCo(transform=rob.fkine)
    dc.support_points, dc.support_transformed = sup_q, rob.fkine(sup_q)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), w
    starts = torch.rand((300, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    s0 = dc.poly_score(starts)
    margin, n, lr = float(s0.median()) - 0.03, 12, 4e-2
    final, checks = OptimSampler(rob, dc.poly_score, {"N_WAYPOINTS": n, "safety_margin": margin, "lr": lr}).optim_escape_batch(starts)

    sup64 = oracle.fkine(desc, sup_q.numpy().astype(np.float64), dtype=np.float64).reshape(500, -1)
    w64 = w.numpy().astype(np.float64)[:, None]
    q = starts.double().numpy().copy()
    m, v = np.zeros_like(q), np.zeros_like(q)
    alive, evals, near = np.ones(len(q), bool), np.zeros(len(q), np.int64), np.zeros(len(q), bool)
    for t in range(1, n + 1):
        s, gr, _ = oracle.score_grad(desc, 1, 1.0, 1.0, sup64, w64, q, dtype=np.float64)
        ex = s[:, 0] - margin
        near |= alive & (np.abs(ex) < 1e-5)
        evals += alive
        alive &= ex > 0
        m = np.where(alive[:, None], 0.9 * m + 0.1 * gr, m)
        v = np.where(alive[:, None], 0.999 * v + 0.001 * gr * gr, v)
        step = lr / (1 - 0.9 ** t) * m / (np.sqrt(v) / np.sqrt(1 - 0.999 ** t) + 1e-8)
        q = np.where(alive[:, None], q - step, q)
    ok = ~near
    assert ok.sum() > 280
    np.testing.assert_array_equal(_np(checks)[ok], evals[ok])
    assert relerr(_np(final)[ok], q[ok]) < TOL
    assert (evals > 1).sum() > 50 and (evals < n).sum() > 50


@pytest.mark.parametrize("k", [1, 3, 7])
def test_stopped_loops_taken_out_of_the_sweep(k):
    """compact_every = k: the loops that stopped leave the sweep's batch after every k-th step (dcx_escape_adam reads the count
    back and sizes the next launches by it).  Same evaluation counts, configurations and records as the un-compacted call up to
    the rounding of sums taken in another launch geometry; the golden batch as well; and the early return when nothing is left"""
    from diffco_amd import utils
    from diffco_amd.escape import OptimSampler
    d = load("escape")
    rob, dc = _baxter(d)
    g = torch.Generator().manual_seed(11 + k)
    lim = rob.limits
    starts = torch.rand((5000, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    s0 = dc.poly_score(starts)
    sampler = OptimSampler(rob, dc.poly_score, {"N_WAYPOINTS": 14, "safety_margin": float(s0.median()) - 0.03, "lr": 4e-2,
                                                "record_freq": 3, "post_transform": utils.wrap2pi})
    f0, c0, h0, n0 = sampler.optim_escape_batch(starts, history=True, compact_every=0)
    f1, c1, h1, n1 = sampler.optim_escape_batch(starts, history=True, compact_every=k)
    assert 0.2 < float((c0 == 1).float().mean()) < 0.8 and int(c0.max()) == 14      # a mix: free at once ... never free
    same = c0 == c1            # a loop whose excess passes within rounding of zero may stop a step apart
    assert float(same.float().mean()) > 0.995
    assert torch.equal(n0[same], n1[same])
    assert relerr(_np(f1[same]), _np(f0[same])) < 1e-5 and relerr(_np(h1[:, same]), _np(h0[:, same])) < 1e-5
    # the golden batch (12 loops, records every 4 steps)
    sampler = OptimSampler(rob, dc.poly_score, {"N_WAYPOINTS": 15, "safety_margin": float(d["bx_marginb"]) - 0.05, "lr": 5e-2,
                                                "record_freq": 4})
    final, checks, hist, n_rec = sampler.optim_escape_batch(torch.from_numpy(d["bx_starts"]), history=True, compact_every=k)
    np.testing.assert_array_equal(_np(checks), d["bx_batch_checks"])
    np.testing.assert_array_equal(_np(n_rec), d["bx_batch_nrec"])
    assert relerr(_np(final), d["bx_batch_final"]) < TOL and relerr(_np(hist[:4]), d["bx_batch_hist"]) < TOL
    # everything free at the first evaluation: one sweep, then the call returns
    free = OptimSampler(rob, dc.poly_score, {"N_WAYPOINTS": 50, "safety_margin": 1e3})
    f, c = free.optim_escape_batch(starts, compact_every=1)
    assert torch.equal(f, starts) and int(c.min()) == int(c.max()) == 1


def test_compaction_is_refused_under_stream_capture():
    """compact_every > 0 reads a count back: on a capturing stream the call returns DCX_ERR_UNSUPPORTED before touching anything
    (the capture stays valid); compact_every = 0 is captured (test_the_loop_is_captured_in_a_hip_graph)"""
    import ctypes as C

    from diffco_amd import _lib, _ops, traj
    d = load("escape")
    rob, dc = _baxter(d)
    model = traj._resolve_model(dc.poly_score)
    lib = _lib.require_gpu()
    q = torch.from_numpy(d["bx_starts"]).cuda()
    steps = torch.zeros((len(q), 2), device="cuda", dtype=torch.int32)
    work = torch.empty(int(lib.dcx_escape_work_bytes(model._h, len(q))), device="cuda", dtype=torch.uint8)
    opts = _lib.EscapeOpts(5e-2, 0.9, 0.999, 1e-8, 6, 0, 0, 2, 0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s0 = model.score_raw(q)              # (this stream's scratch exists before the capture)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
        rc = lib.dcx_escape_adam(model._h, _ops._ptr(q), len(q), None, C.byref(opts), _ops._ptr(work), work.numel(), None,
                                 _ops._ptr(steps), C.c_void_p(side.cuda_stream))
        msg = lib.dcx_last_error()
        s1 = model.score_raw(q)
    graph.replay()
    torch.cuda.synchronize()
    assert rc == 2 and b"captured" in msg and torch.equal(s1, s0)


def test_one_loop_over_a_large_batch():
    """more than 1024 configurations in ONE loop: the decision is its own launch there (traj_kernels.hip); against the host loop
    on the same score"""
    from diffco_amd.escape import OptimSampler
    d = load("escape")
    rob, dc = _baxter(d)
    g = torch.Generator().manual_seed(7)
    lim = rob.limits
    starts = torch.rand((1500, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    s0 = dc.poly_score(starts)
    for margin, n in ((float(s0.mean()) - 0.02, 12), (float(s0.mean()) - 1e3, 5)):
        args = {"N_WAYPOINTS": n, "safety_margin": margin, "lr": 3e-2, "record_freq": 2}
        fused, host = OptimSampler(rob, dc.poly_score, args), OptimSampler(rob, dc.poly_score, dict(args, post_transform=lambda x: x))
        hf, cf = fused.optim_escape(starts)
        hh, ch = host.optim_escape(starts)
        assert (fused.last_route, host.last_route) == ("fused", "host")
        assert cf == ch and hf.shape == hh.shape and relerr(_np(hf), _np(hh)) < TOL
    assert cf == 5 and hf.shape == (4, 1500, 7)


def test_argument_checks_and_routes():
    from diffco_amd import utils
    from diffco_amd.escape import OptimSampler, resampling_escape
    d = load("escape")
    rob, dc = _baxter(d)
    starts = torch.from_numpy(d["bx_starts"])
    cfg = resampling_escape(rob)
    assert cfg.shape == (1, 7) and bool(((cfg >= rob.limits[:, 0]) & (cfg <= rob.limits[:, 1])).all())
    # a foreign optimiser or transform: the host loop; the batch entry point refuses
    sgd = OptimSampler(rob, dc.poly_score, {"optimizer": torch.optim.SGD, "opt_args": {"lr": 0.1}, "N_WAYPOINTS": 4,
                                            "safety_margin": -1e3})
    h, c = sgd.optim_escape(starts[:2])
    assert sgd.last_route == "host" and c == 4 and h.shape == (5, 2, 7)
    with pytest.raises(TypeError):
        sgd.optim_escape_batch(starts)
    # a start that is already free: one evaluation, one record, nothing moved
    free = OptimSampler(rob, dc.poly_score, {"safety_margin": 1e3, "post_transform": utils.wrap2pi})
    h, c = free.optim_escape(starts[:1])
    assert free.last_route == "fused" and c == 1 and h.shape == (1, 1, 7) and torch.equal(h[0], starts[:1])
    f, c = free.optim_escape_batch(starts)
    assert torch.equal(f, starts) and _np(c).tolist() == [1] * len(starts)
    # empty batch
    f, c = free.optim_escape_batch(starts[:0])
    assert f.shape == (0, 7) and c.numel() == 0
    # margin with the wrong number of entries
    with pytest.raises(ValueError):
        OptimSampler(rob, dc.poly_score, {"safety_margin": torch.zeros(3)}).optim_escape(starts[:1])


def test_the_loop_is_captured_in_a_hip_graph():
    """no allocation and no synchronisation inside dcx_escape_adam: the 2 x n_steps launches of a loop replay as one graph"""
    import ctypes as C

    from diffco_amd import _lib, _ops, traj
    d = load("escape")
    rob, dc = _baxter(d)
    model = traj._resolve_model(dc.poly_score)
    lib = _lib.require_gpu()
    starts = torch.from_numpy(d["bx_starts"]).cuda()
    B, dof = starts.shape
    margin = torch.tensor([float(d["bx_marginb"]) - 0.05], device="cuda")
    opts = _lib.EscapeOpts(5e-2, 0.9, 0.999, 1e-8, 15, 0, 0, 0, 0)
    q, steps = starts.clone(), torch.zeros((B, 2), device="cuda", dtype=torch.int32)
    work = torch.empty(int(lib.dcx_escape_work_bytes(model._h, B)), device="cuda", dtype=torch.uint8)

    def enqueue(st):
        _lib.check(lib.dcx_escape_adam(model._h, _ops._ptr(q), B, _ops._ptr(margin), C.byref(opts), _ops._ptr(work), work.numel(),
                                       None, _ops._ptr(steps), C.c_void_p(st.cuda_stream)))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        enqueue(side)
        side.synchronize()
        want_q, want_steps = q.clone(), steps.clone()
        np.testing.assert_array_equal(_np(want_steps[:, 0]), d["bx_batch_checks"])
        graph = torch.cuda.CUDAGraph()
        q.copy_(starts)
        with torch.cuda.graph(graph, stream=side):
            enqueue(side)
        q.copy_(starts)
        steps.zero_()
        graph.replay()
        side.synchronize()
    assert torch.equal(q, want_q) and torch.equal(steps, want_steps)


def test_a_foreign_transform_keeps_the_host_route():
    """a checker whose transform is not a diffco_amd fkine (here a width-preserving scaling) cannot run as
    dcx_escape_adam - its model scores FEATURES, and the library would differentiate the raw configuration.  The sampler must take
    the host loop, whose score runs the transform in torch first; and the history buffer is float32 whatever torch's default."""
    from diffco_amd import kernel
    from diffco_amd.escape import OptimSampler
    from diffco_amd.kernel_perceptrons import DiffCo
    g = torch.Generator().manual_seed(3)
    sup = torch.randn((80, 4), generator=g)
    dc = DiffCo(transform=lambda x: 2 * x)
    dc.support_points = sup
    dc.support_transformed = 2 * sup
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), 0.1 * torch.randn(80, generator=g) + 0.05
    start = torch.randn((1, 4), generator=g)
    args = {"N_WAYPOINTS": 6, "lr": 5e-2, "record_freq": 1, "safety_margin": float(dc.poly_score(start)) - 0.05}
    sampler = OptimSampler(None, dc.poly_score, args)
    hist, checks = sampler.optim_escape(start)
    assert sampler.last_route == "host"
    # a float64 torch restatement of the loop with the transform inside the score
    w, s2 = dc.rbf_nodes.double(), (2 * sup).double()
    p = start.double().clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=5e-2)
    kept, n = [], 0
    for _ in range(6):
        ex = (torch.cdist(2 * p, s2) @ w - args["safety_margin"]).sum()
        n += 1
        if ex <= 0:
            break
        kept.append(p.detach().clone())
        opt.zero_grad(); ex.backward(); opt.step()
    kept.append(p.detach().clone())
    assert checks == n and relerr(_np(hist), torch.stack(kept).numpy()) < TOL
    with pytest.raises(TypeError):
        sampler.optim_escape_batch(start)
    # the same checker WITHOUT a transform is fusable; float64 as torch's default dtype must not change the history's type
    d = load("escape")
    rob, bx = _baxter(d)
    torch.set_default_dtype(torch.float64)
    try:
        smp = OptimSampler(rob, bx.poly_score, {"N_WAYPOINTS": 20, "lr": 5e-2, "record_freq": 1, "safety_margin": float(d["bx_margin1"])})
        h, c = smp.optim_escape(torch.from_numpy(d["bx_starts"])[:1])
    finally:
        torch.set_default_dtype(torch.float32)
    assert smp.last_route == "fused" and c == int(d["bx_single_checks"]) and relerr(_np(h), d["bx_single_hist"]) < TOL
