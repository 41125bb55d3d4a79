#!/usr/bin/env python3
"""tools/api_latency.py — developer tool (GPU box): wall-clock latency of the Python API around the fused kernel for
trajectory-sized batches (the reference's optimisers call poly_score with 20-50 waypoints): poly_score forward,
forward + backward, and the raw C-ABI call, per call."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffco_amd import kernel, model  # noqa: E402
from diffco_amd.kernel_perceptrons import DiffCo  # noqa: E402

rob = model.BaxterLeftArmFK()
lim = rob.limits
S = 2000
torch.manual_seed(0)
dev = torch.device("cuda")
dc = DiffCo(kernel_func=kernel.RQKernel(10.0), transform=rob.fkine)
sq = torch.rand(S, 7) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
dc.support_points = sq.to(dev)
dc.support_transformed = rob.fkine(sq.to(dev))
dc.gains = torch.randn(S, device=dev)
dc.rbf_kernel = kernel.Polyharmonic(1, 1.0)
dc.rbf_nodes = torch.randn(S, device=dev)


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


# what torch's autograd engine costs by itself on this box: one elementwise op + sum, backward through it (no diffco_amd code)
qe = torch.rand(50, 7, device=dev, requires_grad=True)


def engine_only():
    (g,) = torch.autograd.grad((qe * 2.0).sum(), qe)
    return g


print(f"torch alone, (q * 2).sum() forward + autograd.grad on a [50, 7] CUDA tensor: {timeit(engine_only):7.1f} us per call")

for where in ("cuda", "cpu"):
    for B in (20, 50, 256, 4096):
        q = (torch.rand(B, 7) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(where)
        qg = q.clone().requires_grad_(True)
        m = dc._poly_fused.model(dc.transform, dc.rbf_kernel, dc.support_transformed, dc.rbf_nodes, dev)
        q32 = q.to(dev).float().contiguous()

        def fwd():
            with torch.no_grad():
                return dc.poly_score(q)

        def fwd_bwd():
            s = dc.poly_score(qg)
            (g,) = torch.autograd.grad(s.sum(), qg)
            return g

        def raw():
            return m.score_grad_raw(q32)

        def sg():   # the non-autograd Python entry: score and gradient of one launch, on the caller's device and dtype
            return m.score_and_grad(q)

        print(f"q on {where:<4} B={B:<5} poly_score {timeit(fwd):7.1f} us   + backward {timeit(fwd_bwd):7.1f} us   "
              f"ScoreModel.score_and_grad {timeit(sg):7.1f} us   raw dcx_score_grad {timeit(raw):7.1f} us")

# a HIP-graph replay of the same call (VERDICT r4 item 4 asked for the offer): one captured dcx_score_grad on static buffers, replayed
# after copying the new q into the graph's input - against simply calling the library again
for B in (50, 4096):
    m = dc._poly_fused.model(dc.transform, dc.rbf_kernel, dc.support_transformed, dc.rbf_nodes, dev)
    q_new = (torch.rand(B, 7) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev)
    q_static = q_new.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):          # the split-launch scratch of this stream must exist before the capture
        m.score_grad_raw(q_static)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        out_static = m.score_grad_raw(q_static)
    torch.cuda.synchronize()

    def replay():
        q_static.copy_(q_new)
        graph.replay()
        return out_static

    def replay_only():
        graph.replay()
        return out_static

    print(f"B={B:<5} raw dcx_score_grad {timeit(lambda: m.score_grad_raw(q_new)):7.1f} us   graph: copy q in + replay {timeit(replay):7.1f} us   "
          f"replay alone {timeit(replay_only):7.1f} us   (one kernel per call either way: a one-node graph has nothing to amortise)")

# PCIe-inclusive rate at the headline shape: host q in, host score + grad out (pageable and pinned buffers)
B = 65536
m = dc._poly_fused.model(dc.transform, dc.rbf_kernel, dc.support_transformed, dc.rbf_nodes, dev)
for pinned in (False, True):
    q = torch.rand(B, 7) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    s_h, g_h = torch.empty(B, 1), torch.empty(B, 7)
    if pinned:
        q, s_h, g_h = q.pin_memory(), s_h.pin_memory(), g_h.pin_memory()

    def roundtrip():
        qd = q.to(dev, non_blocking=True)
        s, g = m.score_grad_raw(qd)
        s_h.copy_(s, non_blocking=True)
        g_h.copy_(g, non_blocking=True)
        torch.cuda.synchronize()

    us = timeit(roundtrip, 100)
    print(f"headline B={B} host->device->host ({'pinned' if pinned else 'pageable'}): {us:8.1f} us  = {B / us:7.1f} M evals/s "
          f"(4.06 MB over PCIe per call)")
