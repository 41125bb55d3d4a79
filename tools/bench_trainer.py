#!/usr/bin/env python3
"""tools/bench_trainer.py — time the device perceptron trainer (dcx_train_perceptron: several workgroups with a grid
barrier per iteration, knob train_grid = 1 / rule; one persistent workgroup, knob 0) against the host loop (same
algorithm, kernel rows from the HIP kernel-matrix kernel) on synthetic Baxter data."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from diffco_amd import _lib, kernel, model  # noqa: E402
from diffco_amd.kernel_perceptrons import DiffCo  # noqa: E402

rob = model.BaxterLeftArmFK()
lim = rob.limits
lib = _lib.require_gpu()
for N in (3000, 10000, 30000, 100000):
    g = torch.Generator().manual_seed(N)
    X = torch.rand((N, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    P = rob.fkine(X)
    y = torch.where(((P - torch.tensor([0.7, 0.3, 0.3])).norm(dim=-1) < 0.3).any(dim=1), 1.0, -1.0)
    for mode in ("grid", "grid", "one-wg", "one-wg", "host"):
        if mode == "host":
            if N > 10000:
                continue
            os.environ["DCX_HOST_TRAINER"] = "1"
        else:
            os.environ.pop("DCX_HOST_TRAINER", None)
            lib.dcx_debug_set(b"train_grid", 1 if mode == "grid" else 0)
            if mode == "one-wg" and N > 30000:
                continue
        dc = DiffCo(kernel_func=kernel.RQKernel(10.0), beta=1.0, transform=rob.fkine)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dc.train(X, y, max_iteration=N)
        torch.cuda.synchronize()
        print(f"N={N:6d} {mode:7s} {1e3 * (time.perf_counter() - t0):9.1f} ms  supports={dc.valid_supports}", flush=True)
