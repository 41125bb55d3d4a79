#!/usr/bin/env python3
"""tools/model_latency.py — developer tool (GPU box): what (re)building the fused model costs: wall-clock
per call of dcx_model_create through the host (CPU tensors in: the round-3 path for every input), dcx_model_create_ex from
device tensors (rows packed by one kernel, 16 bytes back) and dcx_model_update (the same into existing storage), for the
model sizes of the BASELINE configs; then the round the reference's active-learning loop repeats
(collision_checkers.py:220-252): new weights into a checker -> first poly_score, with the in-place refill and without."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffco_amd import _ops, kernel, model  # noqa: E402
from diffco_amd.kernel_perceptrons import DiffCo  # noqa: E402

dev = torch.device("cuda", 0)
rob = model.BaxterLeftArmFK()
lim = rob.limits
desc = rob.fk_desc()
g = torch.Generator().manual_seed(0)


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for S, C, kspec in ((200, 1, (0, 10.0, 2.0)), (1000, 1, (1, 1.0, 1.0)), (2000, 1, (1, 1.0, 1.0)), (2000, 5, (0, 10.0, 2.0)), (10000, 1, (1, 1.0, 1.0))):
    sq = torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    sup = _ops.fkine(desc, sq.to(dev)).reshape(S, -1)
    W = torch.randn((S, C), generator=g).to(dev)
    sup_h, W_h = sup.cpu(), W.cpu()
    keep = _ops.ScoreModel(desc, *kspec, sup, W, device=dev, capacity=S)
    t_host = timeit(lambda: _ops.ScoreModel(desc, *kspec, sup_h, W_h, device=dev))
    t_dev = timeit(lambda: _ops.ScoreModel(desc, *kspec, sup, W, device=dev))
    t_upd = timeit(lambda: keep.update(sup, W))
    print(f"S={S:<6} C={C} kernel={kspec[0]}  create via host {t_host:8.1f} us   create on device {t_dev:8.1f} us   update in place {t_upd:8.1f} us", flush=True)

# one round of an active-learning loop on a checker: new spline nodes (fit_poly's output) -> the first score afterwards
S = 2000
sq = torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
dc = DiffCo(kernel_func=kernel.RQKernel(10.0), transform=rob.fkine)
dc.support_points = sq.to(dev)
dc.support_transformed = rob.fkine(sq.to(dev))
dc.rbf_kernel = kernel.Polyharmonic(1, 1.0)
q = (torch.rand((50, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev)
for reuse in (True, False):
    def round_():
        dc.rbf_nodes = torch.randn(S, device=dev)
        if not reuse:
            dc._poly_fused._retired = None
            dc._poly_fused._model = None
        with torch.no_grad():
            return dc.poly_score(q)
    print(f"new rbf_nodes -> poly_score(50 waypoints), {'model refilled in place' if reuse else 'model rebuilt        '}: {timeit(round_):8.1f} us", flush=True)

# the reference's recommended facade: ForwardKinematicsDiffCo.update (collision_checkers.py:220-252) on a URDF Panda with a
# synthetic ground truth (a sphere the hand must avoid) - the whole round: sampling, ground truth, device trainer, fit_poly,
# verification scores - with the model refilled in place and rebuilt
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import urdf_robot  # noqa: E402
from diffco_amd.collision_checkers import ForwardKinematicsDiffCo  # noqa: E402

urob = urdf_robot("urdf_panda")
k_tip = urob.unique_position_link_names.index("panda_virtual_ee_link")
centre = torch.tensor([0.35, 0.0, 0.55], device=dev)


def ground_truth(qq):
    return ((urob.fkine(qq.to(dev))[:, :, k_tip] - centre).norm(dim=1) < 0.35).float().cpu()


for reuse in (True, False, True, False):
    torch.manual_seed(0)
    fk = ForwardKinematicsDiffCo(robot=urob, gamma=10, gt_check_func=ground_truth)
    fk.fit(num_samples=1500, verify_ratio=0.2, fix_joints=[7], fix_joint_values=[0.04])
    fk.update(num_samples=200, verify=0.2)
    torch.cuda.synchronize()
    t0, n = time.perf_counter(), 12
    for _ in range(n):
        if not reuse:
            fk.perceptron._poly_fused._retired = None
            fk.perceptron._poly_fused._model = None
        fk.update(num_samples=200, verify=0.2)
    torch.cuda.synchronize()
    print(f"ForwardKinematicsDiffCo.update (200 new samples, {len(fk.perceptron.gains)} supports), "
          f"{'model refilled in place' if reuse else 'model rebuilt        '}: {(time.perf_counter() - t0) / n * 1e3:8.2f} ms per round", flush=True)
