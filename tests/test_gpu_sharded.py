"""The N > 1 path with the REAL kernels: two processes share the one leased GPU, rendezvous over gloo (RCCL needs one GPU
per rank) and run `ShardedScorer(ScoreModel.score_and_grad)` and the sharded fused Adam optimiser; with the launch
geometry pinned, the gathered results are bit-identical to a single-process run of the same inputs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    """model + batch + trajectory problem, built from seeds (identical in every process)"""
    from diffco_amd import _lib, _ops, model
    lib = _lib.require_gpu()
    for k, v in (("nw", 16), ("ys", 1)):  # pinned slicing: results do not depend on how many rows a rank holds
        _lib.check(lib.dcx_debug_set(k.encode(), v))
    rob = model.BaxterLeftArmFK()
    g = torch.Generator().manual_seed(123)
    lim = rob.limits
    S = 700
    sup_q = torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    desc = rob.fk_desc()
    sup = _ops.fkine(desc, sup_q.cuda()).reshape(S, -1)
    m1 = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, (0.05 * torch.randn(S, generator=g)).cuda())       # C = 1, Polyharmonic
    m5 = _ops.ScoreModel(desc, 0, 10.0, 2.0, sup, torch.randn((S, 5), generator=g).cuda())           # C = 5, RQ
    q = (torch.rand((1001, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    up = torch.randn((1001, 5), generator=g).cuda()
    start = torch.rand(7, generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    target = torch.rand(7, generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    options = {"N_WAYPOINTS": 18, "NUM_RE_TRIALS": 5, "MAXITER": 40, "safety_margin": -1e3, "max_speed": 0.3, "seed": 3,
               "history": False, "extra_optimizer_options": {"lr": 0.05}}
    return rob, m1, m5, q, up, start, target, options


def _compute(rob, m1, m5, q, up, start, target, options, group):
    from diffco_amd import fused_adam_traj_optimize
    from diffco_amd.sharded import ShardedScorer
    out = {}
    for n in (1000, 1001):  # even and ragged split
        s, g = ShardedScorer(m1.score_and_grad, group=group)(q[:n])
        s5, g5 = ShardedScorer(m5.score_and_grad, group=group)(q[:n], up[:n])
        out[f"s1_{n}"], out[f"g1_{n}"], out[f"s5_{n}"], out[f"g5_{n}"] = s.cpu(), g.cpu(), s5.cpu(), g5.cpu()
    rec = fused_adam_traj_optimize(rob, m1, start, target, dict(options), group=group)
    out["solution"] = torch.tensor(rec["solution"])
    out["scalars"] = torch.tensor([rec["cost"], float(rec["cnt_check"]), float(rec["success"]), float(rec["trial"])], dtype=torch.float64)
    # the same optimiser on the five-class model under per-class margins (round 6: dcx_traj_adam_run_mc, restarts sharded)
    margins = m5.score_raw(q[:256]).median(dim=0).values.cpu()
    rec5 = fused_adam_traj_optimize(rob, m5, start, target, dict(options, safety_margin=margins), group=group)
    out["solution5"] = torch.tensor(rec5["solution"])
    out["scalars5"] = torch.tensor([rec5["cost"], float(rec5["cnt_check"]), float(rec5["success"]), float(rec5["trial"])], dtype=torch.float64)
    return out


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = _compute(*_problem(), group=dist.group.WORLD)
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_processes_on_one_gpu_match_the_single_process_run(tmp_path, knob):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in (0, 1))
    knob("nw", 16)   # restored by the fixture (this process computes the unsharded reference)
    knob("ys", 1)
    ref = _compute(*_problem(), group=None)
    assert set(a) == set(b) == set(ref)
    for k in ref:
        assert torch.equal(a[k], b[k]), k          # every rank holds the same gathered result
        assert torch.equal(a[k], ref[k]), (k, float((a[k].double() - ref[k].double()).abs().max()))
    assert ref["s1_1001"].shape == (1001, 1) and ref["g5_1001"].shape == (1001, 7)
    assert float(ref["g5_1000"].abs().max()) > 0 and np.isfinite(float(ref["scalars"][0]))
