"""N > 1 path on CPU: two processes over gloo exercise the batch sharding and the all-gather of
per-rank results (RCCL on the GPU box, gloo here).  The per-rank compute is a stand-in function:
what is under test is the sharding/gather logic the bench and ShardedScorer rely on."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffco_amd.sharded import ShardedScorer, all_gather_rows, shard_bounds
        q = torch.arange(n_total * 3, dtype=torch.float32).reshape(n_total, 3)  # identical on every rank
        up = torch.arange(n_total, dtype=torch.float32).reshape(n_total, 1)

        def fake_score_and_grad(q_local, up_local):  # stand-in for ScoreModel.score_and_grad
            return (q_local.sum(1, keepdim=True) * up_local, q_local * 2.0)
        scorer = ShardedScorer(fake_score_and_grad)
        s, g = scorer(q, up)
        assert s.shape == (n_total, 1) and g.shape == (n_total, 3)
        assert torch.equal(s, q.sum(1, keepdim=True) * up) and torch.equal(g, q * 2.0)
        # no-gather mode keeps the local slice
        lo, hi = shard_bounds(n_total, rank, world)
        s_loc, _ = ShardedScorer(fake_score_and_grad, gather=False)(q, up)
        assert s_loc.shape == (hi - lo, 1)
        # ragged explicit gather
        rows = all_gather_rows(q[lo:hi], n_total)
        assert torch.equal(rows, q)
        torch.save(s, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _run(n_total, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in (0, 1))
    assert torch.equal(a, b)  # every rank holds the same gathered result


def test_even_split(tmp_path):
    _run(64, tmp_path)


def test_ragged_split(tmp_path):
    _run(65, tmp_path)
