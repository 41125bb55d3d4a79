"""Fused, batched-restart Adam trajectory optimiser (BASELINE config #5).

Same inputs, option keys and result record as `optim.adam_traj_optimize` (reference diffco/optim.py:13-163), but
all NUM_RE_TRIALS restarts advance together on the GPU and each iteration is two launches enqueued from native
code (`dcx_traj_adam_run`): the fused score + hinge-gradient sweep over every waypoint of every restart, and
the fused Adam step (FK-coupled path terms, J^T, Adam update, best-so-far bookkeeping).  No host
synchronisation inside the loop (the reference syncs every iteration, optim.py:107-118).

The restarts are initialised exactly like the reference's sequential trials (trial 0: init_solution or the
straight line, later trials: torch.rand in joint limits, same seed, same order), and the returned record follows
the reference's policy — the first trial, in order, that produced a valid path wins; otherwise the lowest loss
seen anywhere — so a reference run and this one answer the same question.  Arithmetic is fp32 on the device (the
reference optimises in fp64 on the host).
"""
import ctypes as C
import time

import torch

from . import _lib, _ops
from .optim import (COLLISION_WEIGHT, DIF_WEIGHT, JOINT_LIMIT_WEIGHT, MAX_MOVE_WEIGHT, STATIONARY_GRAD_NORM,
                    VALID_CONSTRAINT_LOSS, _np, _PathProblem)


def _resolve_model(dist_est, device=None):
    """ScoreModel behind `dist_est`: a ScoreModel, or a bound score method of a diffco_amd checker"""
    if isinstance(dist_est, _ops.ScoreModel):
        return dist_est
    owner, name = getattr(dist_est, "__self__", None), getattr(dist_est, "__name__", "")
    if owner is None:
        raise TypeError("fused_adam_traj_optimize needs a ScoreModel or a bound poly_score / rbf_score / score method "
                        "of a diffco_amd checker (an arbitrary callable cannot be fused)")
    from .deprecated import DiffCo as OldDiffCo
    from .kernel_perceptrons import DiffCo as NewDiffCo
    if isinstance(owner, NewDiffCo):
        if name == "poly_score":
            return owner._poly_fused.model(owner.transform, owner.rbf_kernel, owner.support_transformed, owner.rbf_nodes, device)
        if name in ("score", "score_original"):
            return owner._score_fused.model(owner.transform, owner.kernel_func, owner.support_transformed, owner.gains, device)
    if isinstance(owner, OldDiffCo):
        if name in ("rbf_score", "poly_score"):
            feats = owner.support_fkine if owner.fkine is not None else owner.support_points
            return owner._rbf_fused.model(owner.fkine, owner.rbf_kernel, feats, owner.rbf_nodes, device)
        if name in ("score", "score_original"):
            tf, pk, feats = owner._score_state()
            return owner._score_fused.model(tf, pk, feats, owner.gains, device)
    raise TypeError(f"cannot fuse {dist_est!r}")


def select_trial(best_valid_obj, lowest_loss, lowest_obj, steps, n_waypoints):
    """The reference's selection policy over per-trial summaries (1-D tensors over trials, in trial order):
    the first trial that found a valid path wins (optim.py:129-131, sequential trials stop there), otherwise the
    lowest loss seen in any trial (optim.py:134-141).  Returns (trial, found, cost, cnt_check) with cnt_check =
    what the sequential reference would have evaluated."""
    valid = torch.isfinite(best_valid_obj)
    if bool(valid.any()):
        t = int(torch.nonzero(valid)[0])
        return t, True, float(best_valid_obj[t]), int(steps[:t + 1].sum().item()) * n_waypoints
    t = int(torch.argmin(lowest_loss))
    return t, False, float(lowest_obj[t]), int(steps.sum().item()) * n_waypoints


def fused_adam_traj_optimize(robot, dist_est, start_cfg, target_cfg, options, group=None):
    """Drop-in for `optim.adam_traj_optimize` with all restarts batched on the GPU.  With an initialised
    torch.distributed `group` (or the default group when options['distributed'] is true) the restarts are sharded
    across ranks; every rank returns the same record."""
    import torch.distributed as dist
    lib = _lib.require_gpu()
    n_trials, max_iter = options['NUM_RE_TRIALS'], options['MAXITER']
    lr = options.get('extra_optimizer_options', {}).get('lr', 5e-1)
    seed = options['seed']
    torch.manual_seed(seed)
    prob = _PathProblem(robot, start_cfg, target_cfg, options)
    t0 = time.time()
    model = _resolve_model(dist_est)
    if model.C != 1:
        raise ValueError("the fused optimiser needs a single-output collision score (C == 1)")
    desc = robot.fk_desc()
    if desc.key() != model.desc.key():
        raise ValueError("the checker's transform is not this robot's fkine: the fused step shares one FK")
    dev = model.dev

    # ---- initial paths, exactly as the reference's sequential trials would draw them ----------------------
    inits = [prob.make_init(t).clone() for t in range(n_trials)]
    W, dof = inits[0].shape
    if W == 2:  # nothing to optimise (reference: optim.py:61-72)
        cp = robot.fkine(inits[0])
        return {'start_cfg': _np(start_cfg).tolist(), 'target_cfg': _np(target_cfg).tolist(), 'cnt_check': 0,
                'cost': 0.0, 'time': time.time() - t0, 'success': True, 'seed': seed,
                'solution': inits[0].numpy().tolist()}
    if any(p.shape != (W, dof) for p in inits):
        raise ValueError("all restarts must have the same number of waypoints (init_solution vs N_WAYPOINTS)")

    sharded = (group is not None or options.get('distributed', False)) and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if sharded else (0, 1)
    from .sharded import all_gather_rows, shard_bounds
    lo, hi = shard_bounds(n_trials, rank, world)
    R = hi - lo

    f32 = dict(device=dev, dtype=torch.float32)
    path = torch.stack(inits[lo:hi]).to(**f32).contiguous() if R else torch.empty((0, W, dof), **f32)
    adam_m, adam_v = torch.zeros_like(path), torch.zeros_like(path)
    limits = robot.limits.to(**f32).contiguous()
    col_score = torch.empty((R * W,), **f32)
    col_grad = torch.empty((R * W, dof), **f32)
    stats = torch.zeros((R, 8), **f32)
    inf = float('inf')
    lowest_loss, lowest_obj = torch.full((R,), inf, **f32), torch.full((R,), inf, **f32)
    best_valid_obj = torch.full((R,), inf, **f32)
    lowest_path, best_valid_path = path.clone(), path.clone()
    done = torch.zeros((R,), device=dev, dtype=torch.int32)
    steps = torch.zeros((R,), device=dev, dtype=torch.int32)

    st = _lib.TrajState(R, W, *(C.c_void_p(t.data_ptr() if t.numel() else 0) for t in (
        path, adam_m, adam_v, limits, col_score, col_grad, stats, lowest_loss, lowest_obj, lowest_path, best_valid_obj,
        best_valid_path, done, steps)))
    opt = _lib.TrajOpts(lr, 0.9, 0.999, 1e-8, DIF_WEIGHT, COLLISION_WEIGHT, MAX_MOVE_WEIGHT, JOINT_LIMIT_WEIGHT,
                        float(prob.safety_margin), float(prob.max_speed), VALID_CONSTRAINT_LOSS, STATIONARY_GRAD_NORM)
    chunk = int(options.get('fused_chunk', 50))  # iterations enqueued between two "all done?" checks
    it = 0
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        while it < max_iter and R:
            n = min(chunk, max_iter - it)
            _lib.check(lib.dcx_traj_adam_run(model._h, C.byref(st), C.byref(opt), it + 1, n, stream))
            it += n
            if it < max_iter and bool(done.all()):
                break
    # ---- gather the per-restart summaries (and paths) and apply the reference's selection policy ----------
    summ = torch.stack([best_valid_obj, lowest_loss, lowest_obj, steps.float()], dim=1)
    if sharded:
        summ = all_gather_rows(summ, n_trials, group)
        best_valid_path = all_gather_rows(best_valid_path, n_trials, group)
        lowest_path = all_gather_rows(lowest_path, n_trials, group)
    summ = summ.cpu()
    bvo, lol, loo, nst = summ[:, 0], summ[:, 1], summ[:, 2], summ[:, 3]
    t_win, found, cost, cnt = select_trial(bvo, lol, loo, nst, W)
    solution = (best_valid_path if found else lowest_path)[t_win]
    return {'start_cfg': _np(start_cfg).tolist(), 'target_cfg': _np(target_cfg).tolist(), 'cnt_check': cnt,
            'cost': cost, 'time': time.time() - t0, 'success': found, 'seed': seed,
            'solution': solution.double().cpu().numpy().tolist(),
            'trial': t_win, 'cnt_check_batched': int(nst.sum().item()) * W, 'iterations_enqueued': it}
