"""Fused, batched-restart Adam trajectory optimiser (BASELINE config #5).

Same inputs, option keys and result record as `optim.adam_traj_optimize` (reference diffco/optim.py:13-163), but
all NUM_RE_TRIALS restarts advance together on the GPU inside native code (`dcx_traj_adam_run_mc`): one persistent
launch per <= 192 iterations when a path is one tile of the sweep, else per iteration the fused score + hinge-gradient
sweep over every waypoint of every restart and the fused Adam step (FK-coupled path terms, J^T, Adam update,
best-so-far bookkeeping).  No host synchronisation inside the loop (the reference syncs every iteration,
optim.py:107-118).  `dist_est` may have several outputs (a MultiDiffCo's rbf_score) with options['safety_margin'] one
number or one per class - the reference's `clamp(dist_est(p) - safety_margin, min=0).sum()` (optim.py:88-89 as
scripts/2d_trajopt.py:94-102 and scripts/active.py:28-121 call it).

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
    """ScoreModel behind `dist_est`: a ScoreModel, or a bound score method of a diffco_amd checker.  A checker whose transform is
    not a diffco_amd robot's `fkine` cannot be fused (its model scores FEATURES; only `FusedScorer.score` runs the foreign
    transform in front of it): TypeError, and the callers keep their host route (the escape loop used to score the
    raw configurations of such a checker)."""
    if isinstance(dist_est, _ops.ScoreModel):
        return dist_est
    owner, name = getattr(dist_est, "__self__", None), getattr(dist_est, "__name__", "")
    if owner is None:
        raise TypeError("fused_adam_traj_optimize needs a ScoreModel or a bound poly_score / rbf_score / score method "
                        "of a diffco_amd checker (an arbitrary callable cannot be fused)")
    from .deprecated import DiffCo as OldDiffCo
    from .kernel_perceptrons import DiffCo as NewDiffCo
    tf, model = None, None
    if isinstance(owner, NewDiffCo):
        tf = owner.transform
        if name == "poly_score":
            model = owner._poly_fused.model(tf, owner.rbf_kernel, owner.support_transformed, owner.rbf_nodes, device)
        elif name in ("score", "score_original"):
            model = owner._score_fused.model(tf, owner.kernel_func, owner.support_transformed, owner.gains, device)
    elif isinstance(owner, OldDiffCo):
        if name in ("rbf_score", "poly_score"):
            tf = owner.fkine
            feats = owner.support_fkine if tf is not None else owner.support_points
            model = owner._rbf_fused.model(tf, owner.rbf_kernel, feats, owner.rbf_nodes, device)
        elif name in ("score", "score_original"):
            tf, pk, feats = owner._score_state()
            model = owner._score_fused.model(tf, pk, feats, owner.gains, device)
    if model is None:
        raise TypeError(f"cannot fuse {dist_est!r}")
    if tf is not None and model.desc.kind == 0:
        raise TypeError(f"cannot fuse {dist_est!r}: its transform is not the fkine of a diffco_amd robot")
    return model


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


class ShardedAdamRun:
    """R_total restarts of one trajectory problem advanced together by the fused Adam step, sharded over the ranks of
    a torch.distributed group (one process per GPU; `group=None` and no initialised process group = one rank).

    Rank r owns the contiguous restarts `shard_bounds(R_total, r, world)` and runs them with `dcx_traj_adam_run`
    (one persistent launch per <= 192 iterations when a path fits one tile); nothing is exchanged while iterating —
    restarts are independent — and `finish()` gathers only the per-restart summaries (4 floats each) plus the candidate
    paths.  `fused_adam_traj_optimize` and bench.py's config #5 are both built on this class."""

    def __init__(self, model, limits, init_paths, lr, safety_margin, max_speed, valid_tol=None, grad_tol=None,
                 group=None, sharded=None):
        import torch.distributed as dist
        from .sharded import shard_bounds
        self.lib = _lib.require_gpu()
        self.model, self.group = model.acquire(), group   # a lease: the checker's cache must not refill these rows under the run
        self.n_total = len(init_paths)
        self.sharded = (group is not None or bool(sharded)) and dist.is_initialized()
        self.rank, self.world = (dist.get_rank(group), dist.get_world_size(group)) if self.sharded else (0, 1)
        lo, hi = shard_bounds(self.n_total, self.rank, self.world)
        self.lo, self.hi, self.R = lo, hi, hi - lo
        dev = model.dev
        f32 = dict(device=dev, dtype=torch.float32)
        W, dof = init_paths.shape[1], init_paths.shape[2]
        self.W, self.dof = W, dof
        path = init_paths[lo:hi].to(**f32).contiguous().clone() if self.R else torch.empty((0, W, dof), **f32)
        R, inf = self.R, float('inf')
        self.margin = model.margins(safety_margin)    # one per class (a number serves all of them)
        self.t = dict(path=path, adam_m=torch.zeros_like(path), adam_v=torch.zeros_like(path),
                      limits=limits.to(**f32).contiguous(), col_score=torch.empty((R * W * model.C,), **f32),
                      col_grad=torch.empty((R * W, dof), **f32), stats=torch.zeros((R, 8), **f32),
                      lowest_loss=torch.full((R,), inf, **f32), lowest_obj=torch.full((R,), inf, **f32),
                      lowest_path=path.clone(), best_valid_obj=torch.full((R,), inf, **f32),
                      best_valid_path=path.clone(), done=torch.zeros((R,), device=dev, dtype=torch.int32),
                      steps=torch.zeros((R,), device=dev, dtype=torch.int32))
        self.st = _lib.TrajState(R, W, *(C.c_void_p(v.data_ptr() if v.numel() else 0) for v in self.t.values()))
        self.opt = _lib.TrajOpts(lr, 0.9, 0.999, 1e-8, DIF_WEIGHT, COLLISION_WEIGHT, MAX_MOVE_WEIGHT, JOINT_LIMIT_WEIGHT,
                                 float(self.margin[0]), float(max_speed),
                                 VALID_CONSTRAINT_LOSS if valid_tol is None else float(valid_tol),
                                 STATIONARY_GRAD_NORM if grad_tol is None else float(grad_tol))
        self.it = 0

    def close(self):
        """give the model's lease back (idempotent; also done when the run is garbage-collected)"""
        m, self._released = self.model, getattr(self, "_released", False) or not hasattr(self, "model")
        if not self._released:
            self._released = True
            m.release()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass

    def run(self, n_iters):
        """enqueue n_iters more iterations on torch's current stream (no host synchronisation)"""
        if self.R and n_iters > 0:
            dev = self.model.dev
            with torch.cuda.device(dev):
                stream = self.model._st()
                _lib.check(self.lib.dcx_traj_adam_run_mc(self.model._h, C.byref(self.st), C.byref(self.opt), self.margin,
                                                         self.it + 1, int(n_iters), stream))
        self.it += n_iters

    def all_done(self):
        return self.R == 0 or bool(self.t['done'].all())

    def finish(self):
        """(summaries [R_total, 4] on the host: best_valid_obj, lowest_loss, lowest_obj, steps; best_valid_path,
        lowest_path [R_total, W, dof] on the device) — identical on every rank"""
        from .sharded import all_gather_rows
        t = self.t
        if self.R and bool((t['stats'][:, 7] < 0).any()):
            # the cluster form of the persistent kernel (csrc/traj_fused.h) waited ~1 s for a peer workgroup's rows
            raise _lib.DcxError("dcx_traj_adam_run: a workgroup exchange gave up (stats[:, 7] == -1); the paths were left "
                                "as the last launch found them.  DCX_TRAJ_YS=1 runs one workgroup per path")
        summ = torch.stack([t['best_valid_obj'], t['lowest_loss'], t['lowest_obj'], t['steps'].float()], dim=1)
        bvp, lop = t['best_valid_path'], t['lowest_path']
        if self.sharded and self.world > 1:
            summ = all_gather_rows(summ, self.n_total, self.group)
            bvp = all_gather_rows(bvp, self.n_total, self.group)
            lop = all_gather_rows(lop, self.n_total, self.group)
        return summ.cpu(), bvp, lop


def fused_adam_traj_optimize(robot, dist_est, start_cfg, target_cfg, options, group=None):
    """Drop-in for `optim.adam_traj_optimize` with all restarts batched on the GPU.  With an initialised
    torch.distributed `group` (or the default group when options['distributed'] is true) the restarts are sharded
    across ranks (`ShardedAdamRun`); every rank returns the same record."""
    n_trials, max_iter = options['NUM_RE_TRIALS'], options['MAXITER']
    lr = options.get('extra_optimizer_options', {}).get('lr', 5e-1)
    seed = options['seed']
    torch.manual_seed(seed)
    prob = _PathProblem(robot, start_cfg, target_cfg, options)
    t0 = time.time()
    model = _resolve_model(dist_est)   # any class count: a MultiDiffCo score under options['safety_margin'] = a number or [C]
    desc = robot.fk_desc()
    if desc.key() != model.desc.key():
        raise ValueError("the checker's transform is not this robot's fkine: the fused step shares one FK")

    # ---- initial paths, exactly as the reference's sequential trials would draw them ----------------------
    inits = [prob.make_init(t).clone() for t in range(n_trials)]
    W, dof = inits[0].shape
    if W == 2:  # nothing to optimise (reference: optim.py:61-72)
        return {'start_cfg': _np(start_cfg).tolist(), 'target_cfg': _np(target_cfg).tolist(), 'cnt_check': 0,
                'cost': 0.0, 'time': time.time() - t0, 'success': True, 'seed': seed,
                'solution': inits[0].numpy().tolist()}
    if any(p.shape != (W, dof) for p in inits):
        raise ValueError("all restarts must have the same number of waypoints (init_solution vs N_WAYPOINTS)")

    run = ShardedAdamRun(model, robot.limits, torch.stack(inits), lr, prob.safety_margin, prob.max_speed,
                         grad_tol=options.get('stationary_grad_norm'), group=group,
                         sharded=options.get('distributed', False))
    chunk = int(options.get('fused_chunk', 50))  # iterations enqueued between two "all done?" checks
    while run.it < max_iter:
        run.run(min(chunk, max_iter - run.it))
        if run.it < max_iter and run.all_done():  # this rank's restarts are all frozen (no collective inside the loop)
            break
    # ---- gather the per-restart summaries (and paths) and apply the reference's selection policy ----------
    summ, best_valid_path, lowest_path = run.finish()
    run.close()
    bvo, lol, loo, nst = summ[:, 0], summ[:, 1], summ[:, 2], summ[:, 3]
    t_win, found, cost, cnt = select_trial(bvo, lol, loo, nst, W)
    solution = (best_valid_path if found else lowest_path)[t_win]
    return {'start_cfg': _np(start_cfg).tolist(), 'target_cfg': _np(target_cfg).tolist(), 'cnt_check': cnt,
            'cost': cost, 'time': time.time() - t0, 'success': found, 'seed': seed,
            'solution': solution.double().cpu().numpy().tolist(),
            'trial': t_win, 'cnt_check_batched': int(nst.sum().item()) * W, 'iterations_enqueued': run.it}
