"""Trajectory optimisers that consume a differentiable collision estimate `dist_est(q[N, dof])`.

Same entry points, option keys and result records as the reference's diffco/optim.py
(`adam_traj_optimize` 13-163, `givengrad_traj_optimize` 166-321, `trustconstr_traj_optimize`
324-516, `gradient_free_traj_optimize` 519-629); the objective/constraint definitions below restate
that file's behaviour (weights, margins, validity test, re-trial policy).  These are CALLERS of the
hot path and stay Python; every `dist_est(p)` / `robot.fkine(p)` they issue lands in the fused HIP
kernels of libdcx.  `fused_adam_traj_optimize` (diffco_amd/traj.py) is the batched-restart variant.

Differences worth knowing:
  * the straight-line initial path is built with numpy from array views of the endpoints (the reference's
    `torch.from_numpy(np.linspace(tensor, tensor))` breaks under numpy 2 / torch 2.10 — SURVEY.md §8c);
  * for a fusable dist_est (a diffco_amd checker's score of this robot) the collision constraint's Jacobian comes from ONE
    hinge-gradient launch over the densified path and its Hessian (the reference double-backwards through dist_est) from
    the analytic per-point Hessians of `dcx_score_hess`, both chained exactly through the dense-path geometry in fp64
    tensor ops on the device (`_ScipyTerms.jac_collision / hess_collision`); a foreign callable keeps the reference's
    autograd route (BFGS model for the Hessian by default).  Pinned to the reference's own constraint values, Jacobian,
    Hessian and SLSQP / trust-constr records (tests/golden/optim_scipy_baxter.npz, tools/make_golden.py gen_optim_scipy).
"""
import time
from typing import Dict

import numpy as np
import torch
from scipy.optimize import BFGS, NonlinearConstraint, minimize

from . import _lib, utils

DIF_WEIGHT = 1           # path-length term; fixed by the reference ("should NOT be changed")
MAX_MOVE_WEIGHT = 10
COLLISION_WEIGHT = 10
JOINT_LIMIT_WEIGHT = 10
VALID_CONSTRAINT_LOSS = 1e-2
STATIONARY_GRAD_NORM = 1e-4


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def _coord_axis(robot):
    """axis of `robot.fkine(p)` that runs over a control point's coordinates: 2 for the [W, m, d] layout of the
    reference's model.* robots (optim.py:93 sums dim=2), 1 for a URDF robot's [W, 3, L] link-origin stack"""
    fk = getattr(robot, "fk_desc", None)
    desc = fk() if callable(fk) else None
    return 1 if desc is not None and desc.feature_shape[0] == desc.point_dim and desc.kind == 5 and desc.t_coord_major else 2


def _record(start_cfg, target_cfg, cnt_check, cost, elapsed, success, seed, solution, **extra):
    rec = {'start_cfg': _np(start_cfg).tolist(), 'target_cfg': _np(target_cfg).tolist(), 'cnt_check': int(cnt_check),
           'cost': cost.item() if torch.is_tensor(cost) else float(cost), 'time': elapsed, 'success': bool(success), 'seed': seed,
           'solution': _np(solution).tolist()}
    rec.update(extra)
    return rec


class _PathProblem:
    """Waypoint path with fixed endpoints: initialisation policy and the cost / constraint terms."""

    def __init__(self, robot, start_cfg, target_cfg, options):
        self.robot, self.options = robot, options
        self.start, self.target = start_cfg, target_cfg
        self.n_waypoints = options['N_WAYPOINTS']
        self.max_speed = options['max_speed']
        self.safety_margin = options.get('safety_margin', 0.0)
        self.cnt_check = 0
        self.init_path = None

    def trivial(self):
        """options['init_solution'] of just two states: nothing to optimise"""
        init = self.options.get('init_solution')
        return init is not None and len(init) == 2

    def make_init(self, trial):
        """trial 0: the given init_solution or the straight line; later trials: uniform random waypoints"""
        if trial == 0:
            if 'init_solution' in self.options:
                init = self.options['init_solution']
                assert isinstance(init, torch.Tensor) and len(init) >= 2
                path = init.clone().double()
            else:
                line = np.linspace(_np(self.start).astype(np.float64), _np(self.target).astype(np.float64),
                                   num=self.n_waypoints)
                path = torch.from_numpy(line).double()
        else:
            lim = self.robot.limits.double()
            path = torch.rand((self.n_waypoints, self.robot.dof)).double() * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
        path[0] = self.start
        path[-1] = self.target
        self.init_path = path
        return path

    def full(self, inner):
        """inner waypoints (flat numpy or tensor) -> full path tensor with the fixed endpoints, requires grad"""
        p = torch.as_tensor(np.asarray(inner), dtype=torch.float64).reshape(-1, self.robot.dof)
        return torch.cat([self.init_path[:1], p, self.init_path[-1:]], dim=0).requires_grad_(True)

    # -- terms -----------------------------------------------------------------------------
    def path_length(self, p):
        cp = self.robot.fkine(p)
        return (cp[1:] - cp[:-1]).square().sum(), cp

    def joint_limit_violation(self, p):
        lim = self.robot.limits.to(p.dtype)
        return (torch.clamp(lim[:, 0] - p, min=0) + torch.clamp(p - lim[:, 1], min=0)).sum()

    def segment_collision(self, p, dist_est, dense_cap=None):
        """>= 0 when collision-free: per segment, the sum over its densified points of min(0, margin - score)"""
        dense = utils.dense_path(p, self.max_speed, dense_cap) if dense_cap is not None else \
            utils.dense_path(p, self.max_speed)
        self.cnt_check += len(dense)
        c = torch.clamp(-(dist_est(dense[1:-1]) - self.safety_margin), max=0).reshape(-1)
        n_seg, n_pt = len(p) - 1, len(dense) - 2
        per = -(-n_pt // n_seg)
        if per * n_seg != n_pt:
            c = torch.cat([c, c.new_zeros(per * n_seg - n_pt)])
        return c.reshape(n_seg, -1).sum(dim=1)


def _trivial_record(prob, robot, seed, t0):
    init = prob.options['init_solution']
    cp = robot.fkine(init[1:-1])
    cost = (cp[1:] - cp[:-1]).square().sum() if len(cp) > 1 else torch.zeros(())
    return _record(prob.start, prob.target, 0, cost, time.time() - t0, True, seed, init)


# ================================================================================ Adam
def adam_traj_optimize(robot, dist_est, start_cfg, target_cfg, options):
    """Penalty-method Adam on the waypoints; restarts (sequentially) from random paths until one trial
    yields a path whose constraint loss is <= 1e-2.  Returns the reference's record dict."""
    n_trials, max_iter = options['NUM_RE_TRIALS'], options['MAXITER']
    keep_history = options['history']
    lr = options.get('extra_optimizer_options', {}).get('lr', 5e-1)
    print('Adam lr = {}'.format(lr))
    seed = options['seed']
    torch.manual_seed(seed)
    prob = _PathProblem(robot, start_cfg, target_cfg, options)
    t0 = time.time()
    if prob.trivial():
        return _trivial_record(prob, robot, seed, t0)

    lowest = dict(loss=np.inf, obj=np.inf, sol=None, step=None, trial=None)
    valid = dict(obj=np.inf, sol=None, step=None, trial=None)
    histories = []
    found = False
    for trial in range(n_trials):
        p = prob.make_init(trial).requires_grad_(True)
        opt = torch.optim.Adam([p], lr=lr)
        hist = []
        for step in range(max_iter):
            opt.zero_grad()
            collision = torch.clamp(dist_est(p) - prob.safety_margin, min=0).sum()
            prob.cnt_check += len(p)
            objective, cp = prob.path_length(p)
            max_move = torch.clamp((cp[1:] - cp[:-1]).square().sum(dim=_coord_axis(robot)) - prob.max_speed ** 2, min=0).sum()
            constraint = (COLLISION_WEIGHT * collision + MAX_MOVE_WEIGHT * max_move
                          + JOINT_LIMIT_WEIGHT * prob.joint_limit_violation(p))
            loss = DIF_WEIGHT * objective + constraint
            loss.backward()
            p.grad[[0, -1]] = 0.0  # endpoints stay put
            opt.step()
            if keep_history:
                hist.append(p.data.clone())
            lv, ov, cv = loss.item(), objective.item(), constraint.item()
            if lv < lowest['loss']:
                lowest.update(loss=lv, obj=ov, sol=p.data.clone(), step=step, trial=trial)
            if cv <= VALID_CONSTRAINT_LOSS:
                if ov < valid['obj']:
                    valid.update(obj=ov, sol=p.data.clone(), step=step, trial=trial)
                if float(torch.norm(p.grad)) < STATIONARY_GRAD_NORM:
                    break
        histories.append(hist)
        if valid['sol'] is not None:
            found = True
            break
    chosen = valid if found else lowest
    return _record(start_cfg, target_cfg, prob.cnt_check, chosen['obj'], time.time() - t0, found, seed, chosen['sol'])


# ================================================================================ scipy drivers
class _ScipyTerms:
    """cost / constraints as numpy callables with analytic first derivatives from autograd"""

    def __init__(self, prob, dist_est, dense_cap=None):
        self.prob, self.dist_est, self.dense_cap = prob, dist_est, dense_cap

    def cost(self, x):
        obj, _ = self.prob.path_length(self.prob.full(x))
        return obj.item()

    def grad_cost(self, x):
        p = self.prob.full(x)
        obj, _ = self.prob.path_length(p)
        (g,) = torch.autograd.grad(obj, p, allow_unused=True)
        return np.zeros(len(x)) if g is None else g[1:-1].numpy().reshape(-1)

    def collision(self, x):
        with torch.no_grad():
            return self.prob.segment_collision(self.prob.full(x).detach(), self.dist_est, self.dense_cap).numpy()

    def _fused_model(self):
        """the ScoreModel behind dist_est when it is a diffco_amd checker method of this robot (any class count), else None"""
        if not hasattr(self, "_model"):
            self._model = None
            try:
                from .traj import _resolve_model
                m = _resolve_model(self.dist_est)
                if m.desc.key() == self.prob.robot.fk_desc().key():
                    self._model = m.acquire()   # a lease for the lifetime of these terms (released in __del__)
            except Exception:
                self._model = None
        return self._model

    def __del__(self):
        m = getattr(self, "_model", None)
        if m is not None:
            self._model = None
            m.release()

    def jac_collision(self, x):
        """[n_segments, (W-2)*dof] Jacobian of `collision`.  With a fusable dist_est it is assembled analytically
        from ONE score+hinge-gradient launch over the densified path: d c_r / d dense_n =
        -1[score_n > margin] * dscore_n/dq, chained through dense_n = p_i + k * max_step * unit(p_{i+1} - p_i).
        Otherwise: autograd's vectorised Jacobian through dist_est, as the reference does (optim.py:209-218)."""
        m = self._fused_model()
        if m is not None and self.dense_cap is None:
            return self._jac_collision_fused(x, m)
        p = self.prob.full(x)
        # the forward pass inside counts len(dense) checks, like the reference's jac_con_collision_free, which calls
        # con_collision_free once (optim.py:209-218 -> :190-197)
        jac = torch.autograd.functional.jacobian(
            lambda z: self.prob.segment_collision(z, self.dist_est, self.dense_cap), p, create_graph=False,
            strict=False, vectorize=True, strategy='reverse-mode')
        return jac[:, 1:-1].numpy().reshape(jac.shape[0], -1)

    def _jac_collision_fused(self, x, model):
        prob = self.prob
        p = prob.full(x).detach()
        W, dof = p.shape
        ms = prob.max_speed
        dense, seg, step = utils.dense_path_indexed(p, ms)
        prob.cnt_check += len(dense)  # one evaluation of the constraint, as the reference counts it (optim.py:197)
        pts, seg, step = dense[1:-1], seg[1:-1], step[1:-1]
        n_seg, n_pt = W - 1, len(pts)
        per = -(-n_pt // n_seg) if n_pt else 0
        dev = model.dev
        J = torch.zeros((n_seg, W, dof), dtype=torch.float64, device=dev)
        if n_pt:
            # the chain rule stays on the device (round 4): one launch, a handful of fp64 tensor ops, ONE copy back - the
            # [n_seg, (W-2) dof] array scipy asked for
            q32 = pts.to(device=dev, dtype=torch.float32).contiguous()
            pd, segd, stepd = p.to(dev), seg.to(dev), step.to(dev)
            if model.C == 1:
                _, h = model.score_hinge_grad_raw(q32, prob.safety_margin, -1.0)  # h_n = d c_n / d dense_n
                h = h.double()
            else:
                # several classes (round 6): the reference flattens the [n_pt, C] costs and cuts the flat vector into n_seg
                # rows (optim.py:199-207), so entry (n, c) belongs to row (n C + c) // (entries per row) and every entry needs
                # its own gradient: the full Jacobian [n_pt, C, dof] in one launch (dcx_score_jac), masked by the hinge
                sc, jac = model.score_jac_raw(q32)
                mg = torch.tensor(list(model.margins(prob.safety_margin)), device=dev, dtype=torch.float32)
                h = (-((sc - mg) > 0).double()[:, :, None] * jac.double()).reshape(n_pt * model.C, dof)
                per = self._flat_per(n_pt, n_seg, model.C)
                segd, stepd = segd.repeat_interleave(model.C), stepd.repeat_interleave(model.C)
            delta = pd[1:] - pd[:-1]
            length = delta.norm(dim=1)
            unit = delta / length[:, None]
            u = unit[segd]                                                  # [n_pt (C), dof]
            scale = (stepd * ms / length[segd])[:, None]
            a = scale * (h - u * (u * h).sum(dim=1, keepdim=True))          # part carried by p_{i+1}
            row = torch.arange(len(h), device=dev) // per
            J.index_put_((row, segd), h - a, accumulate=True)
            J.index_put_((row, segd + 1), a, accumulate=True)
        return J[:, 1:-1].cpu().numpy().reshape(n_seg, -1)

    @staticmethod
    def _flat_per(n_pt, n_seg, C):
        """entries per constraint row when the reference reshapes its flat [n_pt * C (+ padding)] cost vector into
        [n_seg, -1] (optim.py:199-207; the padding is sized for ONE class there, so with several classes the reshape only
        exists when the total happens to divide - the same RuntimeError otherwise)"""
        mult = n_pt // n_seg + (1 if n_pt % n_seg else 0)
        total = n_pt * C + (n_seg * mult - n_pt if n_pt % n_seg else 0)
        if total % n_seg:
            raise RuntimeError(f"shape '[{n_seg}, -1]' is invalid for input of size {total}")
        return total // n_seg

    HESS_FD_STEP = 4e-3  # rad (or m); only where dcx_score_hess reports DCX_ERR_UNSUPPORTED

    def hess_collision(self, x, v):
        """[(W-2)*dof, (W-2)*dof] Hessian of v . collision(x), the `hess` of the reference's trust-constr constraint
        (optim.py:380-391, a double backward through dist_est).  With a fusable dist_est the per-point score
        Hessians are ANALYTIC (`dcx_score_hess`: forward-mode tangents through the FK, the sweep and the reverse FK
        sweep, one launch for every dense point and direction) and the dense-path geometry is chained exactly: the
        second-order Taylor model of hinge(score) around each dense point, composed with dense_n(p), has the same
        Hessian in p as the constraint itself, and autograd differentiates that small fp64 surrogate twice on the
        host.  Otherwise: the reference's route (needs a twice-differentiable dist_est)."""
        v = torch.as_tensor(np.asarray(v), dtype=torch.float64)
        m = self._fused_model()
        if m is not None and self.dense_cap is None:
            return self._hess_collision_fused(x, v, m)
        p = self.prob.full(x)
        H = torch.autograd.functional.hessian(  # counts len(dense) checks, like hess_con_collision_free (optim.py:380-391)
            lambda z: torch.dot(self.prob.segment_collision(z, self.dist_est, self.dense_cap), v), p,
            create_graph=False, strict=False, vectorize=True, outer_jacobian_strategy='reverse-mode')
        W, dof = p.shape
        return H[1:-1, :, 1:-1, :].numpy().reshape((W - 2) * dof, -1)

    def _hess_collision_fused(self, x, v, model):
        prob = self.prob
        p = prob.full(x).detach()
        W, dof = p.shape
        ms = prob.max_speed
        dense, seg, step = utils.dense_path_indexed(p, ms)
        prob.cnt_check += len(dense)  # the reference's Hessian evaluates the constraint once (optim.py:384)
        pts, seg, step = dense[1:-1], seg[1:-1], step[1:-1]
        n_seg, n_pt = W - 1, len(pts)
        if n_pt == 0:
            return np.zeros(((W - 2) * dof, (W - 2) * dof))
        per = -(-n_pt // n_seg)
        q32 = pts.to(device=model.dev, dtype=torch.float32).contiguous()
        dev = model.dev
        if model.C > 1:
            return self._hess_collision_fused_mc(p, v, model, q32, pts, seg, step, n_seg, n_pt)
        s, g = model.score_grad_raw(q32)
        try:
            _, S = model.score_hess_raw(q32)
            S = S.double()
        except _lib.DcxUnsupported:
            # a feature row too wide for the Hessian kernel's LDS in (value, tangent) pairs (none of the reference's robots
            # since round 3: the 23-joint iiwa7 + Allegro tree pages its frames to global memory): central differences of
            # the analytic fused gradient, all 2 * dof probes of every dense point in one launch (step: fp32 gradient
            # noise ~1e-6 / step against step^2 truncation)
            eps = self.HESS_FD_STEP
            probes = pts[:, None, None, :] + eps * torch.stack([torch.eye(dof), -torch.eye(dof)]).to(pts.dtype)[None]
            _, gp = model.score_grad_raw(probes.reshape(-1, dof).to(device=model.dev, dtype=torch.float32).contiguous())
            gp = gp.double().reshape(n_pt, 2, dof, dof)
            S = (gp[:, 0] - gp[:, 1]) / (2 * eps)
        # everything below stays on the device in fp64 (round 4): the second-order surrogate is differentiated twice there
        # and only the [(W-2) dof]^2 result comes back
        s, g = s.double()[:, 0], g.double()
        segd, stepd, ptsd, vd = seg.to(dev), step.to(dev), pts.to(dev), v.to(dev)
        active = -((s - prob.safety_margin) > 0).double() * vd[torch.arange(n_pt, device=dev) // per]  # d(v.c)/d score_n
        S = 0.5 * (S + S.transpose(1, 2)) * active[:, None, None]
        h = g * active[:, None]

        def taylor(z):
            delta = z[1:] - z[:-1]
            unit = delta / delta.norm(dim=1, keepdim=True)
            d = z[segd] + (stepd * ms)[:, None] * unit[segd] - ptsd
            return (h * d).sum() + 0.5 * torch.einsum('ni,nij,nj->', d, S, d)
        H = torch.autograd.functional.hessian(taylor, p.to(dev), vectorize=True)
        return H[1:-1, :, 1:-1, :].cpu().numpy().reshape((W - 2) * dof, -1)

    def _hess_collision_fused_mc(self, p, v, model, q32, pts, seg, step, n_seg, n_pt):
        """several classes: d(v . c) / d score_nc = -1[score_nc > margin_c] v[row(n, c)] is the per-point upstream of ONE
        dcx_score_hess launch (gradient and Hessian of sum_c upstream_nc score_nc); the dense-path geometry is chained as
        for one class"""
        prob, dev, ms = self.prob, model.dev, self.prob.max_speed
        W, dof = p.shape
        C = model.C
        per = self._flat_per(n_pt, n_seg, C)
        sc = model.score_raw(q32)
        mg = torch.tensor(list(model.margins(prob.safety_margin)), device=dev, dtype=torch.float32)
        vrow = v.to(dev)[torch.arange(n_pt * C, device=dev) // per].reshape(n_pt, C)
        up = (-((sc - mg) > 0).double() * vrow).float().contiguous()
        h, S = model.score_hess_raw(q32, up)
        h, S = h.double(), S.double()
        S = 0.5 * (S + S.transpose(1, 2))
        segd, stepd, ptsd = seg.to(dev), step.to(dev), pts.to(dev)

        def taylor(z):
            delta = z[1:] - z[:-1]
            unit = delta / delta.norm(dim=1, keepdim=True)
            d = z[segd] + (stepd * ms)[:, None] * unit[segd] - ptsd
            return (h * d).sum() + 0.5 * torch.einsum('ni,nij,nj->', d, S, d)
        H = torch.autograd.functional.hessian(taylor, p.to(dev), vectorize=True)
        return H[1:-1, :, 1:-1, :].cpu().numpy().reshape((W - 2) * dof, -1)

    def joint_limit(self, x):
        return -self.prob.joint_limit_violation(self.prob.full(x).detach()).item()

    def grad_joint_limit(self, x):
        p = self.prob.full(x)
        v = -self.prob.joint_limit_violation(p)
        (g,) = torch.autograd.grad(v, p, allow_unused=True)
        return np.zeros(len(x)) if g is None else g[1:-1].numpy().reshape(-1)


def _scipy_driver(robot, dist_est, start_cfg, target_cfg, options, run_minimize, dense_cap=None, extra_info=False):
    n_trials = options['NUM_RE_TRIALS']
    seed = options['seed']
    torch.manual_seed(seed)
    prob = _PathProblem(robot, start_cfg, target_cfg, options)
    t0 = time.time()
    if prob.trivial():
        return _trivial_record(prob, robot, seed, t0)
    terms = _ScipyTerms(prob, dist_est, dense_cap)
    best, best_violation, success = None, np.inf, False
    for trial in range(n_trials):
        x0 = prob.make_init(trial)[1:-1].reshape(-1).numpy()
        res = run_minimize(terms, x0)
        if res.success:
            best, success = res, True
            break
        violation = -(terms.collision(res.x).sum() + terms.joint_limit(res.x))
        if violation < best_violation:
            best, best_violation = res, violation
    elapsed = time.time() - t0
    sol = prob.full(best.x).detach()
    extra = {'info': best} if extra_info else {}
    return _record(start_cfg, target_cfg, prob.cnt_check, best.fun, elapsed, success, seed, sol, **extra)


def givengrad_traj_optimize(robot, dist_est, start_cfg, target_cfg, options):
    """SLSQP with analytic gradients: minimise path length s.t. per-segment collision >= 0, joint limits >= 0"""
    max_iter = options['MAXITER']

    def run(terms, x0):
        return minimize(terms.cost, x0, jac=terms.grad_cost, method='slsqp',
                        constraints=[{'fun': terms.collision, 'type': 'ineq', 'jac': terms.jac_collision},
                                     {'fun': terms.joint_limit, 'type': 'ineq', 'jac': terms.grad_joint_limit}],
                        options={'maxiter': max_iter, **options.get('extra_optimizer_options', {})})
    return _scipy_driver(robot, dist_est, start_cfg, target_cfg, options, run)


def trustconstr_traj_optimize(robot, dist_est, start_cfg, target_cfg, options):
    """trust-constr with analytic first derivatives and a Hessian of the collision constraint (reference
    optim.py:486-492).  options['constraint_hessian']: 'auto' (default) = the fused analytic Hessian
    (dcx_score_hess) when dist_est is a fusable diffco_amd score, else a BFGS model; 'fused' / 'autograd' (the
    reference's double backward through dist_est) / 'bfgs' force one."""
    max_iter = options['MAXITER']
    mode = options.get('constraint_hessian', 'auto')
    if mode not in ('auto', 'fused', 'autograd', 'bfgs'):
        raise ValueError(f"constraint_hessian: {mode!r}")

    def run(terms, x0):
        fused = terms._fused_model() is not None and terms.dense_cap is None
        if mode == 'fused' and not fused:
            raise ValueError("constraint_hessian='fused' needs dist_est to be a diffco_amd score of this robot")
        if mode == 'autograd':
            terms._model = None
        hess = BFGS() if mode == 'bfgs' or (mode == 'auto' and not fused) else terms.hess_collision
        return minimize(terms.cost, x0, jac=terms.grad_cost, method='trust-constr',
                        constraints=[NonlinearConstraint(terms.collision, 0, np.inf, jac=terms.jac_collision, hess=hess),
                                     NonlinearConstraint(terms.joint_limit, 0, np.inf, jac=terms.grad_joint_limit,
                                                         hess=BFGS())],
                        options={'maxiter': max_iter, **options.get('extra_optimizer_options', {})})
    return _scipy_driver(robot, dist_est, start_cfg, target_cfg, options, run, extra_info=True)


def gradient_free_traj_optimize(robot, checker, start_cfg, target_cfg, options: Dict = None):
    """trust-constr with finite-difference derivatives on a (possibly non-differentiable) `checker(q) -> score`"""
    max_iter = options['MAXITER']
    cap = options.get('max_dense_waypoints', None)
    saved_margin = options.get('safety_margin', 0.0)
    options = dict(options, safety_margin=0.0)  # the reference's gradient-free constraint uses the raw checker output

    def run(terms, x0):
        return minimize(terms.cost, x0, method='trust-constr',
                        constraints=[NonlinearConstraint(terms.collision, 0, np.inf),
                                     NonlinearConstraint(terms.joint_limit, 0, np.inf)],
                        options={'maxiter': max_iter, **options.get('extra_optimizer_options', {})})
    rec = _scipy_driver(robot, checker, start_cfg, target_cfg, options, run, dense_cap=cap)
    options['safety_margin'] = saved_margin
    return rec
