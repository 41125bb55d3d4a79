"""Escape from collision by gradient steps on the configuration itself (SURVEY.md §8f-2 lists it beside the trajectory step).

`OptimSampler` keeps the constructor, option keys and return values of the reference's class (scripts/escape.py:4-38, used by
scripts/2d_escape.py:98-110 and scripts/compare_sampling.py:177-195): starting from a configuration in collision it repeats

    excess = sum(dist_est(q) - safety_margin);  stop when excess <= 0;  q <- optimizer step on excess;  q <- post_transform(q)

at most N_WAYPOINTS times and returns the recorded configurations and the number of `dist_est` evaluations.

When `dist_est` is a score method of a diffco_amd checker (or a `ScoreModel`), the optimiser is `torch.optim.Adam` with
lr / betas / eps only and `post_transform` is None, `utils.wrap2pi` or `utils.se2_wrap2pi`, the whole loop is enqueued on the
GPU by ONE library call (`dcx_escape_adam`, include/dcx.h): per step the fused score + gradient sweep and one update launch,
the loop's decisions taken on the device, one read-back at the end (the reference synchronises every step, escape.py:27).
`optim_escape_batch` advances B INDEPENDENT loops together in the same launches - what compare_sampling.py's sampling loop
does one configuration at a time.  Anything else (a foreign `dist_est`, another optimiser or transform) runs the same loop on
the host through autograd; the score inside it is still whatever `dist_est` computes.
"""
import ctypes as C

import torch

from . import _lib, _ops, utils


def resampling_escape(robot, *args, **kwargs):
    """one uniform sample of the joint limits, [1, dof] (escape.py:40-43)"""
    lo, hi = robot.limits[:, 0], robot.limits[:, 1]
    return torch.rand(1, robot.dof) * (hi - lo) + lo


class OptimSampler:
    def __init__(self, robot, dist_est, args=None):
        args = dict(args or {})
        self.robot, self.dist_est = robot, dist_est
        self.N_WAYPOINTS = args.get('N_WAYPOINTS', 20)
        self.safety_margin = args.get('safety_margin', -0.3)
        self.lr = args.get('lr', 5e-2)
        self.record_freq = args.get('record_freq', 1)
        self.post_transform = args.get('post_transform', None)
        self.opt_args = args.get('opt_args', {'lr': self.lr})
        self.optimizer = args.get('optimizer', torch.optim.Adam)
        self.last_route = None   # 'fused' / 'host': which way the last call went (tests)

    # ---- what can be fused ------------------------------------------------------------------------------------------
    def _wrap_mask(self, dof):
        if self.post_transform is None:
            return 0
        if self.post_transform is utils.wrap2pi:
            return (1 << dof) - 1
        if self.post_transform is utils.se2_wrap2pi and dof >= 3:
            return 1 << 2
        return None

    def _adam(self):
        """(lr, beta1, beta2, eps) when the optimiser is plain Adam, else None"""
        if self.optimizer is not torch.optim.Adam:
            return None
        o = dict(self.opt_args)
        lr, (b1, b2), eps = o.pop('lr', 1e-3), o.pop('betas', (0.9, 0.999)), o.pop('eps', 1e-8)
        if o.pop('weight_decay', 0) or o.pop('amsgrad', False) or o.pop('maximize', False):
            return None
        for k in ('foreach', 'capturable', 'differentiable', 'fused'):
            o.pop(k, None)
        return None if o else (float(lr), float(b1), float(b2), float(eps))

    def _model(self):
        from .traj import _resolve_model
        try:
            return _resolve_model(self.dist_est)
        except TypeError:
            return None

    def _plan(self, dof):
        """(model, adam, wrap_mask) when the loop can run as dcx_escape_adam, else None"""
        adam, mask = self._adam(), self._wrap_mask(dof)
        if adam is None or mask is None or int(self.N_WAYPOINTS) < 1:
            return None
        model = self._model()
        if model is None or model.dof != dof:
            return None
        return model, adam, mask

    def _margin_on(self, dev, n_classes):
        """safety_margin as [C] floats on the device, copied over once per value (a host-to-device copy per escape otherwise)"""
        m = self.safety_margin
        key = (dev, n_classes, tuple(m.detach().reshape(-1).tolist()) if torch.is_tensor(m) else float(m))
        if getattr(self, "_margin_key", None) != key:
            t = torch.as_tensor(m, dtype=torch.float32).detach().reshape(-1)
            if t.numel() not in (1, n_classes):
                raise ValueError(f"safety_margin has {t.numel()} entries, the score has {n_classes} columns")
            self._margin_dev, self._margin_key = t.expand(n_classes).contiguous().to(dev), key
        return self._margin_dev

    # ---- the fused loop ---------------------------------------------------------------------------------------------
    def _fused(self, plan, q0, joint, want_history, compact_every=0):
        """q0 [B, dof] -> (final [B, dof], steps [n_loops, 2] int32 (evaluations, Adam steps), history or None), on the GPU"""
        model, (lr, b1, b2, eps), mask = plan
        lib, dev = _lib.require_gpu(), model.dev
        q = q0.detach().to(device=dev, dtype=torch.float32).contiguous().clone()
        B, dof = q.shape
        margin = self._margin_on(dev, model.C)
        n, rf = int(self.N_WAYPOINTS), int(self.record_freq or 0)
        opts = _lib.EscapeOpts(lr, b1, b2, eps, n, rf, 1 if joint else 0, 0 if joint else int(compact_every), mask)
        steps = torch.empty((1 if joint else B, 2), device=dev, dtype=torch.int32)
        # slots behind a loop's last record are never returned as they are (optim_escape cuts, optim_escape_batch overwrites)
        hist = torch.empty((((n + rf - 1) // rf if rf else 0) + 1, B, dof), device=dev, dtype=torch.float32) if want_history else None
        # (no lease on the model: a refill of its rows - ScoreModel.update - is enqueued behind this loop on the same stream, and
        # waits for the streams recorded by _st() otherwise)
        with _ops._on_device(dev):
            work = torch.empty(int(lib.dcx_escape_work_bytes(model._h, B)), device=dev, dtype=torch.uint8)
            _lib.check(lib.dcx_escape_adam(model._h, _ops._ptr(q), B, _ops._ptr(margin), C.byref(opts), _ops._ptr(work),
                                           work.numel(), _ops._ptr(hist), _ops._ptr(steps), model._st()))
        return q, steps, hist

    # ---- the host loop (foreign dist_est / optimiser / transform) ---------------------------------------------------
    def _host(self, start_cfg):
        p = start_cfg.clone().requires_grad_(True)
        opt = self.optimizer([p], **self.opt_args)
        kept, evaluations = [], 0
        for step in range(self.N_WAYPOINTS):
            excess = torch.sum(self.dist_est(p) - self.safety_margin)
            evaluations += 1
            if excess <= 0:
                break
            if self.record_freq and step % self.record_freq == 0:
                kept.append(p.detach().clone())
            opt.zero_grad()
            excess.backward()
            opt.step()
            if self.post_transform:
                p.data = self.post_transform(p.data)
        kept.append(p.detach().clone())
        return torch.stack(kept, dim=0), evaluations

    # ---- the reference's entry point --------------------------------------------------------------------------------
    def optim_escape(self, start_cfg):
        """(recorded configurations [n_records, *start_cfg.shape], evaluations of dist_est) - escape.py:19-38.  The whole
        of `start_cfg` is ONE loop: the excess is summed over all its rows and classes, as the reference sums it."""
        start_cfg = torch.as_tensor(start_cfg)
        dof = start_cfg.shape[-1] if start_cfg.ndim else 0
        plan = self._plan(dof) if start_cfg.ndim >= 1 and start_cfg.numel() else None
        if plan is None:
            self.last_route = 'host'
            return self._host(start_cfg)
        self.last_route = 'fused'
        _, steps, hist = self._fused(plan, start_cfg.reshape(-1, dof), True, True)
        evaluations, taken = (int(v) for v in steps[0].tolist())   # the one read-back
        rf = int(self.record_freq or 0)
        n_rec = ((taken + rf - 1) // rf if rf else 0) + 1
        out = hist[:n_rec].reshape(n_rec, *start_cfg.shape)
        return out.to(device=start_cfg.device, dtype=start_cfg.dtype), evaluations

    COMPACT_FROM = 16384   # batches from this size on take stopped loops out of the sweep every COMPACT_EVERY steps
    COMPACT_EVERY = 4

    def optim_escape_batch(self, start_cfgs, history=False, compact_every=None):
        """B independent escape loops advanced together: `start_cfgs` [B, dof] -> (final configurations [B, dof], evaluations
        [B] int64); with history=True also (records [n_slots, B, dof], n_records [B]) - row b's records are
        records[:n_records[b], b], as `optim_escape(start_cfgs[b:b+1])` would return them; later slots repeat its final
        configuration.  A fused plan is required (there is nothing batched about the host loop).
        compact_every: 0 = every step sweeps all B configurations and nothing synchronises; k > 0 = after every k-th step the
        loops that stopped are taken out of the sweep (one stream synchronisation each time; the call returns as soon as every
        loop has stopped); None = COMPACT_EVERY for B >= COMPACT_FROM (profiles/r05_escape.txt), else 0."""
        start_cfgs = torch.as_tensor(start_cfgs)
        if start_cfgs.ndim != 2:
            raise ValueError("optim_escape_batch takes [B, dof]")
        plan = self._plan(start_cfgs.shape[1])
        if plan is None:
            raise TypeError("optim_escape_batch needs a diffco_amd score method (or ScoreModel) as dist_est, torch.optim.Adam "
                            "with lr / betas / eps, and post_transform None / utils.wrap2pi / utils.se2_wrap2pi")
        self.last_route = 'fused'
        if len(start_cfgs) == 0:
            e = torch.zeros(0, dtype=torch.int64)
            return (start_cfgs.clone(), e) if not history else (start_cfgs.clone(), e, start_cfgs.new_zeros((1, 0, start_cfgs.shape[1])), e)
        if compact_every is None:
            compact_every = self.COMPACT_EVERY if len(start_cfgs) >= self.COMPACT_FROM else 0
        q, steps, hist = self._fused(plan, start_cfgs, False, history, compact_every)
        back = dict(device=start_cfgs.device, dtype=start_cfgs.dtype)
        final, evaluations = q.to(**back), steps[:, 0].to(device=start_cfgs.device, dtype=torch.int64)
        if not history:
            return final, evaluations
        rf = int(self.record_freq or 0)
        taken = steps[:, 1].to(torch.int64)
        n_rec = (torch.div(taken + rf - 1, rf, rounding_mode='floor') if rf else torch.zeros_like(taken)) + 1
        slot = torch.arange(hist.shape[0], device=hist.device)[:, None]
        hist = torch.where((slot < n_rec[None, :])[..., None], hist, q[None])
        return final, evaluations, hist.to(**back), n_rec.to(start_cfgs.device)
