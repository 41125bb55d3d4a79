"""DiffCo — the kernel-perceptron proxy collision checker (new-API generation).

Drop-in for diffco/kernel_perceptrons.py:31-370 (class DiffCo): same constructor, `train`,
`fit_poly`, `score` / `score_original`, `poly_score(point=None, transformed_point=None)`,
`is_collision`, `to` / `cuda`, `device`, `valid_supports`, and the same state attributes
(`support_points`, `support_transformed`, `gains`, `rbf_nodes`, `hypothesis`, `y`, `distance`,
`kernel_matrix`).  What differs is where the arithmetic runs:

  * `score` / `poly_score` (+ their gradient) are ONE fused HIP launch (FK -> kernel block ->
    weight contraction -> analytic gradient), never a materialised K[B,S] and never on the CPU.
  * `train` keeps the reference's sequential perceptron on the host; kernel rows come from the
    HIP kernel-matrix kernel.  `fit_poly` solves the S x S system on the GPU (`_ops.solve`: library LU).

The old-API classes (`DiffCo(obstacles, ...)`, `MultiDiffCo`, `DiffCoBeta`) are in
diffco_amd/deprecated.py, like the reference keeps them in diffco/deprecated/.
"""
from time import time

import torch as th

from . import _ops, kernel
from ._perceptron import FusedScorer, device_trainer_spec, fit_system, run_trainer, sub_block


class Perceptron:
    def __init__(self):
        self.support_points = None

    def score(self, point):
        raise NotImplementedError

    def predict(self, point):
        return self.score(point) > 0

    def line_predict(self, start, target, res=50):
        """is any of the `res` points start + (target - start) i / res, i = 0 .. res - 1, in collision?  (reference
        kernel_perceptrons.py:22-24 asks `is_collision` once per point; here the points are ONE batch: one launch)"""
        return _line_query(self, start, target, res)

    def __call__(self, *args, **kwargs):
        return self.predict(*args, **kwargs)


def _line_query(checker, start, target, res):
    import torch
    start, target = torch.as_tensor(start), torch.as_tensor(target)
    frac = torch.arange(res, dtype=start.dtype if start.is_floating_point() else torch.float32, device=start.device) / res
    points = start[None, :] + (target - start)[None, :] * frac[:, None]
    return bool(torch.as_tensor(checker.is_collision(points)).any())


class DiffCo(Perceptron):
    def __init__(self, kernel_func='rq', gamma=1, beta=1, transform=None, max_batch_size=None,
                 max_num_supports=None):
        super().__init__()
        self.train_method = None
        self.kernel_func = kernel.RQKernel(gamma) if isinstance(kernel_func, str) and kernel_func == 'rq' else kernel_func
        self.beta = beta
        self.transform = transform
        self._cuda = False
        self.support_transformed = None
        self.gains = None
        self.hypothesis = None
        self.y = None
        self.distance = None
        self.kernel_matrix = None
        self.rbf_nodes = None
        self.rbf_kernel = None
        self.max_batch_size = max_batch_size
        self.max_num_supports = max_num_supports  # fixed-size, zero-padded state when set
        self._valid_supports = 0
        self._score_fused, self._poly_fused, self._polyx_fused = FusedScorer(), FusedScorer(), FusedScorer()

    # pickling keeps only the torch/python state; device handles are rebuilt lazily
    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop('_score_fused', None)
        st.pop('_poly_fused', None)
        st.pop('_polyx_fused', None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._score_fused, self._poly_fused, self._polyx_fused = FusedScorer(), FusedScorer(), FusedScorer()

    def _invalidate_fused(self):
        """drop the cached device models: the state they were built from is about to change"""
        for f in (self._score_fused, self._poly_fused, self._polyx_fused):
            f.invalidate()

    @property
    def valid_supports(self):
        return self._valid_supports

    @property
    def device(self):
        return self.support_points.device

    # ------------------------------------------------------------------------------ training
    def train(self, X, y, update=False, exist_mask=None, max_iteration=1000, method='original', distance=None,
              verbose=False):
        if X.device.type == 'cuda':
            self._cuda = True
        self.train_method = method
        self.distance = distance.reshape(-1) if distance is not None else None
        t0 = time()
        self.train_perceptron(X, y, update=update, exist_mask=exist_mask, max_iteration=max_iteration, verbose=verbose)
        if verbose:
            print(f'DiffCo training done. {time() - t0:.4f} secs cost')

    def _features(self, X):
        return X if self.transform is None else self.transform(X)

    def initialize(self, X, y):
        Xt = self._features(X)
        if len(X) <= 10000:  # small problems train on the host, as the reference does
            X, Xt, y = X.cpu(), Xt.cpu(), y.cpu()
        y = y.reshape(-1)
        assert len(y) == len(X)
        n = len(X)
        gains = th.zeros(n, dtype=X.dtype, device=X.device)
        hypo = th.zeros(n, dtype=X.dtype, device=X.device)
        return gains, X, Xt, None, hypo, y  # the n x n kernel matrix is created where the trainer runs

    def jump_start_initialize(self, X, y, exist_mask):
        """Warm start for active learning: rows flagged in exist_mask are the current supports (same
        order); the rest are new samples whose hypothesis is the current score."""
        n = len(X)
        # (the reference assembles the warm start on the host for n <= 10000; the trainer and fit_poly run on the GPU here,
        # so state that already lives there stays there: on a many-core host the CPU tensor ops of this function - an n x n
        # zero fill, three masked scatters - cost more than the training itself, 20 of a 57 ms update round)
        # `home`: where the reference keeps the result (and where the caller finds the state afterwards); `dev`: where the
        # assembly runs - the trainer's GPU for our kernels
        home = th.device('cpu') if n <= 10000 else X.device
        # (the rule is "will the DEVICE trainer run?", not "is the kernel ours / is X on the GPU": the host loop -
        # a foreign kernel callable, an FKKernel, DCX_HOST_TRAINER=1 - updates hypo / gains on `home` row by row and needs the
        # matrix it fills beside them, so then everything is assembled on `home`, as the reference does)
        on_device = device_trainer_spec(self.kernel_func) is not None and th.cuda.is_available()
        dev = (X.device if X.is_cuda else _ops._device(None)) if on_device else home
        exist_mask = exist_mask.to(th.bool)
        # one pair of index vectors, made where the mask lives and copied once; every scatter below is an index_copy_ with
        # them (a boolean-mask assignment on a CUDA tensor is a nonzero + a device synchronisation each time: eight of them
        # were half of this function's 1.5 - 2 ms)
        ei_h, ni_h = th.where(exist_mask)[0], th.where(~exist_mask)[0]
        ei, ni = ei_h.to(dev), ni_h.to(dev)
        novel = X.index_select(0, ni_h.to(X.device))
        v = self.valid_supports
        assert n - len(novel) == v
        hypo = th.zeros(n, dtype=X.dtype, device=dev)
        hypo.index_copy_(0, ei, self.hypothesis[:v].to(device=dev, dtype=X.dtype))
        hypo.index_copy_(0, ni, self.score_original(novel).detach().to(device=dev, dtype=X.dtype).reshape(-1))
        novel_t = (novel if self.transform is None else self.transform(novel)).detach().to(dev)
        sup_t = self.support_transformed[:v].to(dev)
        K = th.zeros((n, n), dtype=X.dtype, device=dev)
        K[ei[:, None], ei[None, :]] = self.kernel_matrix[:v, :v].to(dev)
        if len(ni) and len(ei):
            cross = self.kernel_func(sup_t, novel_t).to(dev)
            K[ei[:, None], ni[None, :]] = cross.reshape(len(ei), len(ni))
            K[ni[:, None], ei[None, :]] = cross.reshape(len(ei), len(ni)).T
        Xt = th.zeros((n,) + tuple(novel_t.shape[1:]), dtype=self.support_transformed.dtype, device=dev)
        Xt.index_copy_(0, ei, sup_t)
        Xt.index_copy_(0, ni, novel_t.to(Xt.dtype))
        gains = th.zeros(n, dtype=X.dtype, device=dev)
        gains.index_copy_(0, ei, self.gains[:v].to(device=dev, dtype=X.dtype))
        check = K @ gains
        assert th.allclose(check, hypo, atol=1e-4), f"diff: {th.abs(check - hypo).max()}"
        # (the n x n matrix stays where it was assembled - the device trainer's GPU, or `home` for the host loop: callers only
        # ever gather the support sub-block of it)
        return gains.to(home), X.to(home), Xt.to(home), K, hypo.to(home), y.to(home).reshape(-1)

    def train_perceptron(self, X, y, update=False, exist_mask=None, max_iteration=1000, verbose=False):
        if update:
            gains, X, Xt, K, hypo, y = self.jump_start_initialize(X, y, exist_mask)
        else:
            gains, X, Xt, K, hypo, y = self.initialize(X, y)
        t0 = time()
        progress = None
        if verbose:
            from tqdm import tqdm
            print('DiffCo training...')
            progress = tqdm(total=max_iteration, ncols=0)
        # persistent device trainer for diffco_amd kernels (one launch for the whole loop), host loop otherwise
        gains, hypo, K, it = run_trainer(self.kernel_func, Xt, y, gains, hypo, K, self.beta, max_iteration, progress,
                                         cold=not update)
        self._invalidate_fused()
        if verbose:
            progress.close()
            print(f'Ended at iteration {it}, cost {time() - t0:.4f} secs')
            print('ACC: {}'.format(th.sum((hypo > 0) == (y > 0)) / float(len(y))))

        keep = gains != 0
        if keep.sum() < 2:  # always keep at least two supports
            keep[th.where(~keep)[0][0]] = True
        idx = th.where(keep)[0]
        if self.max_num_supports is None:
            # (index_select on the kept rows, not boolean-mask indexing: on CPU tensors the latter opens an OpenMP region
            #  for the [n, m, d] features, and on a 128-thread host waking the pool costs more than the training - 13 ms
            #  of an update round, tools/facade_lines.py)
            self.support_points = X.index_select(0, idx)
            self.support_transformed = Xt.index_select(0, idx)
            self.hypothesis = hypo.index_select(0, idx)
            self.y = y.index_select(0, idx)
            self.distance = self.distance.to(keep.device).index_select(0, idx) if self.distance is not None else None
            self.gains = gains.index_select(0, idx)
            self.rbf_nodes = self.gains.new_zeros(len(self.gains))
            self.kernel_matrix = sub_block(K, idx, gains.device, gains.dtype)
            self._valid_supports = len(self.support_points)
            return
        # fixed-size state: pad with zeros / truncate to max_num_supports
        M = self.max_num_supports
        dist_src = None if self.distance is None else self.distance.to(keep.device).clone()
        if self.support_points is None or len(self.support_points) != M:
            self.support_points = th.zeros((M, X.shape[1]), dtype=X.dtype, device=X.device)
            self.support_transformed = th.zeros((M,) + tuple(Xt.shape[1:]), dtype=Xt.dtype, device=Xt.device)
            self.hypothesis = th.zeros(M, dtype=hypo.dtype, device=hypo.device)
            self.y = th.zeros(M, dtype=y.dtype, device=y.device)
            self.gains = th.zeros(M, dtype=gains.dtype, device=gains.device)
            self.kernel_matrix = th.zeros((M, M), dtype=gains.dtype, device=gains.device)
        if dist_src is not None:
            self.distance = th.zeros(M, dtype=dist_src.dtype, device=dist_src.device)
        if len(idx) > M:
            # (sic) the reference keeps the M entries with the SMALLEST |gain| (kernel_perceptrons.py:174-178)
            pick = th.topk(gains.abs(), M, largest=False).indices
            keep.zero_()
            keep[pick] = True
            idx = th.where(keep)[0]
        n = len(idx)
        for buf, src in ((self.support_points, X), (self.support_transformed, Xt), (self.hypothesis, hypo),
                         (self.y, y), (self.gains, gains)):
            buf.zero_()
            buf[:n] = src[idx]
        if dist_src is not None:
            self.distance[:n] = dist_src[idx]
        self.rbf_nodes = self.gains.new_zeros(len(self.gains))
        self.kernel_matrix.zero_()
        self.kernel_matrix[:n, :n] = sub_block(K, idx, gains.device, gains.dtype)
        self._valid_supports = n
        resid = th.abs(self.hypothesis - self.kernel_matrix @ self.gains).max()
        assert resid <= 1e-4 + 1e-8, f"diff: {resid}"

    def filter_support_points_(self, mask):
        self._invalidate_fused()
        idx = th.where(mask)[0]
        self.support_points = self.support_points[mask]
        self.support_transformed = self.support_points if self.transform is None else self.support_transformed[mask]
        self.hypothesis = self.hypothesis[mask]
        self.y = self.y[mask]
        self.distance = self.distance[mask] if self.distance is not None else None
        self.gains = self.gains[mask]
        self.kernel_matrix = self.kernel_matrix[idx[:, None], idx[None, :]]

    def fit_poly(self, kernel_func, target='hypo'):
        """rbf_nodes = solve(K_rbf(supports, supports), target values)"""
        if target == 'hypo':
            t = self.hypothesis
        elif 'dist' in target:
            t = self.distance
        elif 'label' in target:
            t = self.y
        else:
            raise ValueError(f"unknown fit target {target!r}")
        self.rbf_kernel = kernel_func
        self._invalidate_fused()
        v = self.valid_supports
        Xs = self.support_transformed[:v]
        self.rbf_nodes.zero_()
        self.rbf_nodes[:v] = fit_system(self.rbf_kernel, Xs, t[:v, None].to(Xs.dtype)).reshape(-1)
        if self._cuda:
            self.cuda()

    # ------------------------------------------------------------------------------ devices
    def cuda(self):
        self.to(th.device('cuda'))

    def to(self, device):
        device = th.device(device)
        self._invalidate_fused()
        for name in ('support_points', 'support_transformed', 'rbf_nodes', 'gains'):
            t = getattr(self, name, None)
            if t is not None:
                setattr(self, name, t.to(device))
        self._cuda = device.type == 'cuda'

    # ------------------------------------------------------------------------------ the hot path
    def is_collision(self, point):
        return self.score(point) > 0

    def score(self, point):
        return self.score_original(point)

    def score_original(self, point):
        """sum_j K(T(q), support_j) gains_j  ->  [N]  (0-dim for a single query under RQ, like the reference)"""
        single = point.ndim == 1
        if single:
            point = point[None, :]
        s = self._score_fused.score(self.transform, self.kernel_func, self.support_transformed, self.gains, point)
        s = s.reshape(-1)
        if s.shape[0] == 1 and isinstance(self.kernel_func, (kernel.RQKernel, kernel.MultiQuadratic)):
            s = s.reshape(())
        return s

    def poly_score(self, point=None, transformed_point=None):
        """sum_j K_rbf(T(q), support_j) rbf_nodes_j  ->  [N, 1]; `transformed_point` skips the FK."""
        if transformed_point is None:
            if point.ndim == 1:
                point = point.unsqueeze(0)
            point = point.to(device=self.rbf_nodes.device, dtype=self.rbf_nodes.dtype)
            return self._poly_fused.score(self.transform, self.rbf_kernel, self.support_transformed, self.rbf_nodes,
                                          point)
        feats = transformed_point.reshape(len(transformed_point), -1)
        return self._polyx_fused.score(None, self.rbf_kernel, self.support_transformed, self.rbf_nodes, feats)

    def poly_score_and_grad(self, point):
        """(poly_score(point) [N, 1], d poly_score / d point [N, dof]) from ONE fused launch, without an autograd graph.
        Not in the reference (its callers write `s = poly_score(p); s.sum().backward()`, optim.py:88-101): an extension for loops
        that want the gradient at the cost of the raw call - 11 us per call where the route through torch's autograd engine costs
        60 - 140 (profiles/r05_api_latency.txt).  Needs a fusable transform (a diffco_amd robot's `fkine`, or none)."""
        if point.ndim == 1:
            point = point.unsqueeze(0)
        point = point.to(device=self.rbf_nodes.device, dtype=self.rbf_nodes.dtype)
        dev = point.device if point.device.type == "cuda" else (
            self.support_transformed.device if self.support_transformed.device.type == "cuda" else None)
        m = self._poly_fused.model(self.transform, self.rbf_kernel, self.support_transformed, self.rbf_nodes, dev)
        if m.desc.kind == 0 and self.transform is not None:
            raise TypeError("poly_score_and_grad needs a fusable transform (a diffco_amd robot's fkine): a foreign transform's "
                            "Jacobian is only available through its own autograd - use poly_score")
        return m.score_and_grad(point.reshape(-1, m.dof))
