"""Old-API checkers that the reference's scripts/*.py are written against.

Drop-in for diffco/deprecated/DiffCo.py:29-260 (`DiffCo(obstacles, kernel_func, gamma, beta,
gt_checker)`, `train`, `fit_poly(kernel_func, target, fkine)`, `rbf_score` / `poly_score`, `score`),
deprecated/MultiDiffCo.py:19-207 (`MultiDiffCo`, labels y[N, C], `rbf_score -> [N, C]`) and
deprecated/DiffCoBeta.py:14-181 (`rbf_score`).  State attribute names follow the old generation
(`support_fkine`, `fkine`).  Score methods run the fused HIP kernels; training is the host
perceptron of diffco_amd/_perceptron.py.  Ground-truth collision checking (FCL) is out of scope:
`obstacles` / `gt_checker` are stored and otherwise unused.
"""
from time import time

import torch

from . import kernel
from ._perceptron import FusedScorer, fit_system, run_trainer, solve_system, sub_block


class CollisionChecker:
    def __init__(self, obstacles=None):
        self.obstacles = obstacles

    def predict(self, point):
        return self.score(point) > 0

    def line_collision(self, start, target, res=50):
        """any of the `res` points from start towards target in collision (deprecated/DiffCo.py:20-22), as ONE batch"""
        from .kernel_perceptrons import _line_query
        return _line_query(self, start, target, res)

    def __call__(self, *args, **kwargs):
        return self.predict(*args, **kwargs)


def _split_kernel(kernel_func):
    """(feature transform or None, point kernel) of an old-API kernel object"""
    if isinstance(kernel_func, kernel.FKKernel):
        return kernel_func.fkine, kernel_func.rq_kernel
    return None, kernel_func


class DiffCo(CollisionChecker):
    def __init__(self, obstacles=None, kernel_func='rq', gamma=1, beta=1, gt_checker=None):
        super().__init__(obstacles)
        self.gt_checker = gt_checker
        self.train_method = None
        self.kernel_func = kernel.RQKernel(gamma) if isinstance(kernel_func, str) and kernel_func == 'rq' else kernel_func
        self.beta = beta
        self.fkine = None
        self._cuda = False
        self.support_points = self.support_fkine = None
        self.gains = self.hypothesis = self.y = self.distance = self.kernel_matrix = None
        self.rbf_nodes = self.rbf_kernel = None
        self._score_fused, self._rbf_fused = FusedScorer(), FusedScorer()
        self._score_feats = None  # features of the supports under the training kernel's transform

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ('_score_fused', '_rbf_fused'):
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._score_fused, self._rbf_fused = FusedScorer(), FusedScorer()

    def _reset_caches(self):
        """the supports / weights are about to change: drop the cached FK features and device models"""
        self._score_feats = None
        self._score_fused.invalidate()
        self._rbf_fused.invalidate()

    # ------------------------------------------------------------------------------ training
    def initialize(self, X, y):
        self._reset_caches()
        self.support_points = X.clone()
        self.y = y.reshape(-1).clone()
        assert len(self.y) == len(X)
        n = len(X)
        self.gains = torch.zeros(n, dtype=X.dtype)
        self.kernel_matrix = None  # created where the trainer runs
        self.hypothesis = torch.zeros(n, dtype=X.dtype)

    def _train_inputs(self):
        """(point kernel, features of all samples under the training kernel's transform)"""
        tf, point_kernel = _split_kernel(self.kernel_func)
        feats = self.support_points if tf is None else tf(self.support_points).reshape(len(self.support_points), -1)
        return point_kernel, feats.detach()

    def _run_trainer(self, max_iteration, cold):
        pk, feats = self._train_inputs()
        self.gains, self.hypothesis, self.kernel_matrix, it = run_trainer(
            pk, feats, self.y, self.gains, self.hypothesis, self.kernel_matrix, self.beta, max_iteration, cold=cold)
        return it

    def train_perceptron(self, X, y, max_iteration=1000):
        self.initialize(X, y)
        it = self._run_trainer(max_iteration, cold=True)
        print('Ended at iteration {}'.format(it))
        print('ACC: {}'.format(torch.sum((self.hypothesis > 0) == (self.y > 0)) / float(self.y.numel())))

    def train(self, X, y, max_iteration=1000, method='original', distance=None, keep_all=False):
        if method != 'original':
            raise NotImplementedError(f"training method {method!r} (the reference's sgd/svm variants are dead code)")
        self.train_method = method
        self.distance = distance.reshape(-1) if distance is not None else None
        t0 = time()
        self.train_perceptron(X, y, max_iteration)
        if not keep_all:
            self.filter_support_points_(self.gains != 0)
        print('{} training done. {:.4f} secs cost'.format(method, time() - t0))

    def filter_support_points_(self, mask):
        idx = torch.where(mask)[0]
        self.support_points = self.support_points[mask]
        self.hypothesis = self.hypothesis[mask]
        self.y = self.y[mask]
        self.distance = self.distance[mask] if self.distance is not None else None
        self.gains = self.gains[mask]
        self.kernel_matrix = sub_block(self.kernel_matrix, idx, self.gains.device, self.gains.dtype)
        self._reset_caches()

    # ------------------------------------------------------------------------------ spline fit
    def _fit_inputs(self, kernel_func, target, fkine):
        self._reset_caches()
        X = self.support_points
        if fkine is not None:
            X = fkine(X).reshape([len(X), -1]).detach()
            self.fkine = fkine
            self.support_fkine = X
        if target == 'hypo':
            t = self.hypothesis
        elif 'dist' in target:
            t = self.distance
        elif 'label' in target:
            t = self.y
        else:
            raise ValueError(f"unknown fit target {target!r}")
        self.rbf_kernel = kernel.MultiQuadratic(1) if kernel_func is None else kernel_func
        return X, t

    def fit_poly(self, kernel_func=None, target='hypo', fkine=None):
        X, t = self._fit_inputs(kernel_func, target, fkine)
        self.rbf_nodes = fit_system(self.rbf_kernel, X, t.reshape(len(X), 1).to(X.dtype)).reshape(-1)
        if self._cuda:
            self.cuda()

    def cuda(self):
        self.to(torch.device('cuda'))

    def to(self, device):
        device = torch.device(device)
        for name in ('support_points', 'support_fkine', 'rbf_nodes', 'gains'):
            t = getattr(self, name, None)
            if t is not None:
                setattr(self, name, t.to(device))
        self._reset_caches()
        self._cuda = device.type == 'cuda'

    @property
    def device(self):
        return self.support_points.device

    # ------------------------------------------------------------------------------ the hot path
    def _score_state(self):
        tf, point_kernel = _split_kernel(self.kernel_func)
        if tf is None:
            return None, point_kernel, self.support_points
        if self._score_feats is None or len(self._score_feats) != len(self.support_points):
            self._score_feats = tf(self.support_points).reshape(len(self.support_points), -1).detach()
        return tf, point_kernel, self._score_feats

    def score(self, point):
        """K(point, supports) @ gains with the training kernel (FK fused when it is an FKKernel)"""
        single = point.ndim == 1
        if single:
            point = point[None, :]
        tf, pk, feats = self._score_state()
        s = self._score_fused.score(tf, pk, feats, self.gains, point)
        if self.gains.ndim == 1:
            s = s.reshape(-1)
            if single and isinstance(pk, (kernel.RQKernel, kernel.MultiQuadratic)):
                s = s.reshape(())
        return s

    score_original = score

    def is_collision(self, point):
        return self.score(point) > 0

    def rbf_score(self, point):
        """K_rbf(fkine(point), support_fkine) @ rbf_nodes -> [N, 1]; one query under an RQ / MultiQuadratic spline -> [1]
        ([C] for MultiDiffCo): those kernels drop the row axis of a single query (kernel.py:26-27, 56-57) before the
        reference's matmul with rbf_nodes[:, None], Polyharmonic does not"""
        if point.ndim == 1:
            point = point[None, :]
        if self.fkine is not None:
            s = self._rbf_fused.score(self.fkine, self.rbf_kernel, self.support_fkine, self.rbf_nodes, point)
        else:
            s = self._rbf_fused.score(None, self.rbf_kernel, self.support_points, self.rbf_nodes, point)
        if len(point) == 1 and isinstance(self.rbf_kernel, (kernel.RQKernel, kernel.MultiQuadratic)):
            s = s.reshape(-1)
        return s

    def poly_score(self, point):
        if point.ndim == 1:
            point = point[None, :]
        point = point.to(device=self.rbf_nodes.device, dtype=self.rbf_nodes.dtype)
        return self.rbf_score(point)


class MultiDiffCo(DiffCo):
    """One perceptron per label column on a shared kernel matrix; y, gains, hypothesis, rbf_nodes are [N, C]."""

    def __init__(self, objects=None, kernel_func='rq', gamma=1, beta=1, gt_checker=None):
        super().__init__(objects, kernel_func, gamma, beta, gt_checker)
        self.objects = objects
        self.num_class = None

    def initialize(self, X, y, gains=None, hypothesis=None, kernel_matrix=None):
        self._reset_caches()
        self.support_points = X.clone()
        self.y = y.clone()
        n = len(X)
        self.num_class = y.shape[1]
        given = [t is not None for t in (gains, hypothesis, kernel_matrix)]
        if not any(given):
            self.gains = torch.zeros((n, self.num_class), dtype=X.dtype)
            self.hypothesis = torch.zeros((n, self.num_class), dtype=X.dtype)
            self.kernel_matrix = None  # created where the trainer runs
        elif not all(given):
            raise ValueError('DiffCo: you passed in some existing parameters but not all three of gains, '
                             'hypothesis, and kernel_matrix')
        else:
            self.gains, self.hypothesis, self.kernel_matrix = gains, hypothesis, kernel_matrix

    def train_perceptron(self, X, y, max_iteration=1000, gains=None, hypothesis=None, kernel_matrix=None):
        self.initialize(X, y, gains=gains, hypothesis=hypothesis, kernel_matrix=kernel_matrix)
        print('MultiDiffCo training...')
        it = self._run_trainer(max_iteration, cold=(gains is None))
        print('Ended at iteration {}'.format(it))
        print('ACC: {}'.format(torch.sum((self.hypothesis > 0) == (self.y > 0)) / float(self.y.numel())))

    def train(self, X, y, max_iteration=1000, gains=None, hypothesis=None, method='original', distance=None,
              kernel_matrix=None):
        if method != 'original':
            raise NotImplementedError(f"training method {method!r}")
        self.train_method = method
        self.distance = distance
        t0 = time()
        self.train_perceptron(X, y, max_iteration, gains, hypothesis, kernel_matrix)
        self.filter_support_points_(torch.sum(self.gains != 0, dim=1) != 0)  # drop rows unused by every class
        print('{} training done. {:.4f} secs cost'.format(method, time() - t0))

    def predict(self, point):
        return (self.score(point) > 0) * 2 - 1

    def fit_poly(self, kernel_func=None, target='hypo', fkine=None, reg=0):
        X, t = self._fit_inputs(kernel_func, target, fkine)
        kmat = self.rbf_kernel(X, X)
        # decouple, per class, the supports that class uses from those it does not
        for c in range(self.num_class):
            used = self.gains[:, c] != 0
            cross = used[:, None] & (~used)[None, :]
            kmat[cross] = 0
            kmat[cross.T] = 0
        eye = torch.eye(len(kmat), dtype=kmat.dtype, device=kmat.device)
        self.rbf_nodes = solve_system(self.rbf_kernel, kmat + reg * eye, t.to(kmat.dtype))
        self.rbf_nodes[self.gains == 0] = 0
        assert self.rbf_nodes.shape == (len(self.support_points), self.num_class)


class DiffCoBeta(DiffCo):
    """Distance-regression variant; only its inference call is on the hot path.  The reference's
    training routine (deprecated/DiffCoBeta.py:23-60) relies on the removed `torch.solve`; here it is
    perceptron on sign(d) followed by rbf_nodes = solve(K_rbf + 0.1 I, d) over the supports plus the
    last `n_left_out_points` samples, as that routine describes."""

    def __init__(self, obstacles=None, kernel_func='rq', rbf_kernel=None, gamma=1, beta=1, k=1, epsilon=1,
                 gt_checker=None):
        super().__init__(obstacles, kernel_func, gamma, beta, gt_checker)
        self.rbf_kernel = kernel.Polyharmonic(k=1, epsilon=1) if rbf_kernel is None else rbf_kernel

    def train(self, X, d, fkine=None, max_iteration=1000, n_left_out_points=100, dtol=1e-4, keep_all=False):
        t0 = time()
        self.n_left_out_points = n_left_out_points
        self.distance = d[:-n_left_out_points]
        self.train_perceptron(X[:-n_left_out_points], (d[:-n_left_out_points] >= 0) * 2. - 1,
                              max_iteration=max_iteration)
        if not keep_all:
            self.filter_support_points_(self.gains != 0)
            print('Number of gains = ', len(self.gains))
        self.num_origin_supports = len(self.gains)
        Xa = torch.cat([self.support_points, X[-n_left_out_points:]], dim=0)
        da = torch.cat([self.distance, d[-n_left_out_points:]], dim=0)
        self.support_points, self.distance = Xa, da
        feats = Xa
        if fkine is not None:
            feats = fkine(Xa).reshape([len(Xa), -1]).detach()
            self.fkine, self.support_fkine = fkine, feats
        self.kernel_matrix = self.rbf_kernel(feats, feats)
        self.kernel_matrix = self.kernel_matrix + 0.1 * torch.eye(len(feats), dtype=self.kernel_matrix.dtype)
        self.gains = solve_system(self.rbf_kernel, self.kernel_matrix, da.reshape(-1, 1).to(self.kernel_matrix.dtype)).reshape(-1)
        self.rbf_nodes = self.gains
        self._reset_caches()
        self.hypothesis = self.rbf_score(self.support_points)  # [N, 1], without the 0.1 I (deprecated/DiffCoBeta.py:109-110)
        print('DiffCo training done. {:.4f} secs cost'.format(time() - t0))
