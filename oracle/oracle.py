"""ctypes wrapper of liboracle.so — the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
numpy in, numpy out; `dtype=np.float32` runs the fp32 restatement, `np.float64` the referee.
The FK description is any ctypes object laid out like `dcx_fk_desc` (include/dcx.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("dcx_oracle.c", "dcx_oracle_impl.h")] + [
        os.path.join(_HERE, "..", "include", "dcx.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _sfx(dtype):
    return "_f32" if np.dtype(dtype) == np.float32 else "_f64"


def _arr(a, dtype):
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def max_threads():
    return lib().orc_max_threads()


def set_threads(n):
    lib().orc_set_threads(C.c_int(n))


def fkine(desc, q, dtype=np.float32):
    q = _arr(q, dtype).reshape(-1, desc.dof)
    X = np.empty((len(q), desc.n_points * desc.point_dim), dtype=dtype)
    getattr(lib(), "orc_fkine" + _sfx(dtype))(C.byref(desc), _p(q), C.c_int64(len(q)), _p(X))
    return X.reshape(len(q), *getattr(desc, "feature_shape", (desc.n_points, desc.point_dim)))


def fkine_vjp(desc, q, gX, dtype=np.float32):
    q = _arr(q, dtype).reshape(-1, desc.dof)
    gX = _arr(gX, dtype).reshape(len(q), -1)
    gq = np.empty_like(q)
    getattr(lib(), "orc_fkine_vjp" + _sfx(dtype))(C.byref(desc), _p(q), _p(gX), C.c_int64(len(q)), _p(gq))
    return gq


def kernel_matrix(kind, p0, p1, x, s, dtype=np.float32):
    x, s = _arr(x, dtype), _arr(s, dtype)
    x, s = x.reshape(len(x), -1), s.reshape(len(s), -1)
    kp = _arr([p0, p1], dtype)
    K = np.empty((len(x), len(s)), dtype=dtype)
    getattr(lib(), "orc_kernel_matrix" + _sfx(dtype))(C.c_int(kind), _p(kp), _p(x), C.c_int64(len(x)), _p(s),
                                                       C.c_int64(len(s)), C.c_int(x.shape[1]), _p(K))
    return K


def score_grad(desc, kind, p0, p1, sup, W, q, upstream=None, want_jac=False, dtype=np.float32):
    """-> (score [B, C], grad [B, dof], jac [B, C, dof] or None)"""
    sup = _arr(sup, dtype)
    sup = sup.reshape(len(sup), -1)
    W = _arr(W, dtype)
    W = W.reshape(len(sup), -1)
    q = _arr(q, dtype).reshape(-1, desc.dof)
    S, D, Cn, B = len(sup), sup.shape[1], W.shape[1], len(q)
    assert D == desc.n_points * desc.point_dim
    kp = _arr([p0, p1], dtype)
    up = None if upstream is None else _arr(upstream, dtype).reshape(B, Cn)
    score = np.empty((B, Cn), dtype=dtype)
    grad = np.empty((B, desc.dof), dtype=dtype)
    jac = np.empty((B, Cn, desc.dof), dtype=dtype) if want_jac else None
    getattr(lib(), "orc_score_grad" + _sfx(dtype))(C.byref(desc), C.c_int(kind), _p(kp), _p(sup), _p(W), C.c_int64(S),
                                                    C.c_int(D), C.c_int(Cn), _p(q), C.c_int64(B), _p(up), _p(score),
                                                    _p(grad), _p(jac))
    return score, grad, jac


def solve(kmat, rhs, dtype=np.float64):
    """x with kmat x = rhs: the S x S system fit_poly ends in (reference kernel_perceptrons.py:283, deprecated/DiffCo.py:162,
    deprecated/MultiDiffCo.py:151: torch.linalg.solve = LAPACK gesv, LU with partial pivoting).  numpy's solve is the same
    LAPACK routine; `dtype=np.float32` is the reference's arithmetic, float64 the referee."""
    a = np.ascontiguousarray(kmat, dtype=dtype)
    b = np.ascontiguousarray(rhs, dtype=dtype)
    return np.linalg.solve(a, b.reshape(len(a), -1)).reshape(b.shape)
