"""dcx_solve (csrc/solve_kernels.hip): the S x S system of fit_poly in one launch, against LAPACK (oracle.solve) on the
same fp32 inputs.  The kernel factorises in fp64, so its answer must sit at fp32 rounding of the fp64 referee - closer than
the reference's own fp32 LAPACK solve does."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import KIND, relerr

pytestmark = pytest.mark.gpu


def _solve(a, b, flags=0):
    """straight through the C ABI"""
    from diffco_amd import _lib
    lib = _lib.require_gpu()
    dev = torch.device("cuda", 0)
    A = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    B = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(dev)
    n, r = B.shape
    nbytes = int(lib.dcx_solve_work_bytes(n, r))
    work = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.float64)
    X = torch.full((n, r), float("nan"), device=dev)
    info = torch.full((2,), 77, device=dev, dtype=torch.int32)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.dcx_solve(0, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), n, r, C.c_void_p(X.data_ptr()),
                             C.c_void_p(work.data_ptr()), nbytes, C.c_void_p(info.data_ptr()), flags, st))
    torch.cuda.synchronize()
    return X.cpu().numpy(), info.cpu().numpy()


@pytest.fixture(params=[0, 256, 512], ids=["by-size", "256-threads", "512-threads"])
def threads(request):
    """dcx_solve's two workgroup sizes (the kernel body is compiled for each), and the rule that picks one"""
    from diffco_amd import _lib
    lib = _lib.require_gpu()
    lib.dcx_debug_set(b"solve_threads", request.param if request.param else -1)
    yield request.param
    lib.dcx_debug_set(b"solve_threads", -1)


# (the right-hand-side tiling - r >= 9 - is covered at the small sizes: those combinations are not generated beyond n = 600)
@pytest.mark.parametrize("n,r", [(n, r) for r in (1, 5, 9, 17) for n in (1, 2, 3, 7, 31, 32, 33, 64, 65, 100, 257, 438, 512, 513, 777, 1025, 2049)
                                 if not (r >= 9 and n > 600)])
def test_random_systems(n, r, threads):
    from oracle import oracle
    g = np.random.default_rng(1000 * n + r)
    a = g.standard_normal((n, n)).astype(np.float32)
    b = g.standard_normal((n, r)).astype(np.float32)
    x, info = _solve(a, b)
    assert info[0] == 0
    ref = oracle.solve(a, b)
    # (x is the fp64 solution rounded once; LAPACK in fp32 - the reference's arithmetic - is 1e2-1e4 times further away)
    assert relerr(x, ref) < 2e-7, relerr(x, ref)
    assert relerr(x, ref) <= relerr(oracle.solve(a, b, np.float32), ref) + 1e-7


@pytest.mark.parametrize("kind,p0,p1", [("poly", 1, 1.0), ("poly", 3, 2.0), ("mq", 1.0, 0.0), ("rq", 10.0, 2.0)])
@pytest.mark.parametrize("n", [50, 438, 1100])
def test_kernel_matrices(kind, p0, p1, n, threads):
    """what fit_poly solves: symmetric, zero diagonal for the polyharmonic kernels (every pivot needs a row swap)"""
    from oracle import oracle
    g = np.random.default_rng(n)
    pts = g.uniform(-1, 1, (n, 24)).astype(np.float32)
    a = oracle.kernel_matrix(KIND[kind], p0, p1, pts, pts)
    b = np.sign(g.standard_normal((n, 1))).astype(np.float32)
    x, info = _solve(a, b)
    assert info[0] == 0
    ref = oracle.solve(a, b)
    assert relerr(x, ref) < 2e-7
    # the interpolation property fit_poly is after
    assert np.abs(a.astype(np.float64) @ x.astype(np.float64) - b).max() < 5e-4 * max(1.0, np.abs(x).max())


def test_permutations_and_repeated_pivot_rows(threads):
    """row movement is applied as ONE gather per block: swaps that chain through the same rows (a cyclic shift: every
    pivot comes from the row below, every displaced row moves again) must still land where LAPACK's sequential swaps do"""
    from oracle import oracle
    g = np.random.default_rng(5)
    for n in (5, 40, 97, 300):
        d = g.uniform(1.0, 2.0, n).astype(np.float32)
        for perm in (np.roll(np.arange(n), 1), np.roll(np.arange(n), -1), np.arange(n)[::-1], g.permutation(n)):
            a = np.zeros((n, n), np.float32)
            a[np.arange(n), perm] = d
            a += (1e-3 * g.standard_normal((n, n))).astype(np.float32)
            b = g.standard_normal((n, 2)).astype(np.float32)
            x, info = _solve(a, b)
            assert info[0] == 0 and relerr(x, oracle.solve(a, b)) < 2e-7


def test_one_workgroup_form_is_the_same_arithmetic(threads):
    g = np.random.default_rng(9)
    for n in (33, 438, 900):
        a = g.standard_normal((n, n)).astype(np.float32)
        b = g.standard_normal((n, 3)).astype(np.float32)
        x0, i0 = _solve(a, b)
        x1, i1 = _solve(a, b, flags=1)
        assert i0[0] == 0 and i1[0] == 0 and i1[1] == 0 and i0[1] > 0   # (info[1]: grid barriers passed)
        np.testing.assert_array_equal(x0, x1)


def test_both_workgroup_sizes_are_the_same_arithmetic():
    """256 and 512 threads block the elimination differently (panel widths 32 / 16 / 8 by the rows left) but every element
    sees the same fused multiply-adds in the same order: same pivots, bit-identical solutions"""
    from diffco_amd import _lib
    lib = _lib.require_gpu()
    g = np.random.default_rng(21)
    try:
        for n in (100, 600, 1300):
            a = g.standard_normal((n, n)).astype(np.float32)
            b = g.standard_normal((n, 2)).astype(np.float32)
            lib.dcx_debug_set(b"solve_threads", 256)
            x0, i0 = _solve(a, b)
            lib.dcx_debug_set(b"solve_threads", 512)
            x1, i1 = _solve(a, b)
            assert i0[0] == 0 and i1[0] == 0 and (n < 1025 or i1[1] < i0[1])   # (fewer block steps with the wider panels)
            np.testing.assert_array_equal(x0, x1)
    finally:
        lib.dcx_debug_set(b"solve_threads", -1)


def test_singular_matrix_is_reported_not_solved():
    from diffco_amd import _ops
    a = np.ones((40, 40), np.float32)
    x, info = _solve(a, np.ones((40, 1), np.float32))
    assert info[0] == 2       # LAPACK's info: the second pivot is exactly zero
    z, info = _solve(np.zeros((3, 3), np.float32), np.ones((3, 1), np.float32))
    assert info[0] == 1
    with pytest.raises(torch.linalg.LinAlgError):
        _ops.solve(torch.ones(40, 40), torch.ones(40, 1))


def test_argument_checks():
    from diffco_amd import _lib
    lib = _lib.require_gpu()
    t = torch.zeros(64, device="cuda")
    p = C.c_void_p(t.data_ptr())
    assert lib.dcx_solve(0, p, p, 4097, 1, p, p, 1 << 30, p, 0, None) != 0
    assert lib.dcx_solve(0, p, p, 4, 65, p, p, 1 << 30, p, 0, None) != 0
    assert lib.dcx_solve(0, p, p, 4, 1, p, p, 16, p, 0, None) != 0       # workspace too small
    assert lib.dcx_solve(0, None, p, 4, 1, p, p, 1 << 30, p, 0, None) != 0
    assert lib.dcx_solve_work_bytes(4, 1) >= 8 * 4 * 5 and lib.dcx_solve_work_bytes(0, 1) == 0


def test_ops_solve_and_fit_nodes_keep_device_dtype_and_shape():
    from diffco_amd import _ops
    from oracle import oracle
    g = np.random.default_rng(3)
    n = 120
    pts = g.uniform(-1, 1, (n, 8, 3)).astype(np.float32)
    t = g.standard_normal(n).astype(np.float32)
    a = oracle.kernel_matrix(KIND["poly"], 1, 1.0, pts.reshape(n, -1), pts.reshape(n, -1))
    ref = oracle.solve(a, t)
    x = _ops.solve(torch.from_numpy(a).double(), torch.from_numpy(t))       # CPU fp64 in -> CPU fp64 out, 1-d stays 1-d
    assert x.device.type == "cpu" and x.dtype == torch.float64 and x.shape == (n,)
    assert relerr(x.numpy(), ref) < 2e-7
    # a float64 system keeps its precision (it used to be cast to fp32 on the way in - a `reg` of 1e-9 on the diagonal of a
    # float64 matrix vanished): the library's LU in float64
    a64 = a.astype(np.float64) + 1e-9 * np.eye(n)
    x64 = _ops.solve(torch.from_numpy(a64), torch.from_numpy(t.astype(np.float64)))
    assert x64.dtype == torch.float64 and relerr(x64.numpy(), oracle.solve(a64, t.astype(np.float64))) < 1e-9
    nodes = _ops.fit_nodes(KIND["poly"], 1, 1.0, torch.from_numpy(pts), torch.from_numpy(t)[:, None])
    assert nodes.shape == (n, 1) and nodes.device.type == "cpu" and relerr(nodes.numpy()[:, 0], ref) < 5e-5
    # reg goes on the diagonal (deprecated/MultiDiffCo.py:151)
    nodes = _ops.fit_nodes(KIND["poly"], 1, 1.0, torch.from_numpy(pts).cuda(), torch.from_numpy(t).cuda(), reg=0.1)
    assert nodes.is_cuda and relerr(nodes.cpu().numpy(), oracle.solve(a + 0.1 * np.eye(n), t)) < 5e-5
    # beyond the kernel's sizes: the library route, same interface
    big = torch.randn(70, 70)
    rhs = torch.randn(70, 65)
    assert relerr(_ops.solve(big, rhs).numpy(), oracle.solve(big.numpy(), rhs.numpy())) < 1e-3


def test_solve_inside_a_stream_capture():
    """no allocation, no synchronisation, no cooperative launch under capture: the one-workgroup form is recorded"""
    from diffco_amd import _lib
    lib = _lib.require_gpu()
    from oracle import oracle
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(11)
    n = 90
    a = g.standard_normal((n, n)).astype(np.float32)
    b = g.standard_normal((n, 1)).astype(np.float32)
    A, B = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    nbytes = int(lib.dcx_solve_work_bytes(n, 1))
    work = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.float64)
    X = torch.zeros((n, 1), device=dev)
    info = torch.zeros(2, device=dev, dtype=torch.int32)
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.dcx_solve(0, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), n, 1, C.c_void_p(X.data_ptr()),
                                     C.c_void_p(work.data_ptr()), nbytes, C.c_void_p(info.data_ptr()), 0, st))
    graph.replay()
    torch.cuda.synchronize()
    assert int(info[0]) == 0 and relerr(X.cpu().numpy(), oracle.solve(a, b)) < 2e-7
    B.mul_(2.0)
    graph.replay()
    torch.cuda.synchronize()
    assert relerr(X.cpu().numpy(), oracle.solve(a, 2 * b)) < 2e-7


def test_solves_on_two_streams_at_once():
    """two cooperative launches from two streams (each its own workspace): both complete, both right - a solve that could not
    get its workgroups resident in time reports -1 and the wrapper runs it again as one workgroup"""
    from diffco_amd import _ops
    from oracle import oracle
    g = np.random.default_rng(77)
    mats = [(g.standard_normal((n, n)).astype(np.float32), g.standard_normal((n, 1)).astype(np.float32)) for n in (500, 900)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for rep in range(3):
        for i, (a, b) in enumerate(mats):
            A, Bm = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
            torch.cuda.synchronize()
            with torch.cuda.stream(streams[i]):
                outs[i] = _ops.solve(A, Bm)
        torch.cuda.synchronize()
        for (a, b), x in zip(mats, outs):
            assert relerr(x.cpu().numpy(), oracle.solve(a, b)) < 2e-7
