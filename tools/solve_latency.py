"""dcx_solve against torch.linalg.solve (hipSOLVER) on the GPU box: microseconds per solve, one right-hand side.
Usage: python tools/solve_latency.py [n ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from diffco_amd import _lib  # noqa: E402


def main():
    lib = _lib.require_gpu()
    dev = torch.device("cuda", 0)
    sizes = [int(v) for v in sys.argv[1:]] or [100, 200, 438, 700, 1000, 1500, 2000, 3000, 4000]
    for n in sizes:
        g = torch.Generator().manual_seed(n)
        pts = torch.rand((n, 24), generator=g).to(dev)
        A = torch.cdist(pts, pts).contiguous()          # a polyharmonic(1) matrix: zero diagonal
        B = torch.randn((n, 1), generator=g).to(dev)
        nbytes = int(lib.dcx_solve_work_bytes(n, 1))
        work = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.float64)
        X = torch.empty((n, 1), device=dev)
        info = torch.zeros(2, device=dev, dtype=torch.int32)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def ours(flags=0):
            _lib.check(lib.dcx_solve(0, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), n, 1, C.c_void_p(X.data_ptr()),
                                     C.c_void_p(work.data_ptr()), nbytes, C.c_void_p(info.data_ptr()), flags, st))

        def timed(f, reps):
            f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6
        reps = 20 if n <= 1000 else 5
        t_ours = timed(ours, reps)
        bars = int(info[1])
        t_one = timed(lambda: ours(1), max(2, reps // 4)) if n <= 1000 else float("nan")
        t_lib = timed(lambda: torch.linalg.solve(A, B), max(2, reps // 4))
        ref = torch.linalg.solve(A.double(), B.double())
        err = float((X.double() - ref).abs().max() / ref.abs().max())
        print(f"n={n:5d}  dcx_solve {t_ours:9.1f} us ({bars} grid barriers)   one workgroup {t_one:9.1f} us   "
              f"torch.linalg.solve {t_lib:9.1f} us   err vs fp64 {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
