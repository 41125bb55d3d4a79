// solve_kernels.h — the dense solve behind dcx_solve (solve_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcx {

// head of the caller's workspace: the grid barrier's words, LAPACK's info, and what a panel's row swaps amount to
struct SolveSync {
    unsigned int counter, abort;
    int info;          // 0, or 1 + the first column whose pivot is exactly zero
    unsigned int panel_done;   // look-ahead: groups of the next panel's columns updated so far (monotonic)
    // the net row movement of a panel, double-buffered by panel parity (the next panel is factorised while the other
    // workgroups still apply this one's):
    int n_low[2];      // rows below the block that received another row's content
    int pad2[2];
    int top_src[2][32];   // the row that ends in top row c of the block
    int low_dst[2][32], low_src[2][32];
};

struct SolveArgs {
    const float* A;    // [n, n] row-major
    const float* B;    // [n, nrhs] row-major
    float* X;          // [n, nrhs] row-major
    double* W;         // [n + nrhs][ld] column-major working copy
    SolveSync* gs;
    int32_t* info;     // [2]: LAPACK info (or -1: a grid barrier gave up), barriers passed
    int n, nrhs, ld;
};

constexpr int kSolveMaxN = 4096;    // 12 bits of a pivot key carry the row (and 512 threads x 8 rows x 8 columns hold the widest panel)
constexpr int kSolveSmallN = 736;   // up to here the 256-thread form is the faster one (tools/solve_latency.py)
size_t solve_work_bytes(int64_t n, int64_t nrhs);
// threads: 256, 512, or 0 = by size
hipError_t launch_solve(const float* A, const float* B, float* X, int n, int nrhs, void* work, int32_t* info, int n_cu,
                        bool one_workgroup, int threads, hipStream_t st);

}  // namespace dcx
