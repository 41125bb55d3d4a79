// solve_kernels.hip — the S x S system of fit_poly solved in ONE launch (round 4; dcx_solve).
//
// fit_poly (reference kernel_perceptrons.py:271-283, deprecated/MultiDiffCo.py:125-151) ends in
// torch.linalg.solve(K(supports, supports), targets): a few hundred to a few thousand unknowns, once per active-learning
// round.  Through the library route (hipSOLVER getrf + getrs) that is a few thousand tiny launches: 1.8 ms of GPU time at
// S = 438 in a tight loop, 4.6 ms inside the facade with OMP_NUM_THREADS=8 and 12 ms on a 128-thread host - launch-bound, and
// the largest single item of an update round (tools/facade_lines.py).  Here: LU with partial pivoting, right-looking and
// blocked, one cooperative launch (solve_body.h, compiled for 256 and for 512 threads per workgroup).
//
//   * the working copy W is COLUMN-major fp64, [n + nrhs columns][n rows]: the right-hand sides are extra columns, so the
//     forward elimination needs no separate pass, and every sweep over rows is coalesced.  fp64 because the reference
//     solves in fp32 LAPACK and a polyharmonic kernel matrix (zero diagonal) is not well conditioned: the rounding of this
//     factorisation stays far below the reference's own (3e-8 of the fp64 referee against 1e-5 .. 1e-3);
//   * per block step: workgroup 0 factorises the panel (rows k0..n x nb columns) in REGISTERS, 64 doubles per thread - nb
//     adapts to the rows left (32 columns while they fit, then 16, 8, 4).  Rows never move: a swap exchanges two position
//     labels.  Two __syncthreads per column: a DPP wave reduction and one LDS atomic find the pivot, its owner publishes the
//     row through LDS, everybody updates its own rows.  Besides L and U the panel leaves the NET effect of its swaps: which
//     original row ends in each of the nb top rows, and the (at most nb) displaced rows below them;
//   * every workgroup takes groups of 8 (16) trailing columns: gathers the rows the swaps moved (reads, barrier, writes: no
//     sequential chain of swaps), forward-substitutes the nb x 8 block in registers (one lane per element, the solved row
//     broadcast by v_readlane), and applies the rank-nb update to its rows with L's row in registers and U's block in LDS;
//   * LOOK-AHEAD: the groups that hold the next panel's columns are updated first, one workgroup each; workgroup 0 waits for
//     exactly those (a counter, not a barrier) and factorises the next panel while the others are still updating the rest.
//     One grid barrier per block step; the panel's dependent chain runs beside the trailing update, not in front of it;
//   * after the last panel, workgroup 0 back-substitutes blockwise and writes X in fp32.
// The barriers are the trainer's (arrival counter, agent-scope release/acquire, a time-out that flags instead of hanging).
// If the cooperative launch is refused (or the stream is being captured) the same kernel runs as ONE workgroup.
// Measured (profiles/r04_solve.txt): n = 438: 0.81 ms (hipSOLVER through torch 1.77), 1000: 2.7 (4.8), 2000: 9.1 (15 - 18),
// 3000: 24.5 (25 - 34), 4000: 52 (44): the Python side hands systems beyond 3072 unknowns to the library.
#include "solve_kernels.h"
#include <utility>

namespace dcx {
namespace {

#define DCX_PIN8(x, o) asm volatile("" : "+v"(x[o]), "+v"(x[o + 1]), "+v"(x[o + 2]), "+v"(x[o + 3]), "+v"(x[o + 4]), \
                                    "+v"(x[o + 5]), "+v"(x[o + 6]), "+v"(x[o + 7]) : : "memory")
// developer builds (-DDCX_SOLVE_TS, tools/solve_probe.hip): workgroup 0 stamps the phases of every block step
#ifdef DCX_SOLVE_TS
__device__ unsigned long long g_solve_ts[8 * 2048];
#define DCX_STS(slot) do { if (blockIdx.x == 0 && tid == 0 && (slot) < 8 * 2048) g_solve_ts[slot] = wall_clock64(); } while (0)
// panel columns: durations summed per phase (the last eight slots: search + exchange, publish, read + update, panel load)
#define DCX_PTS_DECL unsigned long long pts_ = wall_clock64()
#define DCX_PTS(k) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); g_solve_ts[8 * 2047 + (k)] += n_ - pts_; pts_ = n_; } } while (0)
#else
#define DCX_STS(slot) do { } while (0)
#define DCX_PTS_DECL do { } while (0)
#define DCX_PTS(k) do { } while (0)
#endif

// Two workgroup sizes.  256 threads (one wave per SIMD) has the shorter barriers and is the faster form up to ~730 unknowns;
// 512 threads hold a panel twice as wide in their registers (32 columns up to 1024 rows, 16 up to 2048, 8 up to 4096): half
// the block steps - grid barriers and passes over the trailing matrix - for the larger systems (n = 2000: 19.1 -> 10.7 ms
// before the look-ahead).
#define DCX_SOLVE_NT 256
namespace nt256 {
#include "solve_body.h"
}
#undef DCX_SOLVE_NT
#define DCX_SOLVE_NT 512
namespace nt512 {
#include "solve_body.h"
}
#undef DCX_SOLVE_NT

}  // namespace

size_t solve_work_bytes(int64_t n, int64_t nrhs) {
    return sizeof(SolveSync) + sizeof(double) * (size_t)(n + nrhs) * (size_t)n;
}

hipError_t launch_solve(const float* A, const float* B, float* X, int n, int nrhs, void* work, int32_t* info, int n_cu,
                        bool one_workgroup, int threads, hipStream_t st) {
    static_assert(sizeof(SolveSync) % 16 == 0, "the fp64 copy starts behind the sync block");
    SolveArgs a;
    a.A = A; a.B = B; a.X = X; a.info = info;
    a.gs = reinterpret_cast<SolveSync*>(work);
    a.W = reinterpret_cast<double*>(reinterpret_cast<char*>(work) + sizeof(SolveSync));
    a.n = n; a.nrhs = nrhs; a.ld = n;
    hipError_t e = hipMemsetAsync(work, 0, sizeof(SolveSync), st);
    if (e != hipSuccess) return e;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusActive;
    }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (threads == 0) threads = n <= kSolveSmallN ? 256 : 512;
    return threads == 256 ? nt256::launch(a, n_cu, one_workgroup, capturing, st) : nt512::launch(a, n_cu, one_workgroup, capturing, st);
}

}  // namespace dcx
