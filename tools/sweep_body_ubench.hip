// tools/sweep_body_ubench.hip — developer tool (GPU box): what does the sweep's pair body cost by itself?  The product's own
// sweep_rows<> (score_kernel.h) in a bare kernel: no FK, no fold, no J^T, every wave sweeps the same few rows (scalar-cache
// hits), W waves per SIMD on every SIMD of the chip.  Reports shader cycles (clock64: the clock the SIMD really ran at) and
// nanoseconds per wave-row per SIMD, for the one-class Polyharmonic(1) sweep of the headline and the five-class RQ2 sweep of
// config #3, both in the expanded form.  The fused kernel's figure for the same quantity = launch time x SIMDs / wave-rows.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idiffco_amd/csrc tools/sweep_body_ubench.hip -o devlibs/sweep_body_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "score_kernel.h"

using namespace dcx;

template <int D, int KF, int CC, int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void body_kernel(const ScoreArgs a, int rows_n, int reps, float* out, unsigned long long* ts) {
    float x[D], up[CC], sc[CC], gx[D];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < D; ++k) { x[k] = 0.01f * (float)(lane + k) - 0.3f; gx[k] = 0.f; }
#pragma unroll
    for (int c = 0; c < CC; ++c) { up[c] = 1.f; sc[c] = 0.f; }
#ifdef UBENCH_SKEW   // do waves that do NOT run in lockstep get a different clock?  every wave starts up to 15 x UBENCH_SKEW x 64 cycles late
    for (int i = 0; i < (int)((blockIdx.x * 7u + (threadIdx.x >> 6) * 3u) & 15u); ++i) __builtin_amdgcn_s_sleep(UBENCH_SKEW);
#endif
    const unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) sweep_rows<D, KF, CC, MODE, true>(a, x, up, 0, rows_n, sc, gx);
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) s += gx[k];
#pragma unroll
    for (int c = 0; c < CC; ++c) s += sc[c];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ts[blockIdx.x] = t1 - t0;
}

template <int D, int KF, int CC, int WPS>
void run(const char* name, int cus) {
    using L = RowLayout<D, CC>;
    const int rows_n = 96, reps = 400;
    std::vector<float> rows((size_t)(rows_n + 16) * L::RS, 0.f);
    for (int j = 0; j < rows_n + 16; ++j) {
        float ss = 0.f;
        for (int k = 0; k < D; ++k) { const float v = 0.02f * (float)((j * 7 + k * 3) % 23) - 0.2f; rows[(size_t)j * L::RS + k] = v; ss += v * v; }
        for (int c = 0; c < CC; ++c) rows[(size_t)j * L::RS + L::W_OFF + c] = 0.01f * (float)((j + c) % 5 - 2);
        if (CC > 1) rows[(size_t)j * L::RS + L::WSUM_OFF] = 0.01f;
        rows[(size_t)j * L::RS + L::SS_OFF] = ss + (KF == KF_RQ2 ? 0.2f : 0.f);
    }
    float *d_rows, *d_out;
    unsigned long long* d_ts;
    const int blocks = cus * WPS;   // 256 threads = one wave per SIMD per block
    hipMalloc(&d_rows, rows.size() * sizeof(float));
    hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(float));
    hipMalloc(&d_ts, (size_t)blocks * sizeof(unsigned long long));
    hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice);
    ScoreArgs a{};
    a.rows = d_rows;
    a.kp0 = 0.2f;
    a.kp1 = 2.0f;
    a.S = rows_n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto kern = body_kernel<D, KF, CC, MODE_GRAD_ROW, WPS>;
    for (int warm = 0; warm < 3; ++warm) kern<<<blocks, 256>>>(a, rows_n, reps, d_out, d_ts);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(a, rows_n, reps, d_out, d_ts);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> ts(blocks);
    hipMemcpy(ts.data(), d_ts, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(ts.begin(), ts.end());
    const double wave_rows_per_simd = (double)WPS * rows_n * reps;
    const double ns = ms * 1e6 / wave_rows_per_simd;
    const double cyc = (double)ts[blocks / 2] / ((double)rows_n * reps) / WPS;   // a wave's own span / its rows / waves sharing the SIMD
    printf("%-34s waves/SIMD=%d  %8.3f ms   %6.2f ns = %6.1f shader cycles per wave-row per SIMD  (clock %.2f GHz; a wave alone: %.0f cycles per row)\n",
           name, WPS, ms, ns, cyc, cyc / ns, (double)ts[blocks / 2] / ((double)rows_n * reps));
    hipFree(d_rows); hipFree(d_out); hipFree(d_ts);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs=%d\n", prop.gcnArchName, cus);
    run<12, KF_POLY1, 1, 1>("headline: D=12 C=1 POLY1 XF", cus);
    run<12, KF_POLY1, 1, 2>("headline: D=12 C=1 POLY1 XF", cus);
    run<12, KF_POLY1, 1, 4>("headline: D=12 C=1 POLY1 XF", cus);
    run<12, KF_POLY1, 1, 8>("headline: D=12 C=1 POLY1 XF", cus);
    run<12, KF_RQ2, 5, 1>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 5, 2>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 5, 4>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 5, 6>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 5, 8>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 1, 8>("headline_rq: D=12 C=1 RQ2 XF", cus);
    run<12, KF_POLY1, 5, 8>("config #3 poly: D=12 C=5 POLY1 XF", cus);
    return 0;
}
