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

// ---- experiment (round 6): the EXPANDED form with two rows per packed instruction -----------------------------
// The product's expanded body packs a row's features two by two (6 v_pk_fma for the distance, 6 for the gradient at D = 12, an add of
// the two halves, then a scalar chain: seed add, clamp, rsq, coefficient, score, sum of coefficients, near test = 20 VALU per row).
// Here the two rows of a stage ride in the two halves instead (rows pair-interleaved in memory, like the direct form's pair2 of
// config #4): 12 + 12 v_pk_fma per TWO rows, no add of halves, the chain's multiplies / adds packed (the clamp, the rsq and the near
// test stay one per row: no packed form) = 34 per two rows, 24 gradient accumulators instead of 12.  Polyharmonic(1), one class,
// gradient with the row weight; the near-pair correction block is left out (timing only: the data has no near pairs).
template <int D, int WPS>
__global__ __launch_bounds__(256, WPS) void xf2_kernel(const float* rows_il, int rows_n, int reps, float* out, unsigned long long* ts) {
    constexpr int RS = RowLayout<D, 1>::RS, W = RowLayout<D, 1>::W_OFF, SS = RowLayout<D, 1>::SS_OFF, R2 = 2 * RS;
    const int lane = threadIdx.x & 63;
    float x[D];
    float xx = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) { x[k] = 0.01f * (float)(lane + k) - 0.3f; xx = fmaf(x[k], x[k], xx); }
    const float thr = fmaxf(0.01f * xx, 1e-30f);
    v2f ga[D], sc2 = {0.f, 0.f}, as2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < D; ++k) ga[k] = v2f{0.f, 0.f};
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)rows_il;
    unsigned long long near_any = 0;
    auto load2 = [&](float (&dst)[R2], int j) __attribute__((always_inline)) {
        cfloat_ptr r = rows + (size_t)j * RS;
#pragma unroll
        for (int e = 0; e < R2; ++e) dst[e] = r[e];
    };
    auto body = [&](const float (&b)[R2]) __attribute__((always_inline)) {
        v2f acc = v2f{xx, xx} + v2f{b[2 * SS], b[2 * SS + 1]};
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const v2f xm = {-2.0f * x[k], -2.0f * x[k]};
            acc = __builtin_elementwise_fma(xm, v2f{b[2 * k], b[2 * k + 1]}, acc);
        }
        const v2f d2c = {fmaxf(acc.x, thr), fmaxf(acc.y, thr)};
        const v2f g = {__builtin_amdgcn_rsqf(d2c.x), __builtin_amdgcn_rsqf(d2c.y)};
        const v2f coef = g * v2f{b[2 * W], b[2 * W + 1]};
        sc2 = __builtin_elementwise_fma(coef, d2c, sc2);
#pragma unroll
        for (int k = 0; k < D; ++k) ga[k] = __builtin_elementwise_fma(coef, v2f{b[2 * k], b[2 * k + 1]}, ga[k]);
        as2 += coef;
        near_any |= __builtin_amdgcn_ballot_w64(fminf(d2c.x, d2c.y) <= thr);
    };
    const unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        float ab[R2], cd[R2];
        load2(ab, 0);
        int j = 0;
        for (; j + 3 < rows_n; j += 4) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load2(cd, j + 2);
            __builtin_amdgcn_sched_barrier(0);
            body(ab);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load2(ab, (j + 4 < rows_n) ? j + 4 : rows_n - 2);
            __builtin_amdgcn_sched_barrier(0);
            body(cd);
            __builtin_amdgcn_sched_barrier(0);
            if ((j & 60) == 0) {   // the flush of the expanded gradient, once per 64 rows: H <- H + (-2 x) (A / 2)
                const v2f ah = {0.5f * as2.x, 0.5f * as2.y};
#pragma unroll
                for (int k = 0; k < D; ++k) ga[k] = __builtin_elementwise_fma(v2f{-2.0f * x[k], -2.0f * x[k]}, ah, ga[k]);
                as2 = v2f{0.f, 0.f};
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = sc2.x + sc2.y + as2.x + as2.y + (near_any ? 1.f : 0.f);
#pragma unroll
    for (int k = 0; k < D; ++k) s += ga[k].x + ga[k].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ts[blockIdx.x] = t1 - t0;
}

template <int D, int WPS>
void run_xf2(const char* name, int cus) {
    using L = RowLayout<D, 1>;
    const int rows_n = 96, reps = 400;
    std::vector<float> rows((size_t)(rows_n + 16) * L::RS, 0.f);
    for (int j = 0; j < rows_n + 16; ++j) {     // the same rows as run<>, pair-interleaved
        float ss = 0.f;
        auto at = [&](int e) -> float& { return rows[(size_t)(j >> 1) * 2 * L::RS + 2 * e + (j & 1)]; };
        for (int k = 0; k < D; ++k) { const float v = 0.02f * (float)((j * 7 + k * 3) % 23) - 0.2f; at(k) = v; ss += v * v; }
        at(L::W_OFF) = 0.01f * (float)(j % 5 - 2);
        at(L::SS_OFF) = ss;
    }
    float *d_rows, *d_out;
    unsigned long long* d_ts;
    const int blocks = cus * WPS;
    hipMalloc(&d_rows, rows.size() * sizeof(float));
    hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(float));
    hipMalloc(&d_ts, (size_t)blocks * sizeof(unsigned long long));
    hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto kern = xf2_kernel<D, WPS>;
    for (int warm = 0; warm < 3; ++warm) kern<<<blocks, 256>>>(d_rows, rows_n, reps, d_out, d_ts);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(d_rows, rows_n, reps, d_out, d_ts);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> ts(blocks);
    hipMemcpy(ts.data(), d_ts, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(ts.begin(), ts.end());
    const double ns = ms * 1e6 / ((double)WPS * rows_n * reps);
    const double cyc = (double)ts[blocks / 2] / ((double)rows_n * reps) / WPS;
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)kern);
    printf("%-34s waves/SIMD=%d  %8.3f ms   %6.2f ns = %6.1f shader cycles per wave-row per SIMD  (clock %.2f GHz; %d VGPRs)\n", name, WPS, ms, ns, cyc,
           cyc / ns, fa.numRegs);
    hipFree(d_rows); hipFree(d_out); hipFree(d_ts);
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
    if (x2_applies(D, CC, KF)) {   // (a -DDCX_XF2=1 build: the product's expanded sweep of this shape reads pair-interleaved rows)
        std::vector<float> il(rows.size(), 0.f);
        for (int j = 0; j < rows_n + 16; ++j)
            for (int e = 0; e < L::RS; ++e) il[(size_t)(j >> 1) * 2 * L::RS + 2 * e + (j & 1)] = rows[(size_t)j * L::RS + e];
        rows.swap(il);
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
    run_xf2<12, 2>("EXPERIMENT xf2: D=12 two rows/instr", cus);
    run_xf2<12, 4>("EXPERIMENT xf2: D=12 two rows/instr", cus);
    run_xf2<12, 7>("EXPERIMENT xf2: D=12 two rows/instr", cus);
    run_xf2<12, 8>("EXPERIMENT xf2: D=12 two rows/instr", cus);
    run<12, KF_RQ2, 5, 1>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 5, 2>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 5, 4>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 5, 6>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 5, 8>("config #3: D=12 C=5 RQ2 XF", cus);
    run<12, KF_RQ2, 1, 8>("headline_rq: D=12 C=1 RQ2 XF", cus);
    run<12, KF_POLY1, 5, 8>("config #3 poly: D=12 C=5 POLY1 XF", cus);
    return 0;
}
