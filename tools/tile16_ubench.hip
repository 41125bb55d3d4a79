// tools/tile16_ubench.hip — developer microbenchmark (round 4): the sweep of the round-3 verdict's item 5, priced.
//
// The proposal: for small batches map a wave to 16 configurations x 4 support quarters, rows read from LDS instead of
// through the scalar cache, so that config #2's 4096 configurations are 256 single-block tiles and the cross-block hand-over
// (~2 us of its 11.3) disappears.  What that does to the SWEEP is measurable without building the whole kernel: the pair
// body stays the same 20 VALU instructions, but its 14 row operands become four ds_read_b128 per pair and lane (four
// distinct rows per wave instruction, each broadcast to 16 lanes), and every block first copies ALL the rows into LDS.
// Two kernels at config #2's shape (D = 12, expanded form, Polyharmonic(1), S = 1000, B = 4096, 256 blocks of 8 waves):
//   S: today's layout - 64 configurations per block, a quarter of the rows per block (ys = 4), row operands uniform
//      (s_load_dwordx8 / x4 through the scalar cache), waves interleave the block's rows;
//   L: the proposal   - 16 configurations per block, all rows staged into LDS, lane = (configuration, quarter), the 8 waves
//      x 4 quarters sweep 32 slices of the rows.
// Both end with their partial sums in registers (the in-block fold that follows is the same work in either); printed:
// microseconds per launch (HIP events over 200 launches) and the median in-kernel cycles of staging and sweep.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o devlibs/tile16_ubench tools/tile16_ubench.hip && devlibs/tile16_ubench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int D = 12, RS = 16;   // a row: 12 coordinates, |s|^2, weight, 2 pad floats
constexpr int NT = 512;

struct Acc {
    float score, gsum, g[D];
};

// one pair in the expanded form, Polyharmonic(1): K = r, dK/d(d^2) = 1 / (2 r)
__device__ __forceinline__ void pair(const float (&x)[D], float xx, const float (&s)[RS], Acc& a) {
    float dot = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) dot = __builtin_fmaf(x[k], s[k], dot);
    float d2 = __builtin_fmaf(-2.0f, dot, xx + s[D]);
    d2 = d2 > 1e-12f ? d2 : 1e-12f;
    const float ri = __builtin_amdgcn_rsqf(d2);
    a.score = __builtin_fmaf(s[D + 1] * d2, ri, a.score);
    const float coef = s[D + 1] * ri;
    a.gsum += coef;
#pragma unroll
    for (int k = 0; k < D; ++k) a.g[k] = __builtin_fmaf(coef, s[k], a.g[k]);
}

__device__ __forceinline__ void finish(const Acc& a, float* out, int idx) {
    float t = a.score + a.gsum;
#pragma unroll
    for (int k = 0; k < D; ++k) t += a.g[k];
    out[idx] = t;
}

// S: row operands through the scalar cache
__global__ __launch_bounds__(NT) void sweep_scalar(const float* __restrict__ rows, const float* __restrict__ xs, int S, int ys, float* out,
                                                   unsigned long long* cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x / ys, part = blockIdx.x % ys;
    float x[D], xx = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) { x[k] = xs[(size_t)(tile * 64 + lane) * D + k]; xx += x[k] * x[k]; }
    Acc a{};
    const int r0 = S * part / ys, r1 = S * (part + 1) / ys;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int j = r0 + wave; j < r1; j += NT / 64) {
        const float* r = rows + (size_t)__builtin_amdgcn_readfirstlane(j) * RS;   // uniform: s_load
        float s[RS];
#pragma unroll
        for (int k = 0; k < RS; ++k) s[k] = r[k];
        pair(x, xx, s, a);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    finish(a, out, blockIdx.x * NT + tid);
    if (tid == 0) { cyc[2 * blockIdx.x] = 0; cyc[2 * blockIdx.x + 1] = t1 - t0; }
}

// L: all rows in LDS, lane = (configuration, quarter)
__global__ __launch_bounds__(NT) void sweep_lds(const float* __restrict__ rows, const float* __restrict__ xs, int S, float* out,
                                                unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cfg = lane & 15, quarter = lane >> 4;
    const int per = (S + 31) / 32;                    // rows per slice (8 waves x 4 quarters)
    const int stride = per * RS + 4;                  // a slice starts four banks further than the one before
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int e = tid; e < S * (RS / 4); e += NT) {    // stage: one float4 per thread and step
        const int j = e / (RS / 4), q = e % (RS / 4);
        const float4 v = reinterpret_cast<const float4*>(rows)[(size_t)j * (RS / 4) + q];
        const int sl = j / per, jj = j % per;
        *reinterpret_cast<float4*>(&lds[sl * stride + jj * RS + q * 4]) = v;
    }
    float x[D], xx = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) { x[k] = xs[(size_t)(blockIdx.x * 16 + cfg) * D + k]; xx += x[k] * x[k]; }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    Acc a{};
    const int sl = wave * 4 + quarter;
    const int n = S - sl * per < per ? (S - sl * per > 0 ? S - sl * per : 0) : per;
    const float* base = lds + sl * stride;
    for (int jj = 0; jj < per; ++jj) {                // (uniform trip count; a short last slice repeats its last row with weight 0)
        const float* r = base + (jj < n ? jj : 0) * RS;
        float s[RS];
#pragma unroll
        for (int q = 0; q < RS / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(r + q * 4);
            s[q * 4] = v.x; s[q * 4 + 1] = v.y; s[q * 4 + 2] = v.z; s[q * 4 + 3] = v.w;
        }
        if (jj >= n) s[D + 1] = 0.0f;
        pair(x, xx, s, a);
    }
    const unsigned long long t2 = __builtin_readcyclecounter();
    finish(a, out, blockIdx.x * NT + tid);
    if (tid == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = t2 - t1; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static unsigned long long median(std::vector<unsigned long long> v, int off) {
    std::vector<unsigned long long> w;
    for (size_t i = off; i < v.size(); i += 2) w.push_back(v[i]);
    std::sort(w.begin(), w.end());
    return w[w.size() / 2];
}

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 1000, B = argc > 2 ? atoi(argv[2]) : 4096, ys = argc > 3 ? atoi(argv[3]) : 4;
    std::vector<float> rows((size_t)S * RS), xs((size_t)B * D);
    srand(1);
    for (auto& v : rows) v = (float)rand() / RAND_MAX - 0.5f;
    for (int j = 0; j < S; ++j) {
        float ss = 0;
        for (int k = 0; k < D; ++k) ss += rows[(size_t)j * RS + k] * rows[(size_t)j * RS + k];
        rows[(size_t)j * RS + D] = ss;
    }
    for (auto& v : xs) v = (float)rand() / RAND_MAX - 0.5f;
    float *dr, *dx, *dout; unsigned long long* dc;
    const int blocksS = B / 64 * ys, blocksL = B / 16, maxb = std::max(blocksS, blocksL);
    CK(hipMalloc(&dr, rows.size() * 4)); CK(hipMalloc(&dx, xs.size() * 4)); CK(hipMalloc(&dout, (size_t)maxb * NT * 4));
    CK(hipMalloc(&dc, (size_t)maxb * 16));
    CK(hipMemcpy(dr, rows.data(), rows.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, xs.data(), xs.size() * 4, hipMemcpyHostToDevice));
    const int per = (S + 31) / 32;
    const size_t ldsL = (size_t)32 * (per * RS + 4) * 4;
    CK(hipFuncSetAttribute((const void*)sweep_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsL));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("S = %d rows of %d floats, B = %d configurations, D = %d (expanded form, Polyharmonic(1))\n", S, RS, B, D);
    for (int which = 0; which < 2; ++which) {
        const int blocks = which ? blocksL : blocksS;
        auto launch = [&]() {
            if (which) sweep_lds<<<blocks, NT, ldsL, 0>>>(dr, dx, S, dout, dc);
            else sweep_scalar<<<blocks, NT, 0, 0>>>(dr, dx, S, ys, dout, dc);
        };
        for (int w = 0; w < 20; ++w) launch();
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 200; ++r) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c((size_t)blocks * 2);
        CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
        const double pairs_per_lane = which ? per : (double)S / ys / (NT / 64);
        printf("%s: %4d blocks x %d threads, %6.2f us per launch; in-kernel medians: staging %6llu cycles, sweep %6llu cycles = %5.1f cycles per pair and wave (%.0f pairs per lane)\n",
               which ? "L  rows in LDS, 16 configurations x 4 quarters per wave " : "S  rows through the scalar cache, 64 configurations per wave",
               blocks, NT, ms / 200 * 1e3, median(c, 0), median(c, 1), (double)median(c, 1) / pairs_per_lane, pairs_per_lane);
    }
    printf("(LDS per block in L: %.1f KB)\n", ldsL / 1024.0);
    return 0;
}
