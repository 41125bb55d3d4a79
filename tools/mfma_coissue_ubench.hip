// mfma_coissue_ubench.hip — does the fp32 matrix instruction (v_mfma_f32_16x16x4_f32) run BESIDE the fp32 VALU on
// gfx950, or do they share the SIMD's fp32 rate?  The fused score kernel could move its gradient fold (a
// (configurations x supports) . (supports x features) GEMM) onto the matrix cores only if the two pipes
// overlap.  Each test runs N waves per SIMD on every CU and reports the time of
//   M   : MFMA only (4 independent accumulators per wave)
//   V   : VALU only (v_fma_f32 or v_pk_fma_f32, 8 independent chains)
//   M+V : the same number of each, interleaved in ONE wave's instruction stream (1 MFMA : R VALU)
//   M|V : waves 0-3 of every 512-thread block MFMA only, waves 4-7 VALU only (each SIMD hosts one of each)
// If the pipes overlap, M+V and M|V cost max(M, V); if they share one datapath they cost M + V.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue_ubench.hip -o build/mfma_coissue_ubench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0: M, 1: V, 2: M+V interleaved, 3: M|V by wave parity.  PK: VALU op is v_pk_fma_f32 (else v_fma_f32).
// R = VALU instructions per MFMA.
template <int MODE, int PK, int R>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    v4f acc[4];
    float a[8];
    v2f p[8];
    for (int i = 0; i < 4; ++i) acc[i] = v4f{seed, seed, seed, seed};
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1.f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const v2f m2 = {m, m}, c2 = {c, c};
    const float av = seed * 1e-3f, bv = seed * 2e-3f;
    const int wave = threadIdx.x >> 8;  // M|V: a 512-thread block puts two waves on every SIMD, w and w + 4: one of each kind
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (do_m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(av), "v"(bv));
            if (do_v) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i % 8]) : "v"(m2), "v"(c2));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i % 8]) : "v"(m), "v"(c));
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int PK, int R>
float run(int waves_per_simd, int iters) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd / 2;  // 512 threads = 2 waves per SIMD per block
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, PK, R><<<blocks, 512>>>(out, 50, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, PK, R><<<blocks, 512>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

template <int PK, int R>
void suite(int w) {
    const int iters = 4000;
    const float tm = run<0, PK, R>(w, iters), tv = run<1, PK, R>(w, iters), tb = run<2, PK, R>(w, iters), ts = run<3, PK, R>(w, iters);
    // MODE 3 runs half the waves on each stream: its M and V parts are half of tm / tv each
    printf("waves/SIMD=%d  %-13s x%-2d per MFMA:  M %.3f ms   V %.3f ms   M+V interleaved %.3f ms  (max %.3f, sum %.3f)   "
           "M|V by wave %.3f ms  (max %.3f, sum %.3f)\n",
           w, PK ? "v_pk_fma_f32" : "v_fma_f32", R, tm, tv, tb, tm > tv ? tm : tv, tm + tv, ts, (tm > tv ? tm : tv) / 2, (tm + tv) / 2);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s  CUs=%d  clock=%.0f MHz; MFMA = v_mfma_f32_16x16x4_f32 (2048 flop, 32 cycles/SIMD nominal)\n", prop.gcnArchName,
           prop.multiProcessorCount, prop.clockRate / 1e3);
    for (int w : {2, 4, 8}) {
        suite<0, 4>(w);
        suite<0, 8>(w);
        suite<0, 16>(w);
        suite<1, 4>(w);
        suite<1, 8>(w);
    }
    return 0;
}
