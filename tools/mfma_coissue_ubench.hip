// mfma_coissue_ubench.hip — does the fp32 matrix instruction (v_mfma_f32_16x16x4_f32) run BESIDE the fp32 VALU on
// gfx950, or do they share the SIMD's fp32 rate?  The fused score kernel could move its gradient fold (a
// (configurations x supports) . (supports x features) GEMM) onto the matrix cores only if the two pipes
// overlap.  Each test runs N waves per SIMD on every CU and reports the time of
//   M   : MFMA only (4 independent accumulators per wave)
//   V   : VALU only (v_fma_f32 or v_pk_fma_f32, 8 independent chains)
//   M+V : the same number of each, interleaved in ONE wave's instruction stream (1 MFMA : R VALU)
//   M|V : half the waves of every SIMD MFMA only, the other half VALU only (roles by hardware SIMD id)
// If the pipes overlap, M+V and M|V cost max(M, V); if they share one datapath they cost M + V.
// Build: hipcc --offload-arch=gfx950 -O3 [-DBF16 [-DM32]] tools/mfma_coissue_ubench.hip -o tools/mfma_coissue_ubench
//   default: v_mfma_f32_16x16x4_f32;  -DBF16: v_mfma_f32_16x16x32_bf16;  -DBF16 -DM32: v_mfma_f32_32x32x16_bf16
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// MODE 0: M, 1: V, 2: M+V interleaved, 3: M|V by wave parity.  PK: VALU op is v_pk_fma_f32 (else v_fma_f32).
// R = VALU instructions per MFMA.
template <int MODE, int PK, int R>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
#ifdef M32
    v16f acc[4];
#else
    v4f acc[4];
#endif
    float a[8];
    v2f p[8];
#ifdef M32
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed;
#else
    for (int i = 0; i < 4; ++i) acc[i] = v4f{seed, seed, seed, seed};
#endif
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1.f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const v2f m2 = {m, m}, c2 = {c, c};
    const float av = seed * 1e-3f, bv = seed * 2e-3f;
    v8bf a8, b8;
    for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)(seed * 0.01f * e); b8[e] = (__bf16)(seed * 0.02f * e); }
    // M|V: every SIMD hosts as many MFMA-only waves as VALU-only waves.  The role is decided ONCE, outside the loops (each
    // loop body is fixed at compile time), from the SIMD the wave actually landed on (HW_ID bits 5:4) and its order of
    // arrival there within the block - the hardware does not place wave w on SIMD w % 4.  (Round 2's version tested a
    // run-time flag inside the loop and computed the wave index with the wrong shift: its M|V column is void.)
    __shared__ int arrivals[4];
    if (threadIdx.x < 4) arrivals[threadIdx.x] = 0;
    __syncthreads();
    int role = 0;
    if (MODE == 3) {
        const int simd = (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4)) & 3;
        int slot = 0;
        if ((threadIdx.x & 63) == 0) slot = atomicAdd(&arrivals[simd], 1);
        role = __builtin_amdgcn_readfirstlane(slot) & 1;
    }
    auto m_step = [&](int u) __attribute__((always_inline)) {
#ifdef M32
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a8), "v"(b8));
#elif defined(BF16)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a8), "v"(b8));
#else
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(av), "v"(bv));
#endif
    };
    auto v_step = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i % 8]) : "v"(m2), "v"(c2));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i % 8]) : "v"(m), "v"(c));
        }
    };
    const bool m_only = MODE == 0 || (MODE == 3 && role == 0);
    const bool v_only = MODE == 1 || (MODE == 3 && role == 1);
    if (m_only) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) m_step(u);
        }
    } else if (v_only) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v_step();
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                m_step(u);
                v_step();
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
    // MODE 3 runs half the waves on each stream (one MFMA wave + one VALU wave per SIMD at 2 waves/SIMD): its M and V parts
    // are half of tm / tv each
    printf("waves/SIMD=%d  %-13s x%-2d per MFMA:  M %.3f ms   V %.3f ms   M+V interleaved %.3f ms  (max %.3f, sum %.3f)   "
           "M|V by wave %.3f ms  (max %.3f, sum %.3f)\n",
           w, PK ? "v_pk_fma_f32" : "v_fma_f32", R, tm, tv, tb, tm > tv ? tm : tv, tm + tv, ts, (tm > tv ? tm : tv) / 2, (tm + tv) / 2);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s  CUs=%d  clock=%.0f MHz; MFMA = %s", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3,
#ifdef BF16
 #ifdef M32
"v_mfma_f32_32x32x16_bf16 (32768 flop, 32 cycles/SIMD nominal)"
#else
"v_mfma_f32_16x16x32_bf16 (16384 flop, 16 cycles/SIMD nominal)"
#endif

#else
 "v_mfma_f32_16x16x4_f32 (2048 flop, 32 cycles/SIMD nominal)"
#endif
);
    if (0) printf("%s %d %f\n", prop.gcnArchName,
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
