// mfma_bf16_coissue_ubench.hip — does v_mfma_f32_16x16x32_bf16 run BESIDE fp32 VALU work on gfx950?
// (tools/mfma_coissue_ubench.hip showed that the *fp32* matrix instruction does not: it shares the fp32 datapath.)
// The split-operand sweep puts two GEMMs on bf16 matrix instructions and keeps ~10 VALU
// instructions per pair (kernel function + three-way bf16 split of the gradient coefficient); it only pays if the
// two pipes overlap.  Mix per "pair block" (16 supports x 16 configurations = 4 values per lane):
//   6 MFMA 16x16x32 bf16 (3 distance GEMM + 3 of the 6 gradient-fold products)  = 96 matrix cycles nominal
//   4 x { v_max, v_rsq, v_mul, v_fma, v_mul, v_and, v_sub, v_and, v_sub } + 6 v_perm + 2 v_min3 = 44 VALU
//   M: MFMA only   V: VALU only   M+V: both in one wave's stream, interleaved
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_coissue_ubench.hip -o build/mfma_bf16_coissue_ubench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int MODE, int NT>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    v4f acc1[NT], acc2[NT];
    v8bf a[3], b[NT][3], c[3];
    float w[4], sc[NT], d2min = 1e30f;
    for (int i = 0; i < NT; ++i) {
        acc1[i] = v4f{seed, seed, seed, seed};
        acc2[i] = v4f{0, 0, 0, 0};
        sc[i] = 0;
    }
    for (int p = 0; p < 3; ++p) {
        for (int e = 0; e < 8; ++e) {
            a[p][e] = (__bf16)(seed * 0.01f * (e + p + 1) + 1e-3f * threadIdx.x);
            c[p][e] = (__bf16)(seed * 0.02f * (e + p + 1));
            for (int i = 0; i < NT; ++i) b[i][p][e] = (__bf16)(seed * 0.03f * (e + i + 1));
        }
    }
    for (int t = 0; t < 4; ++t) w[t] = seed + t;
    constexpr bool DM = MODE == 0 || MODE >= 2, DV = MODE >= 1, DEP = MODE == 2;
    v4f own = v4f{seed + 1.f, seed + 2.f, seed + 3.f, seed + 4.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            v4f d2 = acc1[i];
            if (DM) {
                d2 = v4f{seed, seed, seed, seed};
#pragma unroll
                for (int p = 0; p < 3; ++p) d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], b[i][p], d2, 0, 0, 0);
            }
            unsigned int ch[4], cm[4], cl[4];
            v4f d2m = d2;
            if (DV && !DEP) { d2 = own; own = own + v4f{1e-3f, 1e-3f, 1e-3f, 1e-3f}; }
            if (DV) {
                d2min = fminf(d2min, fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3])));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float d2c = fmaxf(d2[t], 1e-30f);
                    const float ri = __builtin_amdgcn_rsqf(d2c);
                    const float val = d2c * ri;
                    sc[i] = fmaf(w[t], val, sc[i]);
                    const float cf = ri * w[t];
                    const unsigned int hb = __float_as_uint(cf) & 0xffff0000u;
                    const float r1 = cf - __uint_as_float(hb);
                    const unsigned int mb = __float_as_uint(r1) & 0xffff0000u;
                    const float r2 = r1 - __uint_as_float(mb);
                    ch[t] = hb;
                    cm[t] = mb;
                    cl[t] = __float_as_uint(r2);
                }
                // high halves of two registers -> one packed bf16 pair
                typedef unsigned int v4u __attribute__((ext_vector_type(4)));
                v4u ph, pm, pl;
                ph[0] = __builtin_amdgcn_perm(ch[1], ch[0], 0x07060302u);
                ph[1] = __builtin_amdgcn_perm(ch[3], ch[2], 0x07060302u);
                pm[0] = __builtin_amdgcn_perm(cm[1], cm[0], 0x07060302u);
                pm[1] = __builtin_amdgcn_perm(cm[3], cm[2], 0x07060302u);
                pl[0] = __builtin_amdgcn_perm(cl[1], cl[0], 0x07060302u);
                pl[1] = __builtin_amdgcn_perm(cl[3], cl[2], 0x07060302u);
                ph[2] = ph[0]; ph[3] = ph[1]; pm[2] = pm[0]; pm[3] = pm[1]; pl[2] = pl[0]; pl[3] = pl[1];
                if (DEP) {
                    c[0] = __builtin_bit_cast(v8bf, ph);
                    c[1] = __builtin_bit_cast(v8bf, pm);
                    c[2] = __builtin_bit_cast(v8bf, pl);
                } else {
                    sc[i] += __uint_as_float(ph[0] ^ pm[0] ^ pl[0] ^ ph[1] ^ pm[1] ^ pl[1]);
                }
            }
            if (DM) {
#pragma unroll
                for (int p = 0; p < 3; ++p) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], c[p], acc2[i], 0, 0, 0);
            }
            if (!DM) acc1[i] = d2 + acc1[i];
            else if (!DEP) acc1[i] = d2m;
        }
    }
    float s = d2min;
    for (int i = 0; i < NT; ++i) s += acc1[i].x + acc2[i].x + acc2[i].y + acc2[i].z + acc2[i].w + sc[i];
    for (int p = 0; p < 3; ++p) s += (float)c[p][0];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int NT>
float run(int waves_per_simd, int iters) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;  // 256 threads = 1 wave per SIMD per block
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, NT><<<blocks, 256>>>(out, 50, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NT><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

template <int NT>
void suite(int w) {
    const int iters = 2000;
    const float tm = run<0, NT>(w, iters), tv = run<1, NT>(w, iters), tb = run<2, NT>(w, iters), ti = run<3, NT>(w, iters);
    // per pair block: iters * NT blocks per wave, w waves per SIMD -> SIMD cycles per pair block at 2.4 GHz
    const double blocks_per_simd = (double)iters * NT * w;
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / blocks_per_simd; };
    printf("waves/SIMD=%d NT=%d:  M %.3f ms (%.0f cyc/blk)   V %.3f ms (%.0f cyc/blk)   M+V dependent %.3f ms (%.0f cyc/blk)   M+V independent %.3f ms (%.0f cyc/blk)  [max %.0f, sum %.0f]\n",
           w, NT, tm, cyc(tm), tv, cyc(tv), tb, cyc(tb), ti, cyc(ti), cyc(tm > tv ? tm : tv), cyc(tm + tv));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s  CUs=%d  clock=%.0f MHz; pair block = 6 x v_mfma_f32_16x16x32_bf16 (96 cyc nominal) + 44 VALU, 256 pairs\n",
           prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3);
    for (int w : {1, 2, 4}) {
        suite<2>(w);
        suite<4>(w);
    }
    return 0;
}
