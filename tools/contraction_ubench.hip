// contraction_ubench.hip — the three contractions of the fused score kernel, each timed on the VALU and on
// the matrix cores with the kernel's own data flow (one lane = one configuration, support rows arrive through scalar or
// vector loads, MFMA operands are built with the v_permlane swaps the kernel uses), every variant checked against the VALU
// result it replaces:
//
//   fold   gx[b][k]  = sum_j coef[b][j] * s[j][k]            D = 12, C = 1 (the headline's gradient fold)
//            valu    : 6 v_pk_fma_f32 per row                                       (what ships)
//            mfma32  : v_mfma_f32_16x16x4_f32, 4 per 4 rows + the 4x4 lane transpose (DCX_MFMA=1, bitwise equal)
//            bf16x3  : coef and s split into three bf16 planes, six plane products on v_mfma_f32_16x16x32_bf16
//                      (24 per 32 rows), coefficient split + pack + transposes on the VALU  (relative error ~1e-6)
//   kw     sc[b][c]  = sum_j K[b][j] * W[j][c]               C = 5, 8 (MultiDiffCo score)
//            valu    : C v_fma_f32 per row
//            mfma32  : 4 MFMA per 4 rows, classes padded to 16 columns              (bitwise equal)
//   gwt    wb[b][j]  = sum_c up[b][c] * W[j][c]              C = 5, 8 (upstream gradient . W^T)
//            valu    : C v_fma_f32 per row
//            mfma32  : 8 MFMA per 16 rows (A = W block, B = upstream fragment) + 4 lane transposes  (bitwise equal)
//
// K / coef are synthetic (one v_mul per pair: x_b * t_j), so the numbers are the contraction's own cost per support row
// and wave, not the whole sweep's.  Every CU runs `occ` waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/contraction_ubench.hip -o devlibs/contraction_ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const float* cfloat_ptr;

constexpr int D = 12;

// four per-lane values (one per k) -> fragments: f[t][lane i + 16 k] = c_k[lane 16 t + i]
__device__ __forceinline__ void frags(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&f)[4]) {
    auto s02 = __builtin_amdgcn_permlane32_swap(c0, c2, false, false);
    auto s13 = __builtin_amdgcn_permlane32_swap(c1, c3, false, false);
    auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    f[0] = t01[0]; f[1] = t01[1]; f[2] = t23[0]; f[3] = t23[1];
}
__device__ __forceinline__ uint32_t fu(float v) { return __float_as_uint(v); }
__device__ __forceinline__ float uf(uint32_t v) { return __uint_as_float(v); }

// Scalar row loads, two rows per buffer, issued one buffer ahead of their use (the kernel's sweep does the same): the
// VALU forms are then bound by their instructions, not by the scalar-load latency.
template <int N>
__device__ __forceinline__ void load2(float (&dst)[2][N], cfloat_ptr rows, int j, int S) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        cfloat_ptr r = rows + (size_t)((j + e < S) ? j + e : S - 1) * 16;
#pragma unroll
        for (int k = 0; k < N; ++k) dst[e][k] = r[k];
    }
}
template <int N>
__device__ __forceinline__ void loadt(float (&dst)[N], cfloat_ptr t, int j) {  // N consecutive t_j (padded past S)
#pragma unroll
    for (int k = 0; k < N; ++k) dst[k] = t[j + k];
}
#define PIPE_FENCE() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)

// ---- fold -------------------------------------------------------------------------------------------------------------
// rows: [S][16] floats (12 coordinates, t_j at column 12, padding); planes: three bf16 planes of the coordinates laid out
// [S / 8][16 features][8 supports] so that lane (n, k') of a B operand reads 16 contiguous bytes.
template <int VAR>
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ rows_g, const float* __restrict__ t_g,
                                                   const uint16_t* __restrict__ planes, size_t plane_stride,
                                                   float* __restrict__ out, int S) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float xb = 1.0f + 0.001f * (float)((gw * 64 + lane) % 977);
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)rows_g;
    cfloat_ptr tt = (cfloat_ptr)(uintptr_t)t_g;
    float gx[D];
#pragma unroll
    for (int k = 0; k < D; ++k) gx[k] = 0.0f;
    if constexpr (VAR == 0) {
        auto row = [&](const float (&r)[13]) __attribute__((always_inline)) {
            const float coef = xb * r[12];
#pragma unroll
            for (int k = 0; k < D; k += 2) {
                const v2f rv = {r[k], r[k + 1]};
                v2f g = {gx[k], gx[k + 1]};
                g = __builtin_elementwise_fma(v2f{coef, coef}, rv, g);
                gx[k] = g.x;
                gx[k + 1] = g.y;
            }
        };
        float ra[2][13], rb[2][13];
        load2(ra, rows, 0, S);
        for (int j = 0; j < S; j += 4) {
            load2(rb, rows, j + 2, S);
            __builtin_amdgcn_sched_barrier(0);
            row(ra[0]);
            row(ra[1]);
            PIPE_FENCE();
            load2(ra, rows, j + 4, S);
            __builtin_amdgcn_sched_barrier(0);
            row(rb[0]);
            row(rb[1]);
            PIPE_FENCE();
        }
    } else if constexpr (VAR == 1) {
        const int grp = lane >> 4, col = lane & 15;
        v4f acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
        const float* bp = rows_g + grp * 16 + col;
        float bcur = bp[0];
        float tc[4], tn[4];
        loadt(tc, tt, 0);
        for (int j = 0; j < S; j += 4) {
            bp += 64;
            const float bnext = bp[0];  // (the row array is padded)
            loadt(tn, tt, j + 4);
            __builtin_amdgcn_sched_barrier(0);
            float c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = xb * tc[e];
            uint32_t f[4];
            frags(fu(c[0]), fu(c[1]), fu(c[2]), fu(c[3]), f);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf(f[t]), bcur, acc[t], 0, 0, 0);
            bcur = bnext;
            PIPE_FENCE();
#pragma unroll
            for (int e = 0; e < 4; ++e) tc[e] = tn[e];
        }
        __shared__ float scr[4][64 * 13];
        float* w = scr[threadIdx.x >> 6];
        if (col < D) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) w[(16 * t + 4 * grp + i) * 13 + col] = acc[t][i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < D; ++k) gx[k] = w[lane * 13 + k];
    } else {
        const int grp = lane >> 4, col = lane & 15;
        v4f acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
        const v4u* p0 = (const v4u*)planes + grp * 16 + col;             // 16 B per (support octet, feature)
        const v4u* p1 = (const v4u*)(planes + plane_stride) + grp * 16 + col;
        const v4u* p2 = (const v4u*)(planes + 2 * plane_stride) + grp * 16 + col;
        float tc[32], tn[32];
        loadt(tc, tt, 0);
        v4u bh = p0[0], bm = p1[0], bl = p2[0];
        for (int j = 0; j < S; j += 32) {
            loadt(tn, tt, j + 32);
            const v4u nh = p0[(j / 8 + 4) * 16], nm = p1[(j / 8 + 4) * 16], nl = p2[(j / 8 + 4) * 16];  // (planes are padded)
            __builtin_amdgcn_sched_barrier(0);
            uint32_t ph[16], pm[16], pl[16];  // the 32 coefficients of this step as packed bf16 pairs, three planes
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                uint32_t h[2], mi[2], lo[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float c = xb * tc[2 * m + e];
                    h[e] = fu(c) & 0xFFFF0000u;
                    const float r1 = c - uf(h[e]);
                    mi[e] = fu(r1) & 0xFFFF0000u;
                    lo[e] = fu(r1 - uf(mi[e]));
                }
                ph[m] = __builtin_amdgcn_perm(h[1], h[0], 0x07060302u);
                pm[m] = __builtin_amdgcn_perm(mi[1], mi[0], 0x07060302u);
                pl[m] = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u);
            }
            v4u ah[4], am[4], al[4];  // A fragments per tile: dword d of lane (i, k') = pairs 4 k' + d of configuration 16 t + i
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t f[4];
                frags(ph[d], ph[4 + d], ph[8 + d], ph[12 + d], f);
#pragma unroll
                for (int t = 0; t < 4; ++t) ah[t][d] = f[t];
                frags(pm[d], pm[4 + d], pm[8 + d], pm[12 + d], f);
#pragma unroll
                for (int t = 0; t < 4; ++t) am[t][d] = f[t];
                frags(pl[d], pl[4 + d], pl[8 + d], pl[12 + d], f);
#pragma unroll
                for (int t = 0; t < 4; ++t) al[t][d] = f[t];
            }
            auto mm = [&](const v4u& av, const v4u& bv, v4f c) __attribute__((always_inline)) {
                return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, av), __builtin_bit_cast(v8bf, bv), c, 0, 0, 0);
            };
#pragma unroll
            for (int t = 0; t < 4; ++t) {  // small terms first
                v4f c = acc[t];
                c = mm(am[t], bm, c);
                c = mm(al[t], bh, c);
                c = mm(ah[t], bl, c);
                c = mm(am[t], bh, c);
                c = mm(ah[t], bm, c);
                c = mm(ah[t], bh, c);
                acc[t] = c;
            }
            bh = nh; bm = nm; bl = nl;
            PIPE_FENCE();
#pragma unroll
            for (int e = 0; e < 32; ++e) tc[e] = tn[e];
        }
        __shared__ float scr[4][64 * 13];
        float* w = scr[threadIdx.x >> 6];
        if (col < D) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) w[(16 * t + 4 * grp + i) * 13 + col] = acc[t][i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < D; ++k) gx[k] = w[lane * 13 + k];
    }
    float* o = out + ((size_t)gw * 64 + lane) * D;
#pragma unroll
    for (int k = 0; k < D; ++k) o[k] = gx[k];
}

// ---- d2 ---------------------------------------------------------------------------------------------------------------
// The squared distance in its expanded form, d2[b][j] = (|x_b|^2 + |s_j|^2) + sum_k (-2 x_bk) s_jk, D = 12: the one
// contraction of the sweep whose per-lane operand (x) is loop invariant, so a split-operand matrix-core form needs no
// per-pair splitting at all.
//   valu   : 6 v_pk_fma_f32 per row + the adds (what the XF sweep ships)
//   bf16x3 : -2x split once per lane into three bf16 planes (B fragments, loop invariant), s split on the host; the six plane
//            products (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) are laid side by side along K (6 x 16 slots = three
//            K = 32 instructions), the support block is the A operand, the 16 x 16 outputs come back to one lane per
//            configuration through the 4 x 4 lane transpose: 12 v_mfma_f32_16x16x32_bf16 + 16 swaps per 16 rows and wave.
// rows: [S][16] floats, column 12 = t_j, column 13 = |s_j|^2.  aplanes: [S / 16][3 chunks][16 supports][4 k'][8] bf16.
template <int VAR>
__global__ __launch_bounds__(256) void d2_kernel(const float* __restrict__ rows_g, const float* __restrict__ ss_g,
                                                 const uint16_t* __restrict__ aplanes, float* __restrict__ out, int S) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)rows_g;
    cfloat_ptr ss = (cfloat_ptr)(uintptr_t)ss_g;
    float x[D], xx = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        x[k] = 0.37f * (float)(((gw * 64 + lane) * 7 + 13 * k) % 31) / 31.0f - 0.2f;
        xx = fmaf(x[k], x[k], xx);
    }
    float acc = 0.0f;  // consumes d2 the way the sweep would start to: here simply sum_j d2
    if constexpr (VAR == 0) {
        v2f xm[D / 2];
#pragma unroll
        for (int k = 0; k < D; k += 2) xm[k / 2] = v2f{-2.0f * x[k], -2.0f * x[k + 1]};
        auto row = [&](const float (&r)[14]) __attribute__((always_inline)) {
            v2f a = {xx + r[13], 0.0f};
#pragma unroll
            for (int k = 0; k < D; k += 2) a = __builtin_elementwise_fma(xm[k / 2], v2f{r[k], r[k + 1]}, a);
            acc += a.x + a.y;
        };
        float ra[2][14], rb[2][14];
        load2(ra, rows, 0, S);
        for (int j = 0; j < S; j += 4) {
            load2(rb, rows, j + 2, S);
            __builtin_amdgcn_sched_barrier(0);
            row(ra[0]);
            row(ra[1]);
            PIPE_FENCE();
            load2(ra, rows, j + 4, S);
            __builtin_amdgcn_sched_barrier(0);
            row(rb[0]);
            row(rb[1]);
            PIPE_FENCE();
        }
    } else {
        const int grp = lane >> 4, col = lane & 15;
        // B fragments: K slot (term, kk) of chunk c = term 2c + (slot / 16): the x plane each term multiplies
        //   terms: 0 hi.hi  1 hi.mid  2 mid.hi  3 hi.lo  4 lo.hi  5 mid.mid   (x plane: hi hi mid hi lo mid)
        __shared__ uint16_t sxp[4][3][64][16];  // [wave][x plane][configuration][feature, zero padded to 16]
        uint16_t(*xp)[64][16] = sxp[threadIdx.x >> 6];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float r = (k < D) ? -2.0f * x[k] : 0.0f;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const uint32_t u = fu(r) & 0xFFFF0000u;
                xp[pl][lane][k] = (uint16_t)(u >> 16);
                r -= uf(u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        v4u bfrag[4][3];  // [tile][chunk]: lane (n, k') holds K slots 8 k' .. 8 k' + 7 of configuration 16 t + n
        constexpr int xplane_of_term[6] = {0, 0, 1, 0, 2, 1};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int term = 2 * c + (grp >> 1), f0 = 8 * (grp & 1);
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t lo = 0, hi = 0;
#pragma unroll
                    for (int tt = 0; tt < 6; ++tt)
                        if (tt == term) {
                            lo = xp[xplane_of_term[tt]][16 * t + col][f0 + 2 * e];
                            hi = xp[xplane_of_term[tt]][16 * t + col][f0 + 2 * e + 1];
                        }
                    w[e] = lo | (hi << 16);
                }
                bfrag[t][c] = v4u{w[0], w[1], w[2], w[3]};
            }
        const v4u* ap = (const v4u*)aplanes + col * 4 + grp;  // 16 B per (support, k')
        float sc_[16], sn[16];
        loadt(sc_, ss, 0);
        v4u a0 = ap[0], a1 = ap[64], a2 = ap[128];
        for (int j = 0; j < S; j += 16) {
            loadt(sn, ss, j + 16);
            const size_t nb = (size_t)(j / 16 + 1) * 192;
            const v4u n0 = ap[nb], n1 = ap[nb + 64], n2 = ap[nb + 128];
            __builtin_amdgcn_sched_barrier(0);
            v4f d[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v4f c = {0.f, 0.f, 0.f, 0.f};
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a2), __builtin_bit_cast(v8bf, bfrag[t][2]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a1), __builtin_bit_cast(v8bf, bfrag[t][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a0), __builtin_bit_cast(v8bf, bfrag[t][0]), c, 0, 0, 0);
                d[t] = c;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t f[4];
                frags(fu(d[0][i]), fu(d[1][i]), fu(d[2][i]), fu(d[3][i]), f);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc += (xx + sc_[4 * g + i]) + uf(f[g]);
            }
            a0 = n0; a1 = n1; a2 = n2;
            PIPE_FENCE();
#pragma unroll
            for (int e = 0; e < 16; ++e) sc_[e] = sn[e];
        }
    }
    out[(size_t)gw * 64 + lane] = acc;
}

// ---- the whole pair body of the expanded-form sweep, with the distance on the VALU or on the matrix cores --------------
// What a sweep that takes d2 from the bf16x3 form above would cost per row INCLUDING everything that stays on the VALU
// (clamp, v_rsq, coefficient, score, the 6 v_pk_fma of the gradient fold with the row in SGPRs), at the register budget
// it really needs (48 VGPRs of loop-invariant B fragments): the number to hold against the shipped sweep's.
template <int VAR>
__global__ __launch_bounds__(256) void xf_kernel(const float* __restrict__ rows_g, const float* __restrict__ ss_g,
                                                 const uint16_t* __restrict__ aplanes, float* __restrict__ out, int S) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)rows_g;
    float x[D], xx = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        x[k] = 0.37f * (float)(((gw * 64 + lane) * 7 + 13 * k) % 31) / 31.0f - 0.2f;
        xx = fmaf(x[k], x[k], xx);
    }
    const float thr = fmaxf(0.01f * xx, 1e-30f);
    v2f ga[D / 2];
#pragma unroll
    for (int k = 0; k < D / 2; ++k) ga[k] = v2f{0.0f, 0.0f};
    float sc = 0.0f, asum = 0.0f;
    // everything behind the distance: clamp, 1 / r, coefficient (w / r), score (w r = coef d2), fold, sum of coefficients
    auto tail = [&](const float (&r)[14], float d2raw) __attribute__((always_inline)) {
        const float d2 = fmaxf(d2raw, thr);
        const float coef = r[12] * __builtin_amdgcn_rsqf(d2);
        sc = fmaf(coef, d2, sc);
        const v2f c2 = {coef, coef};
#pragma unroll
        for (int k = 0; k < D; k += 2) ga[k / 2] = __builtin_elementwise_fma(c2, v2f{r[k], r[k + 1]}, ga[k / 2]);
        asum += coef;
    };
    if constexpr (VAR == 0) {
        v2f xm[D / 2];
#pragma unroll
        for (int k = 0; k < D; k += 2) xm[k / 2] = v2f{-2.0f * x[k], -2.0f * x[k + 1]};
        auto row = [&](const float (&r)[14]) __attribute__((always_inline)) {
            v2f a = {xx + r[13], 0.0f};
#pragma unroll
            for (int k = 0; k < D; k += 2) a = __builtin_elementwise_fma(xm[k / 2], v2f{r[k], r[k + 1]}, a);
            tail(r, a.x + a.y);
        };
        float ra[2][14], rb[2][14];
        load2(ra, rows, 0, S);
        for (int j = 0; j < S; j += 4) {
            load2(rb, rows, j + 2, S);
            __builtin_amdgcn_sched_barrier(0);
            row(ra[0]);
            row(ra[1]);
            PIPE_FENCE();
            load2(ra, rows, j + 4, S);
            __builtin_amdgcn_sched_barrier(0);
            row(rb[0]);
            row(rb[1]);
            PIPE_FENCE();
        }
    } else {
        const int grp = lane >> 4, col = lane & 15;
        // B fragments straight from registers: the three planes of -2x as packed pairs, then one 4 x 4 lane transpose per
        // (chunk, dword) hands every tile its fragment (no LDS)
        uint32_t pk[3][8];  // [plane][feature pair 2p, 2p + 1], features past D are zero
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float r0 = (2 * p < D) ? -2.0f * x[2 * p] : 0.0f, r1 = (2 * p + 1 < D) ? -2.0f * x[2 * p + 1] : 0.0f;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const uint32_t u0 = fu(r0) & 0xFFFF0000u, u1 = fu(r1) & 0xFFFF0000u;
                pk[pl][p] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
                r0 -= uf(u0);
                r1 -= uf(u1);
            }
        }
        v4u bfrag[4][3];
        constexpr int xplane_of_term[6] = {0, 0, 1, 0, 2, 1};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // lane (n, k'): k' = 0, 1 -> term 2c, features 8 k' + 2e, +1;  k' = 2, 3 -> term 2c + 1
                uint32_t f[4];
                frags(pk[xplane_of_term[2 * c]][e], pk[xplane_of_term[2 * c]][4 + e], pk[xplane_of_term[2 * c + 1]][e],
                      pk[xplane_of_term[2 * c + 1]][4 + e], f);
#pragma unroll
                for (int t = 0; t < 4; ++t) bfrag[t][c][e] = f[t];
            }
        (void)grp;
        const v4u* ap = (const v4u*)aplanes + col * 4 + (lane >> 4);
        v4u a0 = ap[0], a1 = ap[64], a2 = ap[128];
        for (int j = 0; j < S; j += 16) {
            const size_t nb = (size_t)(j / 16 + 1) * 192;
            const v4u n0 = ap[nb], n1 = ap[nb + 64], n2 = ap[nb + 128];
            float dot[16];
            {
                v4f d[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v4f c = {0.f, 0.f, 0.f, 0.f};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a2), __builtin_bit_cast(v8bf, bfrag[t][2]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a1), __builtin_bit_cast(v8bf, bfrag[t][1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a0), __builtin_bit_cast(v8bf, bfrag[t][0]), c, 0, 0, 0);
                    d[t] = c;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t f[4];
                    frags(fu(d[0][i]), fu(d[1][i]), fu(d[2][i]), fu(d[3][i]), f);
#pragma unroll
                    for (int g = 0; g < 4; ++g) dot[4 * g + i] = uf(f[g]);
                }
            }
            a0 = n0; a1 = n1; a2 = n2;
            // the 16 rows of this step through the two-row scalar pipeline
            float ra[2][14], rb[2][14];
            load2(ra, rows, j, S);
#pragma unroll
            for (int q = 0; q < 16; q += 4) {
                load2(rb, rows, j + q + 2, S);
                __builtin_amdgcn_sched_barrier(0);
                tail(ra[0], (xx + ra[0][13]) + dot[q]);
                tail(ra[1], (xx + ra[1][13]) + dot[q + 1]);
                PIPE_FENCE();
                load2(ra, rows, j + q + 4, S);
                __builtin_amdgcn_sched_barrier(0);
                tail(rb[0], (xx + rb[0][13]) + dot[q + 2]);
                tail(rb[1], (xx + rb[1][13]) + dot[q + 3]);
                PIPE_FENCE();
            }
        }
    }
    float* o = out + ((size_t)gw * 64 + lane) * 16;
#pragma unroll
    for (int k = 0; k < D / 2; ++k) {
        o[2 * k] = ga[k].x;
        o[2 * k + 1] = ga[k].y;
    }
    o[12] = sc;
    o[13] = asum;
}

// ---- kw and gwt -----------------------------------------------------------------------------------------------------------
// wrows: [S][16] floats (W[j][0..C-1], zeros up to column 8, t_j at column 12).  up: [configurations][8].
template <int C, int VAR>
__global__ __launch_bounds__(256) void kw_kernel(const float* __restrict__ wrows_g, const float* __restrict__ t_g,
                                                 float* __restrict__ out, int S) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float xb = 1.0f + 0.001f * (float)((gw * 64 + lane) % 977);
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)wrows_g;
    cfloat_ptr tt = (cfloat_ptr)(uintptr_t)t_g;
    float sc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sc[c] = 0.0f;
    if constexpr (VAR == 0) {
        auto row = [&](const float (&r)[C + 1]) __attribute__((always_inline)) {
            const float val = xb * r[C];
#pragma unroll
            for (int c = 0; c < C; ++c) sc[c] = fmaf(r[c], val, sc[c]);
        };
        float ra[2][C + 1], rb[2][C + 1];  // (W row, t_j) sit side by side in these rows: column C holds t_j
        load2(ra, rows, 0, S);
        for (int j = 0; j < S; j += 4) {
            load2(rb, rows, j + 2, S);
            __builtin_amdgcn_sched_barrier(0);
            row(ra[0]);
            row(ra[1]);
            PIPE_FENCE();
            load2(ra, rows, j + 4, S);
            __builtin_amdgcn_sched_barrier(0);
            row(rb[0]);
            row(rb[1]);
            PIPE_FENCE();
        }
    } else {
        const int grp = lane >> 4, col = lane & 15;
        v4f acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
        const float* bp = wrows_g + grp * 16 + (col < C ? col : 15);  // column 15 is zero
        float bcur = bp[0];
        float tc[4], tn[4];
        loadt(tc, tt, 0);
        for (int j = 0; j < S; j += 4) {
            bp += 64;
            const float bnext = bp[0];
            loadt(tn, tt, j + 4);
            __builtin_amdgcn_sched_barrier(0);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = xb * tc[e];
            uint32_t f[4];
            frags(fu(v[0]), fu(v[1]), fu(v[2]), fu(v[3]), f);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf(f[t]), bcur, acc[t], 0, 0, 0);
            bcur = bnext;
            PIPE_FENCE();
#pragma unroll
            for (int e = 0; e < 4; ++e) tc[e] = tn[e];
        }
        __shared__ float scr[4][64 * 9];
        float* w = scr[threadIdx.x >> 6];
        if (col < C) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) w[(16 * t + 4 * grp + i) * 9 + col] = acc[t][i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < C; ++c) sc[c] = w[lane * 9 + c];
    }
    float* o = out + ((size_t)gw * 64 + lane) * 8;
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = sc[c];
}

template <int C, int VAR>
__global__ __launch_bounds__(256) void gwt_kernel(const float* __restrict__ wrows_g, const float* __restrict__ t_g,
                                                  float* __restrict__ out, int S) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float xb = 1.0f + 0.001f * (float)((gw * 64 + lane) % 977);
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)wrows_g;
    cfloat_ptr tt = (cfloat_ptr)(uintptr_t)t_g;
    float up[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) up[c] = (c < C) ? 0.25f + 0.125f * (float)((lane + 3 * c + gw) % 7) : 0.0f;
    float acc = 0.0f;  // consumes wb the way the sweep does: coef = g * wb, here sum_j wb * val
    if constexpr (VAR == 0) {
        auto row = [&](const float (&r)[C + 1]) __attribute__((always_inline)) {
            const float val = xb * r[C];
            float wb = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) wb = fmaf(up[c], r[c], wb);
            acc = fmaf(wb, val, acc);
        };
        float ra[2][C + 1], rb[2][C + 1];
        load2(ra, rows, 0, S);
        for (int j = 0; j < S; j += 4) {
            load2(rb, rows, j + 2, S);
            __builtin_amdgcn_sched_barrier(0);
            row(ra[0]);
            row(ra[1]);
            PIPE_FENCE();
            load2(ra, rows, j + 4, S);
            __builtin_amdgcn_sched_barrier(0);
            row(rb[0]);
            row(rb[1]);
            PIPE_FENCE();
        }
    } else {
        const int grp = lane >> 4, col = lane & 15;
        // B fragments (loop invariant): lane (n, k) of tile t, class chunk kk holds up[configuration 16 t + n][4 kk + k]
        __shared__ float su[4][64 * 9];
        float* w = su[threadIdx.x >> 6];
#pragma unroll
        for (int c = 0; c < 8; ++c) w[lane * 9 + c] = up[c];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        float bu[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bu[t][kk] = w[(16 * t + col) * 9 + 4 * kk + grp];
        // A operand: lane (m, k) reads W[j + m][4 kk + k]
        const float* ap = wrows_g + col * 16 + grp;
        float tc[16], tn[16];
        loadt(tc, tt, 0);
        float a0 = ap[0], a1 = ap[4];
        for (int j = 0; j < S; j += 16) {
            loadt(tn, tt, j + 16);
            const float n0 = ap[(size_t)(j + 16) * 16], n1 = ap[(size_t)(j + 16) * 16 + 4];
            __builtin_amdgcn_sched_barrier(0);
            v4f d[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bu[t][0], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if constexpr (C > 4) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bu[t][1], d[t], 0, 0, 0);
            }
            // d[t][i] at lane (n, g) = wb[configuration 16 t + n][support j + 4 g + i]  ->  lane = configuration
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t f[4];
                frags(fu(d[0][i]), fu(d[1][i]), fu(d[2][i]), fu(d[3][i]), f);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float val = xb * tc[4 * g + i];
                    acc = fmaf(uf(f[g]), val, acc);
                }
            }
            a0 = n0; a1 = n1;
            PIPE_FENCE();
#pragma unroll
            for (int e = 0; e < 16; ++e) tc[e] = tn[e];
        }
    }
    out[(size_t)gw * 64 + lane] = acc;
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static int g_blocks = 0;
template <typename F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
static double max_rel(const std::vector<float>& a, const std::vector<float>& b, bool* bitwise) {
    double m = 0, scale = 0;
    *bitwise = std::memcmp(a.data(), b.data(), a.size() * 4) == 0;
    for (size_t i = 0; i < a.size(); ++i) scale = std::fmax(scale, std::fabs((double)a[i]));
    for (size_t i = 0; i < a.size(); ++i) m = std::fmax(m, std::fabs((double)a[i] - b[i]) / scale);
    return m;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int occ = argc > 1 ? atoi(argv[1]) : 4;   // waves per SIMD
    const int S = 2048, reps = 20;
    g_blocks = prop.multiProcessorCount * occ;      // 256-thread blocks: 4 waves each, one per SIMD
    const size_t nconf = (size_t)g_blocks * 256;
    printf("device %s  CUs=%d  clock=%.0f MHz  waves/SIMD=%d  S=%d rows per wave, %zu configurations\n", prop.gcnArchName,
           prop.multiProcessorCount, prop.clockRate / 1e3, occ, S, nconf);

    // support rows
    // wrows5 / wrows8: W[j][0..C-1] with t_j behind it at column C (the VALU forms read (W row, t_j) as one scalar row)
    std::vector<float> rows((size_t)(S + 64) * 16, 0.0f), wrows5(rows.size(), 0.0f), wrows8(rows.size(), 0.0f), tarr(S + 64, 0.0f);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (int j = 0; j < S; ++j) {
        for (int k = 0; k < D; ++k) rows[(size_t)j * 16 + k] = rnd() * 3.0f;
        rows[(size_t)j * 16 + 12] = rnd();
        tarr[j] = rows[(size_t)j * 16 + 12];
        for (int c = 0; c < 8; ++c) {
            const float wv = rnd();
            wrows8[(size_t)j * 16 + c] = wv;
            if (c < 5) wrows5[(size_t)j * 16 + c] = wv;
        }
        wrows5[(size_t)j * 16 + 5] = wrows8[(size_t)j * 16 + 8] = tarr[j];
    }
    // bf16 planes of the coordinates: [plane][S / 8][16][8]
    const size_t plane_stride = (size_t)(S / 8 + 8) * 16 * 8;
    std::vector<uint16_t> planes(3 * plane_stride, 0);
    for (int j = 0; j < S; ++j)
        for (int k = 0; k < D; ++k) {
            float c = rows[(size_t)j * 16 + k], r = c;
            for (int p = 0; p < 3; ++p) {
                uint32_t u;
                std::memcpy(&u, &r, 4);
                u &= 0xFFFF0000u;
                float h;
                std::memcpy(&h, &u, 4);
                planes[p * plane_stride + ((size_t)(j / 8) * 16 + k) * 8 + (j % 8)] = (uint16_t)(u >> 16);
                r -= h;
            }
        }
    float *d_rows, *d_wrows5, *d_wrows8, *d_t, *d_out, *d_ref;
    uint16_t* d_planes;
    hipMalloc(&d_rows, rows.size() * 4);
    hipMalloc(&d_wrows5, wrows5.size() * 4);
    hipMalloc(&d_wrows8, wrows8.size() * 4);
    hipMalloc(&d_t, tarr.size() * 4);
    hipMalloc(&d_planes, planes.size() * 2);
    hipMalloc(&d_out, nconf * 12 * 4);
    hipMalloc(&d_ref, nconf * 12 * 4);
    hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_wrows5, wrows5.data(), wrows5.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_wrows8, wrows8.data(), wrows8.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_t, tarr.data(), tarr.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_planes, planes.data(), planes.size() * 2, hipMemcpyHostToDevice);

    auto fetch = [&](float* p, size_t n) { std::vector<float> h(n); hipMemcpy(h.data(), p, n * 4, hipMemcpyDeviceToHost); return h; };
    auto report = [&](const char* what, const char* var, float ms, float ms_ref, double rel, bool bitwise) {
        const double ns = ms * 1e6 / ((double)S * occ);  // one SIMD works through occ waves x S rows
        printf("%-10s %-8s %9.3f ms   %6.2f ns = %5.1f cycles of a SIMD per wave-row   x%.2f of the VALU form   %s\n", what, var, ms, ns,
               ns * prop.clockRate / 1e6, ms / ms_ref,
               bitwise ? "bitwise equal" : (rel < 0 ? "" : (std::string("max rel err ") + std::to_string(rel)).c_str()));
    };
    {   // fold
        const size_t n = nconf * D;
        hipMemset(d_ref, 0, n * 4);
        const float t0 = time_ms([&] { fold_kernel<0><<<g_blocks, 256>>>(d_rows, d_t, d_planes, plane_stride, d_ref, S); }, reps);
        const auto ref = fetch(d_ref, n);
        report("fold D=12", "valu", t0, t0, -1, false);
        for (int v = 1; v <= 2; ++v) {
            hipMemset(d_out, 0, n * 4);
            const float t = v == 1 ? time_ms([&] { fold_kernel<1><<<g_blocks, 256>>>(d_rows, d_t, d_planes, plane_stride, d_out, S); }, reps)
                                   : time_ms([&] { fold_kernel<2><<<g_blocks, 256>>>(d_rows, d_t, d_planes, plane_stride, d_out, S); }, reps);
            bool bw;
            const double rel = max_rel(ref, fetch(d_out, n), &bw);
            report("fold D=12", v == 1 ? "mfma32" : "bf16x3", t, t0, rel, bw);
        }
    }
    {   // d2
        // A operand of the split form: per 16 supports three K = 32 chunks; chunk c, K slot (half, kk): term 2 c + half, feature kk
        //   terms: 0 hi.hi  1 hi.mid  2 mid.hi  3 hi.lo  4 lo.hi  5 mid.mid   (s plane: hi mid hi lo hi mid)
        const int splane_of_term[6] = {0, 1, 0, 2, 0, 1};
        std::vector<float> ssarr(S + 64, 0.0f);
        std::vector<uint16_t> ap((size_t)(S / 16 + 2) * 3 * 16 * 4 * 8, 0);
        for (int j = 0; j < S; ++j) {
            uint16_t pl[3][16] = {};
            double s2 = 0;
            for (int k = 0; k < D; ++k) {
                float r = rows[(size_t)j * 16 + k];
                s2 += (double)r * r;
                for (int p = 0; p < 3; ++p) {
                    uint32_t u;
                    std::memcpy(&u, &r, 4);
                    u &= 0xFFFF0000u;
                    float h;
                    std::memcpy(&h, &u, 4);
                    pl[p][k] = (uint16_t)(u >> 16);
                    r -= h;
                }
            }
            ssarr[j] = (float)s2;
            rows[(size_t)j * 16 + 13] = (float)s2;
            for (int c = 0; c < 3; ++c)
                for (int kq = 0; kq < 4; ++kq)
                    for (int e = 0; e < 8; ++e) {
                        const int slot = 8 * kq + e, term = 2 * c + slot / 16, kk = slot % 16;
                        ap[((((size_t)(j / 16) * 3 + c) * 16 + (j % 16)) * 4 + kq) * 8 + e] = pl[splane_of_term[term]][kk];
                    }
        }
        float* d_ss;
        uint16_t* d_ap;
        hipMalloc(&d_ss, ssarr.size() * 4);
        hipMalloc(&d_ap, ap.size() * 2);
        hipMemcpy(d_ss, ssarr.data(), ssarr.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_ap, ap.data(), ap.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
        hipMemset(d_ref, 0, nconf * 4);
        hipMemset(d_out, 0, nconf * 4);
        const float t0 = time_ms([&] { d2_kernel<0><<<g_blocks, 256>>>(d_rows, d_ss, d_ap, d_ref, S); }, reps);
        const float t1 = time_ms([&] { d2_kernel<1><<<g_blocks, 256>>>(d_rows, d_ss, d_ap, d_out, S); }, reps);
        bool bw;
        const double rel = max_rel(fetch(d_ref, nconf), fetch(d_out, nconf), &bw);
        report("d2 D=12", "valu", t0, t0, -1, false);
        report("d2 D=12", "bf16x3", t1, t0, rel, bw);
        {   // the whole pair body
            const size_t n = nconf * 16;
            float *d_o0, *d_o1;
            hipMalloc(&d_o0, n * 4);
            hipMalloc(&d_o1, n * 4);
            hipMemset(d_o0, 0, n * 4);
            hipMemset(d_o1, 0, n * 4);
            const float x0 = time_ms([&] { xf_kernel<0><<<g_blocks, 256>>>(d_rows, d_ss, d_ap, d_o0, S); }, reps);
            const float x1 = time_ms([&] { xf_kernel<1><<<g_blocks, 256>>>(d_rows, d_ss, d_ap, d_o1, S); }, reps);
            const double relx = max_rel(fetch(d_o0, n), fetch(d_o1, n), &bw);
            report("pair body", "valu", x0, x0, -1, false);
            report("pair body", "d2:bf16x3", x1, x0, relx, bw);
            hipFree(d_o0);
            hipFree(d_o1);
        }
    }
    auto kw = [&](auto cc) {
        constexpr int C = decltype(cc)::value;
        char name[32];
        const size_t n = nconf * 8;
        float* d_wrows = C == 5 ? d_wrows5 : d_wrows8;
        hipMemset(d_ref, 0, n * 4);
        hipMemset(d_out, 0, n * 4);
        const float t0 = time_ms([&] { kw_kernel<C, 0><<<g_blocks, 256>>>(d_wrows, d_t, d_ref, S); }, reps);
        const float t1 = time_ms([&] { kw_kernel<C, 1><<<g_blocks, 256>>>(d_wrows, d_t, d_out, S); }, reps);
        bool bw;
        const double rel = max_rel(fetch(d_ref, n), fetch(d_out, n), &bw);
        snprintf(name, sizeof name, "kw C=%d", C);
        report(name, "valu", t0, t0, -1, false);
        report(name, "mfma32", t1, t0, rel, bw);
        hipMemset(d_ref, 0, nconf * 4);
        hipMemset(d_out, 0, nconf * 4);
        const float g0 = time_ms([&] { gwt_kernel<C, 0><<<g_blocks, 256>>>(d_wrows, d_t, d_ref, S); }, reps);
        const float g1 = time_ms([&] { gwt_kernel<C, 1><<<g_blocks, 256>>>(d_wrows, d_t, d_out, S); }, reps);
        const double relg = max_rel(fetch(d_ref, nconf), fetch(d_out, nconf), &bw);
        snprintf(name, sizeof name, "gwt C=%d", C);
        report(name, "valu", g0, g0, -1, false);
        report(name, "mfma32", g1, g0, relg, bw);
    };
    kw(std::integral_constant<int, 5>{});
    kw(std::integral_constant<int, 8>{});
    return 0;
}
