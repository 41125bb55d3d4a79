// hess_kernel.hip — analytic second derivatives of the score: dcx_score_hess.
//
// Replaces the double backward the reference runs through dist_est for trust-constr's constraint Hessian
// (diffco/optim.py:380-391, torch.autograd.functional.hessian over DiffCo.score / poly_score).  For
//     f(q) = sum_c up_c * sum_j W[j, c] K(|x(q) - s_j|^2),       x = T(q) (the FK control points)
// row i of the Hessian is the directional derivative of the gradient along e_i.  One lane takes one
// (configuration, direction) pair and pushes a value + tangent pair (fk_device.h `Dual`) through the SAME three phases
// as the fused gradient kernel:
//   1. forward FK in duals:           x,  dx = J e_i
//   2. the sweep over the supports:   gX  = sum_j c_j delta_j,                 c_j = w_j g(d2_j), delta_j = x - s_j
//                                     dgX = sum_j c_j dx + e_j delta_j,        e_j = w_j h(d2_j) (delta_j . dx)
//      with g = 2 K'(d2) (the gradient coefficient of score_kernel.h kernel_eval) and h = 2 dg/dd2 in closed form
//   3. the reverse FK sweep in duals: the tangent of J^T gX is  J^T dgX + (dJ/dq_i)^T gX  = H[i, :]
// so the second derivatives of the transform come from forward-mode differentiation of the code that produces the first
// ones — no finite differences anywhere (the previous route took central differences of the analytic gradient: 2e-3 of
// the float64 Hessian; this one is fp32 round-off, ~1e-6).
//
// This is a callers'-side kernel (hundreds to thousands of dense-path points per trust-constr iteration), written for
// exactness and generality — any feature width, every kernel function and transform: for D <= 32 x, dx and the sums sit in
// registers, beyond that in LDS (4 D ds operations per pair and lane).
//
// Round 6, the moments form (hess_moments_kernel below) for the narrow compiled widths (D <= 16) and B >= 1024: phase 2 does not
// need the direction - dgX = (sum_j c_j) dx + (sum_j e'_j delta_j delta_j^T) dx - so one lane per CONFIGURATION sweeps the
// supports for gX, sum c and the symmetric D x D matrix (81 VALU instructions per configuration and pair at D = 12 instead of
// 7 x 46), and the (configuration, direction) lanes of this kernel read M dx from it: B = 8192 288 -> 77 us, B = 65536
// 2010 -> 510 us on the headline model (profiles/r06_hess_moments.txt).
#include "dcx_internal.h"

#include <algorithm>
#include <cstdlib>

namespace dcx {

struct HessArgs {
    const float* rows;      // model rows [S][RS]
    const FkProg* fk;
    const float* q;         // [B][dof]
    const float* upstream;  // [B][C] or null (= all ones)
    float* grad;            // [B][dof] or null
    float* hess;            // [B][dof][dof]
    int64_t n_lanes;        // B * dof
    int32_t S, D, C, RS, w_off, wsum_off;   // C: the compiled class count (row layout)
    int32_t c_out;                          // the caller's class count (row stride of upstream)
    int32_t dof, d_fk, frame_floats;
    int32_t kind, kf;
    int32_t fk_dh;          // 1: a DH arm - the chain and its reverse sweep read the FK program with scalar loads (fk_*_dh_k)
    float kp0, kp1;
    int32_t s_chunk;
    int32_t m_skew;           // moments form, 12-wave blocks: per mille of the block's rows for wave groups 0 and 1 (w0 | w1 << 10; 0 = equal slices)
    // supports split across gridDim.y blocks per 64-lane tile (small batches): block y sweeps [y * s_super, (y + 1) * s_super),
    // leaves its partial sums in `part`, and the block that arrives last at the tile's counter adds them in y order and
    // runs the reverse sweep
    int32_t ys, s_super, counter_stride;
    Dual* part;               // [tile][ys][D][64]
    unsigned int* counters;   // [tile * counter_stride], zero between launches
    // transforms whose frames do not fit the LDS in duals (the 23-joint iiwa7 + Allegro tree): the frames of block b live
    // in global memory at frames_g + b * frame_floats * 64 (same column layout, coalesced 512-byte columns)
    Dual* frames_g;
    int64_t lane0;            // first (configuration, direction) pair of this launch (chunked launches)
    // the moments form (hess_moments_kernel below): the sweep's sums per CONFIGURATION, [tile][m_ys][m_nacc][64] floats; the
    // (configuration, direction) lanes of score_hess_kernel read them instead of sweeping
    float* mom;
    int64_t mom_b0;           // first configuration the buffer holds
    int32_t m_ys, m_nacc;
    int32_t o_m;              // (floats) the block's configurations' folded sums: [configuration of the block][m_nacc]
    // LDS plan, in Dual elements (8 bytes)
    int32_t o_q, o_f, o_x, o_acc, o_fk_floats, prog_floats;
};

// value, gradient coefficient g (dK/dx = g * delta) and h = 2 dg/dd2 (d2K/dx2 = g I + h delta delta^T)
// KFT: the kernel function fixed at compile time (KF_POLY1 / KF_RQ2: the two the sweeps specialise) or KF_GEN = decided
// per launch.  With everything behind a run-time switch the compiler if-converted parts of the general branch (powf / logf)
// into the specialised ones' path: ~170 VALU instructions per pair where ~60 do the work (round 3, cycle stamps).
template <int KFT>
__device__ __forceinline__ void kernel_eval_h(const HessArgs& a, float d2, float& g, float& h) {
    if (KFT == KF_POLY1 || (KFT == KF_GEN && a.kf == KF_POLY1)) {  // r / eps with 1 / eps in the row weights: g = 1 / r, h = -1 / r^3 (v_rsq, like the sweep's kernel_eval)
        const float ri = __builtin_amdgcn_rsqf(fmaxf(d2, 1e-30f));
        const bool on = d2 >= 1e-20f;  // on a support the kernel is not twice differentiable: the pair contributes nothing
        g = on ? ri : 0.0f;
        h = on ? -(ri * ri) * ri : 0.0f;
        return;
    }
    if (KFT == KF_RQ2 || (KFT == KF_GEN && a.kf == KF_RQ2)) {
        // (1 + gamma/2 d2)^-2 with the constants folded as in the sweeps (score_kernel.h sweep_eval): the rows of an RQ2 model
        // carry w (2/gamma)^2 and a.kp0 = 2/gamma, so with u = 1 / (d2 + 2/gamma):  g w = -2 gamma (1 + gamma/2 d2)^-3 w =
        // -4 u^3 w',  h w = 6 gamma^2 (1 + gamma/2 d2)^-4 w = 24 u^4 w'
        const float u = __builtin_amdgcn_rcpf(d2 + a.kp0);
        const float u3 = (u * u) * u;
        g = -4.0f * u3;
        h = 24.0f * (u3 * u);
        return;
    }
    if constexpr (KFT != KF_GEN) {
        return;  // (not reached: the two specialised functions returned above)
    } else
    if (a.kf == KF_RQ2 || (a.kf == KF_GEN && a.kind == DCX_K_RQ)) {
        // K = t^-p, t = 1 + gamma/p d2:  g = -2 gamma t^(-p-1),  h = 4 gamma^2 (p+1)/p t^(-p-2)
        const float p = (a.kf == KF_RQ2) ? 2.0f : a.kp1;
        const float t = fmaf(a.kp0 / p, d2, 1.0f);
        const float u = 1.0f / t;
        const float tp1 = (p == 2.0f) ? u * u * u : powf(t, -p - 1.0f);
        g = -2.0f * a.kp0 * tp1;
        h = 4.0f * a.kp0 * a.kp0 * (p + 1.0f) / p * tp1 / t;
    } else if (a.kf == KF_POLY1 || a.kind == DCX_K_POLY) {
        // K = r^k / eps (k odd) or r^k log r / eps (k even).  KF_POLY1 rows carry 1/eps in the weights.
        const int k = (a.kf == KF_POLY1) ? 1 : (int)a.kp0;
        const float ie = (a.kf == KF_POLY1) ? 1.0f : 1.0f / a.kp1;
        if (d2 < 1e-20f) {  // on a support the kernel is not twice differentiable (k <= 2); the pair contributes nothing
            g = 0.0f;
            h = 0.0f;
            return;
        }
        const float r = sqrtf(d2);
        const float i2 = 1.0f / d2;
        float rk4 = i2 * i2;  // r^(k-4)
        for (int i = 0; i < k; ++i) rk4 *= r;
        if (k & 1) {
            g = (float)k * rk4 * d2 * ie;
            h = (float)(k * (k - 2)) * rk4 * ie;
        } else {
            const float lg = logf(r);
            g = rk4 * d2 * fmaf((float)k, lg, 1.0f) * ie;
            h = rk4 * ((float)(k - 2) * fmaf((float)k, lg, 1.0f) + (float)k) * ie;
        }
    } else {  // DCX_K_MQ: K = sqrt(1 + d2 / eps^2)
        const float ie2 = 1.0f / (a.kp0 * a.kp0);
        const float rv = 1.0f / sqrtf(fmaf(d2, ie2, 1.0f));
        g = rv * ie2;
        h = -rv * rv * rv * ie2 * ie2;
    }
}

__device__ __forceinline__ float pair_weight(const HessArgs& a, const float* r, const float* up) {
    if (!up) return r[a.C > 1 ? a.wsum_off : a.w_off];
    float w = 0.0f;
    for (int c = 0; c < a.c_out; ++c) w = fmaf(up[c], r[a.w_off + c], w);
    return w;
}

// supports [j0, j1) of one wave, any width: x, dx and the sums stay in LDS
template <int KFT>
__device__ __forceinline__ void sweep_hess_lds(const HessArgs& a, const Dual* sX, Dual* sAcc, const float* up, int j0, int j1) {
    for (int j = j0; j < j1; ++j) {
        const float* r = a.rows + (size_t)j * a.RS;  // uniform address: scalar loads
        float d2 = 0.0f, dd = 0.0f;
        for (int k = 0; k < a.D; ++k) {
            const Dual xk = sX[k * 64];
            const float dl = xk.v - r[k];
            d2 = fmaf(dl, dl, d2);
            dd = fmaf(dl, xk.d, dd);
        }
        const float w = pair_weight(a, r, up);
        float g, h;
        kernel_eval_h<KFT>(a, d2, g, h);
        const float cf = w * g, ef = w * h * dd;
        for (int k = 0; k < a.D; ++k) {
            const Dual xk = sX[k * 64];
            const float dl = xk.v - r[k];
            Dual acc = sAcc[k * 64];
            acc.v = fmaf(cf, dl, acc.v);
            acc.d = fmaf(cf, xk.d, fmaf(ef, dl, acc.d));
            sAcc[k * 64] = acc;
        }
    }
}

// the same for one of the compiled widths D <= 32 with x, dx and the sums in registers as packed pairs (v_pk_add /
// v_pk_fma: 3 D instructions per pair for the differences, d2, delta . dx and the two accumulations) and the rows read
// through the scalar cache (constant address space, like the sweep of score_kernel.h)
template <int D, int KFT>
__device__ __forceinline__ void sweep_hess_regs(const HessArgs& a, const Dual* sX, Dual* sAcc, const float* up, int j0, int j1) {
    constexpr int NP = D / 2;
    constexpr bool ODD = (D & 1) != 0;
    v2f xv[NP + 1], xd[NP + 1], av[NP + 1], ad[NP + 1];  // [NP] holds the odd last feature in .x
#pragma unroll
    for (int k = 0; k <= NP; ++k) {
        const bool two = k < NP;
        const Dual t0 = (two || ODD) ? sX[(2 * k) * 64] : Dual(0.0f, 0.0f);
        const Dual t1 = two ? sX[(2 * k + 1) * 64] : Dual(0.0f, 0.0f);
        xv[k] = v2f{t0.v, t1.v}; xd[k] = v2f{t0.d, t1.d}; av[k] = v2f{0.0f, 0.0f}; ad[k] = v2f{0.0f, 0.0f};
    }
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)a.rows;
    const int C = a.C, w_off = a.w_off;
    auto weight = [&](cfloat_ptr r) __attribute__((always_inline)) {
        if (!up) return r[C > 1 ? a.wsum_off : w_off];
        float w = 0.0f;
        for (int c = 0; c < a.c_out; ++c) w = fmaf(up[c], r[w_off + c], w);
        return w;
    };
    constexpr int NV = NP + (ODD ? 1 : 0);
    // one support row: its D coordinates and its weight, read one row ahead of their use (two scalar-register buffers;
    // with one to four waves per SIMD nothing else hides the scalar-load latency)
    auto load_row = [&](float (&dst)[D + 1], int j) __attribute__((always_inline)) {
        cfloat_ptr r = rows + (size_t)j * a.RS;
#pragma unroll
        for (int k = 0; k < D; ++k) dst[k] = r[k];
        dst[D] = up ? 0.0f : r[C > 1 ? a.wsum_off : w_off];
    };
    auto pair = [&](const float (&r)[D + 1], int j) __attribute__((always_inline)) {
        v2f dl[NV], s2a = {0.0f, 0.0f}, s2b = {0.0f, 0.0f}, sda = {0.0f, 0.0f}, sdb = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const v2f rv = (k < NP) ? v2f{r[2 * k], r[2 * k + 1]} : v2f{r[D - 1], 0.0f};
            dl[k] = xv[k] - rv;
            if (k & 1) { s2b = __builtin_elementwise_fma(dl[k], dl[k], s2b); sdb = __builtin_elementwise_fma(dl[k], xd[k], sdb); }
            else       { s2a = __builtin_elementwise_fma(dl[k], dl[k], s2a); sda = __builtin_elementwise_fma(dl[k], xd[k], sda); }
        }
        s2a += s2b;
        sda += sdb;
        const float w = up ? weight(rows + (size_t)j * a.RS) : r[D];
        float g, h;
        kernel_eval_h<KFT>(a, s2a.x + s2a.y, g, h);
        const float cf = w * g, ef = w * h * (sda.x + sda.y);
        const v2f c2 = {cf, cf}, e2 = {ef, ef};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            av[k] = __builtin_elementwise_fma(c2, dl[k], av[k]);
            ad[k] = __builtin_elementwise_fma(c2, xd[k], __builtin_elementwise_fma(e2, dl[k], ad[k]));
        }
    };
    if (j0 < j1) {
        float ra[D + 1], rb[D + 1];
        const int jl = j1 - 1;
        load_row(ra, j0);
        for (int j = j0; j < j1; j += 2) {
            load_row(rb, j + 1 < j1 ? j + 1 : jl);
            __builtin_amdgcn_sched_barrier(0);
            pair(ra, j);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load_row(ra, j + 2 < j1 ? j + 2 : jl);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < j1) pair(rb, j + 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        sAcc[(2 * k) * 64] = Dual(av[k].x, ad[k].x);
        if (k < NP) sAcc[(2 * k + 1) * 64] = Dual(av[k].y, ad[k].y);
    }
}

// ---- the moments form (round 6) ---------------------------------------------------------------------------------------------
// The (configuration, direction) lanes above repeat the distance, the kernel function and g X for every one of the dof
// directions: 46 VALU instructions per lane and pair at D = 12, 322 per configuration and pair.  But the tangent sum is linear
// in dx:    dgX = (sum_j c_j) dx + (sum_j e'_j delta_j delta_j^T) dx,     e'_j = w_j h(d2_j),
// so ONE lane per configuration can sweep the supports for gX [D], sum c [1] and the symmetric D x D matrix M [D (D + 1) / 2]
// - 81 instructions per configuration and pair at D = 12 (6 differences, 6 + 3 distance, ~8 kernel function, 6 t = e' delta,
// 42 packed M += t_k (delta_2p, delta_2p+1) on the pairs p >= k / 2, 6 gX, 1) - and the direction lanes take M dx from it.
// Accumulators: D + 1 + (D / 2)(D / 2 + 1) 2 floats per lane (97 at D = 12, 171 at D = 16): three waves per SIMD at
// D <= 12 (153 VGPRs), two at D = 16.
// Layout of a lane's sums: [0, D) gX, [D] sum c, then row k of M as the packed pairs p = k / 2 .. D / 2 - 1 (entry (k, 2 p) and
// (k, 2 p + 1); for odd k the first one is the lower-triangle duplicate (k, k - 1), never read).
// waves per block: three per SIMD where the sums leave room for it (153 VGPRs at D = 12), else two
__host__ __device__ constexpr int hess_moments_waves(int D, int kft) { return (D <= 12 && kft != KF_GEN) ? 12 : 8; }

template <int D>
struct HessMom {
    static constexpr int NP = D / 2, NM = NP * (NP + 1), NACC = D + 1 + 2 * NM;
    static constexpr int CH = 24;   // accumulators per fold pass (12 waves x 24 x 256 B = 72 KB of LDS)
    static __host__ __device__ constexpr int row_off(int k) {  // first packed pair of M's row k
        int o = 0;
        for (int i = 0; i < k; ++i) o += NP - i / 2;
        return o;
    }
};

// UPW: 0 = no upstream (the row's own weight, or its class sum), 1 / 8 = the caller's upstream over one / up to eight class weights
template <int D, int KFT, int UPW>
__global__ __launch_bounds__(64 * hess_moments_waves(D, KFT)) void hess_moments_kernel(const HessArgs a) {
    using HM = HessMom<D>;
    constexpr int NP = HM::NP, NM = HM::NM, NACC = HM::NACC, CH = HM::CH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int dof = a.dof;
    int64_t b = a.lane0 + (int64_t)blockIdx.x * 64 + lane;   // (lane0, n_lanes: configurations here)
    if (b >= a.n_lanes) b = a.n_lanes - 1;                   // surplus lanes repeat the last configuration; nobody reads their sums
    float* sQ = smem + a.o_q + lane * dof;
    float* sF = smem + a.o_f + lane;
    float* sX = smem + a.o_x + lane;
    float* sR = smem + a.o_acc;
    const fk_cptr fk = stage_fk_prog(a.fk, smem + a.o_fk_floats, threadIdx.x, blockDim.x);
    if (wave == 0)
        for (int k = 0; k < dof; ++k) sQ[k] = a.q[b * dof + k];
    __syncthreads();
    fk_forward_trig<float>(fk, sQ, sF, wave, nw);
    __syncthreads();
    if (wave == 0) {
        if (a.fk_dh) fk_forward_chain_dh_k<float>((fk_kptr)(uintptr_t)a.fk, sX, sF);
        else fk_forward_chain<float>(fk, sQ, sX, sF);
        for (int k = a.d_fk; k < D; ++k) sX[k * 64] = 0.0f;
    }
    __syncthreads();
    const int ybase = (int)blockIdx.y * a.s_super;
    const int yend = (ybase + a.s_super < a.S) ? ybase + a.s_super : a.S;
    // (a SIMD issues oldest-first - score_kernel.h wave_slice: the three wave groups of a 12-wave block take unequal shares of its rows)
    int j0, j1;
    if (a.m_skew > 0 && nw == 12) {
        const int tot = 12 * a.s_chunk;
        const int l0 = tot * (a.m_skew & 1023) / 4000, l1 = tot * ((a.m_skew >> 10) & 1023) / 4000;
        const int l2 = 3 * a.s_chunk - l0 - l1;
        const int g = wave >> 2, q = wave & 3;
        const int len = g == 0 ? l0 : g == 1 ? l1 : (l2 > 0 ? l2 : 0);
        const int start = 4 * (g == 0 ? 0 : g == 1 ? l0 : l0 + l1) + q * len;
        j0 = (ybase + start < yend) ? ybase + start : yend;
        j1 = (j0 + len < yend) ? j0 + len : yend;
    } else {
        j0 = (ybase + wave * a.s_chunk < yend) ? ybase + wave * a.s_chunk : yend;
        j1 = (j0 + a.s_chunk < yend) ? j0 + a.s_chunk : yend;
    }

    v2f xv[NP], gx[NP], mm[NM];
    float cs = 0.0f;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        xv[k] = v2f{sX[(2 * k) * 64], sX[(2 * k + 1) * 64]};
        gx[k] = v2f{0.0f, 0.0f};
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) mm[i] = v2f{0.0f, 0.0f};
    // the caller's upstream: this configuration's row, zero beyond the caller's classes; the weights it multiplies are read from
    // loop-invariant offsets (a class the row does not have re-reads the first one: finite, times zero)
    constexpr int CW = UPW > 0 ? UPW : 1;
    float upv[CW];
    int woff[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        upv[c] = (UPW > 0 && c < a.c_out) ? a.upstream[b * a.c_out + c] : 0.0f;
        woff[c] = UPW > 0 ? a.w_off + (c < a.C ? c : 0) : (a.C > 1 ? a.wsum_off : a.w_off);
    }
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)a.rows;
    // one support row - D coordinates and its weight(s) - read one row ahead of its use into scalar registers
    auto load_row = [&](float (&dst)[D + CW], int j) __attribute__((always_inline)) {
        cfloat_ptr r = rows + (size_t)j * a.RS;
#pragma unroll
        for (int k = 0; k < D; ++k) dst[k] = r[k];
#pragma unroll
        for (int c = 0; c < CW; ++c) dst[D + c] = r[woff[c]];
    };
    // A pair in two stages - A: differences, distance, kernel function, coefficients (a dependent chain of ~25 instructions);
    // B: the 54 independent accumulations - so that stage A of row j + 1 sits in the same scheduling region as stage B of row j
    // and fills its latencies (218.6 -> 206.1 us per 32768 configurations; profiles/r06_hess_moments.txt)
    struct StageA {
        v2f dl[NP];
        float cf, ef;
    };
    auto stage_a = [&](const float (&r)[D + CW], StageA& o, float on) __attribute__((always_inline)) {
        v2f s2a = {0.0f, 0.0f}, s2b = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            o.dl[k] = xv[k] - v2f{r[2 * k], r[2 * k + 1]};
            if (k & 1) s2b = __builtin_elementwise_fma(o.dl[k], o.dl[k], s2b);
            else       s2a = __builtin_elementwise_fma(o.dl[k], o.dl[k], s2a);
        }
        if (NP > 1) s2a += s2b;
        float w = r[D] * on;   // (on = 0: a row past the slice's end, loaded again and left out)
        if constexpr (UPW > 0) {
            w *= upv[0];
#pragma unroll
            for (int c = 1; c < CW; ++c) w = fmaf(upv[c] * on, r[D + c], w);
        }
        float g, h;
        kernel_eval_h<KFT>(a, s2a.x + s2a.y, g, h);
        o.cf = w * g;
        o.ef = w * h;
    };
    auto stage_b = [&](const StageA& o) __attribute__((always_inline)) {
        cs += o.cf;
        const v2f c2 = {o.cf, o.cf}, e2 = {o.ef, o.ef};
        v2f t[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            gx[k] = __builtin_elementwise_fma(c2, o.dl[k], gx[k]);
            t[k] = e2 * o.dl[k];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float tk = (k & 1) ? t[k / 2].y : t[k / 2].x;
            const v2f t2 = {tk, tk};
#pragma unroll
            for (int p = k / 2; p < NP; ++p) {
                v2f& m = mm[HM::row_off(k) + p - k / 2];
                m = __builtin_elementwise_fma(t2, o.dl[p], m);
            }
        }
    };
    if (j0 < j1) {
        float ra[D + CW], rb[D + CW];
        StageA sa, sb;
        const int jl = j1 - 1;
        load_row(ra, j0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        stage_a(ra, sa, 1.0f);
        load_row(ra, j0 + 1 < j1 ? j0 + 1 : jl);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        for (int j = j0; j < j1; j += 2) {
            // here: sa = stage A of row j; ra holds row j + 1
            load_row(rb, j + 2 < j1 ? j + 2 : jl);
            __builtin_amdgcn_sched_barrier(0);
            stage_a(ra, sb, j + 1 < j1 ? 1.0f : 0.0f);
            stage_b(sa);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load_row(ra, j + 3 < j1 ? j + 3 : jl);
            __builtin_amdgcn_sched_barrier(0);
            stage_a(rb, sa, j + 2 < j1 ? 1.0f : 0.0f);
            stage_b(sb);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- fold over the block's waves, CH sums per pass through the LDS: sum i of a pass belongs to wave i % nw, which adds the
    //      waves' rows in wave order and writes the total to this (tile, y)'s row of the moments buffer (coalesced, 256 B per sum)
    float* out = a.mom + (((size_t)blockIdx.x * a.m_ys + blockIdx.y) * NACC) * 64 + lane;
    auto sum_at = [&](int i) __attribute__((always_inline)) -> float {   // (i is a compile-time constant after unrolling)
        if (i < D) return (i & 1) ? gx[i / 2].y : gx[i / 2].x;
        if (i == D) return cs;
        const int m = i - D - 1;
        return (m & 1) ? mm[m / 2].y : mm[m / 2].x;
    };
#pragma unroll
    for (int c0 = 0; c0 < NACC; c0 += CH) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (c0 + i < NACC) sR[((size_t)wave * CH + i) * 64 + lane] = sum_at(c0 + i);
        __syncthreads();
        for (int i = wave; i < CH && c0 + i < NACC; i += nw) {
            float tot = sR[(size_t)i * 64 + lane];
            for (int w = 1; w < nw; ++w) tot += sR[((size_t)w * CH + i) * 64 + lane];
            out[(size_t)(c0 + i) * 64] = tot;
        }
        __syncthreads();
    }
}

// SMALL: compiled widths <= 16 only, up to 16 waves per block (128 VGPRs); otherwise every width, up to 8 waves
template <bool SMALL>
__global__ __launch_bounds__(SMALL ? 1024 : 512) void score_hess_kernel(const HessArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Dual* sd = reinterpret_cast<Dual*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int dof = a.dof;
    int64_t gl = a.lane0 + (int64_t)blockIdx.x * 64 + lane;
    const bool live = gl < a.n_lanes;
    if (!live) gl = a.n_lanes - 1;  // surplus lanes repeat the last pair and store nothing
    const int64_t b = gl / dof;
    const int dir = (int)(gl - b * dof);
    Dual* sQ = sd + a.o_q + lane * dof;  // this lane's row
    Dual* sF = a.frames_g ? a.frames_g + (size_t)blockIdx.x * a.frame_floats * 64 + lane  // columns, stride 64
                          : sd + a.o_f + lane;
    Dual* sX = sd + a.o_x + lane;
    Dual* sAcc = sd + a.o_acc + (size_t)wave * a.D * 64 + lane;

    const fk_cptr fk = stage_fk_prog(a.fk, smem + a.o_fk_floats, threadIdx.x, blockDim.x);
    if (wave == 0)
        for (int k = 0; k < dof; ++k) sQ[k] = Dual(a.q[b * dof + k], k == dir ? 1.0f : 0.0f);
    __syncthreads();
    fk_forward_trig<Dual>(fk, sQ, sF, wave, nw);
    __syncthreads();
    if (wave == 0) {
        if (a.fk_dh) fk_forward_chain_dh_k<Dual>((fk_kptr)(uintptr_t)a.fk, sX, sF);
        else fk_forward_chain<Dual>(fk, sQ, sX, sF);
        for (int k = a.d_fk; k < a.D; ++k) sX[k * 64] = Dual(0.0f, 0.0f);  // zero padding up to the compiled width
    }
    for (int k = 0; k < a.D; ++k) sAcc[k * 64] = Dual(0.0f, 0.0f);
    __syncthreads();

    if (a.mom) {
        // ---- the moments form: this lane's configuration was swept by hess_moments_kernel; gX as it stands, the tangent sum
        //      = (sum c) dx + M dx from the symmetric matrix's packed rows (layout: HessMom), the m_ys partial rows added in y order
        // (a single-wave block.)  First the fold over y, spread over the lanes: the block's configurations (a configuration's
        // directions may straddle two blocks: both fold it) x their sums are dealt out to the 64 lanes - all (at most 16) partial
        // rows of a sum loaded together, added in y order - and left in the LDS for the configurations' lanes
        const int nacc = a.m_nacc, mys = a.m_ys;
        const int64_t g0 = a.lane0 + (int64_t)blockIdx.x * 64;
        const int64_t b_first = g0 / dof;
        const int64_t b_last = (g0 + 63 < a.n_lanes ? g0 + 63 : a.n_lanes - 1) / dof;
        const int items = (int)(b_last - b_first + 1) * nacc;
        float* sMb = smem + a.o_m;
        auto fold_y = [&](auto ny) __attribute__((always_inline)) {
            constexpr int NY = decltype(ny)::value;   // partial rows loaded together (>= m_ys)
            for (int w = lane; w < items; w += 64) {
                const int ci = w / nacc, i = w - ci * nacc;
                const int64_t rel = b_first + ci - a.mom_b0;
                const float* mp = a.mom + ((size_t)(rel >> 6) * mys * nacc) * 64 + (rel & 63);
                float v[NY];
#pragma unroll
                for (int y = 0; y < NY; ++y) v[y] = mp[((size_t)(y < mys ? y : mys - 1) * nacc + i) * 64];
                float t = v[0];
#pragma unroll
                for (int y = 1; y < NY; ++y) t += (y < mys) ? v[y] : 0.0f;
                sMb[w] = t;
            }
        };
        if (mys == 1) fold_y(std::integral_constant<int, 1>{});
        else if (mys <= 4) fold_y(std::integral_constant<int, 4>{});
        else fold_y(std::integral_constant<int, 16>{});
        const float* sM = sMb + (size_t)(b - b_first) * nacc;
        __syncthreads();
        // gX as it stands; the tangent sum = (sum c) dx + M dx, in registers for the compiled widths
        auto apply = [&](auto width) __attribute__((always_inline)) {
            constexpr int W = decltype(width)::value;
            float dx[W], ad[W];
            const float csum = sM[W];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                dx[k] = sX[k * 64].d;
                ad[k] = csum * dx[k];
            }
            int idx = W + 1;
#pragma unroll
            for (int k = 0; k < W; ++k) {
#pragma unroll
                for (int p = k / 2; p < W / 2; ++p, idx += 2) {
                    const int l0 = 2 * p, l1 = l0 + 1;
                    if (l0 >= k) {
                        const float m0 = sM[idx];
                        ad[k] = fmaf(m0, dx[l0], ad[k]);
                        if (l0 != k) ad[l0] = fmaf(m0, dx[k], ad[l0]);
                    }
                    const float m1 = sM[idx + 1];
                    ad[k] = fmaf(m1, dx[l1], ad[k]);
                    if (l1 != k) ad[l1] = fmaf(m1, dx[k], ad[l1]);
                }
            }
#pragma unroll
            for (int k = 0; k < W; ++k) sAcc[k * 64] = Dual(sM[k], ad[k]);
        };
        switch (a.D) {
            case 2: apply(std::integral_constant<int, 2>{}); break;
            case 4: apply(std::integral_constant<int, 4>{}); break;
            case 6: apply(std::integral_constant<int, 6>{}); break;
            case 8: apply(std::integral_constant<int, 8>{}); break;
            case 12: apply(std::integral_constant<int, 12>{}); break;
            case 16: apply(std::integral_constant<int, 16>{}); break;
            default: break;   // (the launcher forms the sums for these widths only)
        }
    } else {
    // ---- the sweep: this wave's slice of this block's supports ----
    const int ybase = (int)blockIdx.y * a.s_super;
    const int yend = (ybase + a.s_super < a.S) ? ybase + a.s_super : a.S;
    const int j0 = (ybase + wave * a.s_chunk < yend) ? ybase + wave * a.s_chunk : yend;
    const int j1 = (j0 + a.s_chunk < yend) ? j0 + a.s_chunk : yend;
    const float* up = a.upstream ? a.upstream + b * a.c_out : nullptr;
#define DCX_HESS_CASE(W) case W: sweep_hess_regs<W, KFT>(a, sX, sAcc, up, j0, j1); break;
    auto sweep = [&](auto kft) __attribute__((always_inline)) {
        constexpr int KFT = decltype(kft)::value;
        if constexpr (SMALL) {
            switch (a.D) {  // a.D is one of the compiled widths (dcx_internal.h kTemplateD)
                DCX_HESS_CASE(2) DCX_HESS_CASE(4) DCX_HESS_CASE(6) DCX_HESS_CASE(8) DCX_HESS_CASE(12) DCX_HESS_CASE(16)
                default: break;
            }
        } else {
            switch (a.D) {
                DCX_HESS_CASE(18) DCX_HESS_CASE(21) DCX_HESS_CASE(24) DCX_HESS_CASE(27) DCX_HESS_CASE(30) DCX_HESS_CASE(32)
                default: sweep_hess_lds<KFT>(a, sX, sAcc, up, j0, j1);
            }
        }
    };
    if (a.kf == KF_POLY1) sweep(std::integral_constant<int, KF_POLY1>{});
    else if (a.kf == KF_RQ2) sweep(std::integral_constant<int, KF_RQ2>{});
    else sweep(std::integral_constant<int, KF_GEN>{});
#undef DCX_HESS_CASE
    }
    __syncthreads();
    // ---- every wave adds its share of the sums (accumulator k belongs to wave k % nw) over the waves' rows, in wave order;
    //      totals land in row 0.  In a split launch (ys > 1) they go straight out instead: the fused gradient kernel's
    //      hand-over (score_kernel.h) - agent-scope (write-through) stores, every storing wave waits for their
    //      acknowledgement, a barrier, ONE arrival atomic per block; the block that arrives last re-reads all ys rows
    //      with agent-scope loads, again one share per wave, and adds them in y order (whatever the arrival order was: the
    //      result does not depend on the schedule), then leaves the counter at zero for the next launch.  No fence on
    //      either side: a release fence's L2 write-back + the acquire's invalidate let 252 blocks evict each other's
    //      support rows.
    typedef unsigned long long u64;
    Dual* row0 = sd + a.o_acc + lane;
    u64* prow = reinterpret_cast<u64*>(a.part + ((size_t)blockIdx.x * a.ys) * a.D * 64 + lane);
    u64* mine = prow + (size_t)blockIdx.y * a.D * 64;
    for (int k = wave; k < a.D; k += nw) {
        Dual t = row0[k * 64];
        for (int w = 1; w < nw; ++w) t += row0[((size_t)w * a.D + k) * 64];
        if (a.ys > 1) {
            const u64 bits = (u64)__float_as_uint(t.v) | ((u64)__float_as_uint(t.d) << 32);
            __hip_atomic_store(mine + k * 64, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            row0[k * 64] = t;
        }
    }
    if (a.ys > 1) {
#ifdef DCX_HANDOVER_FENCE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the drained write-through hand-over relies on gfx942 / gfx950 lowering: build other targets with -DDCX_HANDOVER_FENCE"
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        unsigned int* cnt = a.counters + (size_t)blockIdx.x * a.counter_stride;
        int* flag = reinterpret_cast<int*>(smem + a.o_fk_floats + a.prog_floats);  // one word behind the staged program
        if (threadIdx.x == 0) *flag = (int)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != a.ys - 1) return;
#ifdef DCX_HANDOVER_FENCE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        // the re-read: a wave's first two accumulators go out together, one load per row and accumulator (ys <= 12: at most
        // 24 loads in flight), so most waves pay one agent-scope round trip (one dependent load per value cost 300 cycles
        // each, 30 us at ys = 9, when wave 0 read them one by one; clamped duplicate loads of the last row and one
        // accumulator at a time still cost 4.7 k cycles)
        auto sum_rows = [&](const u64 (&r)[12]) __attribute__((always_inline)) -> Dual {
            Dual t(0.0f, 0.0f);
#pragma unroll
            for (int y = 0; y < 12; ++y)
                if (y < a.ys) {
                    const Dual v(__uint_as_float((unsigned int)r[y]), __uint_as_float((unsigned int)(r[y] >> 32)));
                    t = (y == 0) ? v : t + v;
                }
            return t;
        };
        for (int k0 = wave; k0 < a.D; k0 += 2 * nw) {
            const int k1 = k0 + nw;
            u64 r0[12], r1[12];
#pragma unroll
            for (int y = 0; y < 12; ++y)
                if (y < a.ys) {
                    r0[y] = __hip_atomic_load(prow + ((size_t)y * a.D + k0) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (k1 < a.D) r1[y] = __hip_atomic_load(prow + ((size_t)y * a.D + k1) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            row0[k0 * 64] = sum_rows(r0);
            if (k1 < a.D) row0[k1 * 64] = sum_rows(r1);
        }
        if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // ---- wave 0: the reverse sweep in duals ----
    if (wave != 0) return;
    if (a.fk_dh) fk_vjp_dh_k<Dual>((fk_kptr)(uintptr_t)a.fk, sF, sAcc, sQ);
    else fk_vjp<Dual>(fk, sQ, sF, sAcc, sQ);  // the gradient row is built in place of the q row
    if (live) {
        float* hrow = a.hess + gl * dof;
        for (int k = 0; k < dof; ++k) hrow[k] = sQ[k].d;
        if (a.grad && dir == 0)
            for (int k = 0; k < dof; ++k) a.grad[b * dof + k] = sQ[k].v;
    }
}

// The moments form: per chunk of configurations one hess_moments_kernel launch (lanes = configurations, the supports split over
// gridDim.y when the chunk's tiles do not fill the chip) and one score_hess_kernel launch of single-wave blocks (lanes =
// (configuration, direction) pairs) that reads the sums.  The sums live in stream-ordered scratch.
namespace {
template <int D>
hipError_t launch_moments_width(int kf, const dim3& grid, int nw, size_t lds, hipStream_t stream, const HessArgs& a) {
    auto go = [&](auto kern) {
        if (lds > 64 * 1024)
            if (hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
        hipLaunchKernelGGL(kern, grid, dim3(64 * nw), lds, stream, a);
        return hipGetLastError();
    };
    auto by_up = [&](auto kft) {
        constexpr int KFT = decltype(kft)::value;
        if (!a.upstream) return go(hess_moments_kernel<D, KFT, 0>);
        if (a.c_out == 1) return go(hess_moments_kernel<D, KFT, 1>);
        return go(hess_moments_kernel<D, KFT, 8>);
    };
    if (kf == KF_POLY1) return by_up(std::integral_constant<int, KF_POLY1>{});
    if (kf == KF_RQ2) return by_up(std::integral_constant<int, KF_RQ2>{});
    return by_up(std::integral_constant<int, KF_GEN>{});
}
}  // namespace

static hipError_t launch_hess_moments(const ModelView& m, HessArgs a, int64_t B, hipStream_t stream, bool* not_applicable) {
    *not_applicable = false;
    const int nacc = hess_moments_nacc(m.Dt);
    constexpr int CH = 24;
    const int NW1 = hess_moments_waves(m.Dt, m.kf == KF_POLY1 || m.kf == KF_RQ2 ? m.kf : KF_GEN);
    // first launch: LDS plan in floats - q rows, frames, x, the fold's pass, the staged FK program
    HessArgs k1 = a;
    k1.o_q = 0;
    k1.o_f = k1.o_q + 64 * m.dof;
    k1.o_x = k1.o_f + 64 * m.frame_floats;
    k1.o_acc = k1.o_x + 64 * m.Dt;
    k1.o_fk_floats = k1.o_acc + NW1 * CH * 64;
    const size_t lds1 = sizeof(float) * ((size_t)k1.o_fk_floats + m.prog_floats + 4);
    // second launch: single-wave blocks, LDS plan in duals as in launch_hess
    HessArgs k2 = a;
    k2.o_q = 0;
    k2.o_f = k2.o_q + 64 * m.dof;
    k2.o_x = k2.o_f + 64 * m.frame_floats;
    k2.o_acc = k2.o_x + 64 * m.Dt;
    k2.o_fk_floats = 2 * (k2.o_acc + 64 * m.Dt);
    k2.ys = 1;
    k2.s_super = m.S;
    k2.s_chunk = m.S;
    k2.o_m = k2.o_fk_floats + m.prog_floats + 4;
    const size_t lds2 = sizeof(float) * ((size_t)k2.o_m + (size_t)(64 / m.dof + 2) * nacc);
    if (lds1 > 150 * 1024 || lds2 > 150 * 1024) {
        *not_applicable = true;
        return hipSuccess;
    }
    if (lds2 > 64 * 1024)
        if (hipError_t e = hipFuncSetAttribute((const void*)score_hess_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)) return e;
    const int64_t chunk = std::min<int64_t>(B, (int64_t)kHessMomentRows * 64);   // 512 tiles: two rounds of blocks; 12.7 MB of sums at D = 12
    const int64_t tiles = (chunk + 63) / 64;
    int ys = (int)std::max<int64_t>(1, std::min<int64_t>(16, m.n_cu / tiles));
    while (ys > 1 && (m.S + ys - 1) / ys < NW1 * 16) --ys;   // at least 16 rows per wave
    if (m.ys_knob >= 1) ys = std::min(m.ys_knob, 16);
    while (ys > 1 && tiles * ys > kHessMomentRows) --ys;
    k1.s_super = std::max(1, (m.S + ys - 1) / ys);
    ys = std::max(1, (m.S + k1.s_super - 1) / k1.s_super);
    k1.ys = ys;
    k1.s_chunk = (k1.s_super + NW1 - 1) / NW1;
    {
        static const int skew_env = [] { const char* e = std::getenv("DCX_HESS_SKEW"); return e ? std::atoi(e) : -1; }();
        // 44 / 33 / 23 %: 518 -> 505 us at B = 65536 (profiles/r06_hess_moments.txt; DCX_HESS_SKEW = w0 | w1 << 10 for A/B runs, 0 = equal)
        k1.m_skew = (NW1 == 12 && k1.s_chunk >= 24) ? (skew_env >= 0 ? skew_env : (440 | (330 << 10))) : 0;
    }
    k1.m_ys = k2.m_ys = ys;
    k1.m_nacc = k2.m_nacc = nacc;
    k1.n_lanes = B;   // (configurations)
    float* mom = m.mom;
    if ((size_t)tiles * ys * nacc * 64 * sizeof(float) > m.mom_bytes) {
        *not_applicable = true;
        return hipSuccess;
    }
    k1.mom = k2.mom = mom;
    hipError_t rc = hipSuccess;
    for (int64_t c0 = 0; c0 < B && rc == hipSuccess; c0 += chunk) {
        const int64_t nc = std::min(chunk, B - c0);
        k1.lane0 = c0;
        const dim3 g1((unsigned)((nc + 63) / 64), (unsigned)ys);
        switch (m.Dt) {
            case 2: rc = launch_moments_width<2>(m.kf, g1, NW1, lds1, stream, k1); break;
            case 4: rc = launch_moments_width<4>(m.kf, g1, NW1, lds1, stream, k1); break;
            case 6: rc = launch_moments_width<6>(m.kf, g1, NW1, lds1, stream, k1); break;
            case 8: rc = launch_moments_width<8>(m.kf, g1, NW1, lds1, stream, k1); break;
            case 12: rc = launch_moments_width<12>(m.kf, g1, NW1, lds1, stream, k1); break;
            case 16: rc = launch_moments_width<16>(m.kf, g1, NW1, lds1, stream, k1); break;
            default: rc = hipErrorInvalidValue;
        }
        if (rc != hipSuccess) break;
        k2.lane0 = c0 * m.dof;
        k2.mom_b0 = c0;
        const int64_t nblk = (nc * m.dof + 63) / 64;
        hipLaunchKernelGGL(score_hess_kernel<true>, dim3((unsigned)nblk), dim3(64), lds2, stream, k2);
        rc = hipGetLastError();
    }
    return rc;
}

hipError_t launch_hess(const ModelView& m, const float* q, int64_t B, const float* upstream, float* grad, float* hess,
                       hipStream_t stream) {
    HessArgs a{};
    a.rows = m.rows;
    a.fk = m.fk;
    a.q = q;
    a.upstream = upstream;
    a.grad = grad;
    a.hess = hess;
    a.n_lanes = B * m.dof;
    a.S = m.S;
    a.D = m.Dt;
    a.C = m.C;
    a.c_out = m.c_out;
    a.RS = m.RS;
    a.w_off = m.Dt;
    a.wsum_off = m.Dt + m.C;
    a.dof = m.dof;
    a.d_fk = m.d_fk;
    a.frame_floats = m.frame_floats;
    a.kind = m.kind;
    a.kf = m.kf;
    a.fk_dh = m.fk_dh;
    a.kp0 = m.kp0;
    a.kp1 = m.kp1;
    // Frames that do not fit the LDS as (value, tangent) pairs go to global memory (one 512-byte column per frame float and
    // block, L2-resident): the q row, x and one wave's sums must fit beside the program.
    const int budget = (int)((150 * 1024 - 4 * (size_t)m.prog_floats) / 512);  // Dual columns that fit
    bool paged = m.dof + m.frame_floats + 2 * m.Dt > budget;
    const int fixed = m.dof + (paged ? 0 : m.frame_floats) + m.Dt;  // q row, frames, x
    if (fixed + m.Dt > budget) return hipErrorInvalidValue;
    const int64_t nblk = (a.n_lanes + 63) / 64;
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    // The moments form (hess_moments_kernel) for the narrow compiled widths: knob hess_form 1 = always, 0 = never, otherwise the rule
    if (!paged && m.mom && hess_moments_applies(m, B)) {
        bool na = false;
        const hipError_t e = launch_hess_moments(m, a, B, stream, &na);
        if (!na) return e;
    }
    // Small batches (trust-constr's few hundred dense-path points): split the supports across blocks until the chip is
    // full - each (configuration, direction) tile is swept by ys blocks, the last to arrive folds (kernel above).
    const size_t part_row = (size_t)m.Dt * 64 * sizeof(Dual);
    int ys = 1;
    if (m.scratch && !paged && nblk * 2 <= m.n_cu && nblk <= m.n_counters) {
        ys = (int)(m.n_cu / nblk);
        if (ys > 12) ys = 12;                                                 // (measured: 9-12 blocks per tile, profiles/r03_hess_probe.txt)
        while (ys > 1 && (m.S + ys - 1) / ys < 64) --ys;                      // at least four short slices per block
        if (m.ys_knob >= 1) ys = m.ys_knob < 12 ? m.ys_knob : 12;  // (the hand-over reads at most 12 rows per pass)
        while (ys > 1 && (size_t)nblk * ys * part_row > m.scratch_bytes) --ys;
    }
    a.s_super = std::max(1, (m.S + ys - 1) / ys);            // (dcx_score_hess answers an empty model without a launch)
    ys = std::max(1, (m.S + a.s_super - 1) / a.s_super);     // no empty blocks
    a.ys = ys;
    a.part = reinterpret_cast<Dual*>(m.scratch);
    a.counters = m.counters;
    a.counter_stride = m.counter_stride;
    // waves per block: as many support slices as the LDS allows (Dual = 8 bytes; 64 lanes per column)
    int nw = (budget - fixed) / m.Dt;
    const bool small = m.Dt <= 16;
    const int nw_max = small ? 16 : 8;
    if (nw > nw_max) nw = nw_max;
    while (nw & (nw - 1)) nw &= nw - 1;                                      // a power of two
    while (nw > 1 && (a.s_super + nw - 1) / nw < (ys > 1 ? 14 : 24)) nw >>= 1;  // a slice shorter than that is all FK and fold
    a.s_chunk = (a.s_super + nw - 1) / nw;
    a.o_q = 0;
    a.o_f = a.o_q + 64 * m.dof;
    a.o_x = a.o_f + (paged ? 0 : 64 * m.frame_floats);
    a.o_acc = a.o_x + 64 * m.Dt;
    a.o_fk_floats = 2 * (a.o_acc + nw * 64 * m.Dt);
    a.prog_floats = m.prog_floats;
    const size_t lds = sizeof(float) * ((size_t)a.o_fk_floats + m.prog_floats + 4);  // + the arrival flag of a split launch
    if (lds > 64 * 1024) {
        const void* fn = small ? (const void*)score_hess_kernel<true> : (const void*)score_hess_kernel<false>;
        if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    }
    auto go = [&](int64_t blocks) {
        const dim3 grid((unsigned)blocks, (unsigned)ys);
        if (small)
            hipLaunchKernelGGL(score_hess_kernel<true>, grid, dim3(64 * nw), lds, stream, a);
        else
            hipLaunchKernelGGL(score_hess_kernel<false>, grid, dim3(64 * nw), lds, stream, a);
        return hipGetLastError();
    };
    if (!paged) return go(nblk);
    // paged frames: stream-ordered scratch for at most 2 blocks per CU at a time, the batch in as many launches as that takes
    const int64_t per_launch = std::min<int64_t>(nblk, 2 * (int64_t)m.n_cu);
    const size_t fbytes = (size_t)per_launch * m.frame_floats * 64 * sizeof(Dual);
    if (hipError_t e = hipMallocAsync((void**)&a.frames_g, fbytes, stream)) return e;
    hipError_t rc = hipSuccess;
    for (int64_t b0 = 0; b0 < nblk && rc == hipSuccess; b0 += per_launch) {
        a.lane0 = b0 * 64;
        rc = go(std::min(per_launch, nblk - b0));
    }
    const hipError_t fe = hipFreeAsync(a.frames_g, stream);
    return rc != hipSuccess ? rc : fe;
}

}  // namespace dcx
