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
// exactness and generality — any feature width, every kernel function and transform — not for the instruction-rate
// roofline of the sweep: for D <= 32 x, dx and the sums sit in registers, beyond that in LDS (4 D ds operations per pair
// and lane).
#include "dcx_internal.h"

#include <algorithm>

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
