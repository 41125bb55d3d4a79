// jac_kernel.h — the full Jacobian d score[b, c] / d q[b, :] of a multi-class model in ONE sweep.
//
// Replaces the C one-hot sweeps dcx_score_jac runs for a chip-filling batch (the optimisers' constraint Jacobians,
// reference diffco/optim.py:211-216 `torch.autograd.functional.jacobian(..., vectorize=True)`, and MultiDiffCo's
// per-class gradients, deprecated/MultiDiffCo.py:156-169).  The distance, the kernel function and the differences of a
// pair do not depend on the class; only the coefficient g * W[j, c] does.  One lane keeps C x D gradient accumulators
// (60 at config #3's C = 5, D = 12: ~100 VGPRs, four waves per SIMD) and pays per pair
//     D/2 v_pk_add + D/2 v_pk_fma + kernel function  +  C x (1 fma + 1 mul + D/2 v_pk_fma)
// instead of C times the whole pair body: 57 instead of 5 x 33 instructions at C = 5.  Small batches keep the existing
// route (the C sweeps side by side in one launch, grid z = class): there the launch is latency, not issue.
//
// The arithmetic per (configuration, class) is exactly the direct-form one-hot sweep's — same differences, same
// accumulation order over the supports, same fixed-order fold across waves — so for equal slicing the rows are
// bit-identical to it (tests/test_gpu_parity.py).  The C J^T products run side by side on waves 0 .. C-1 (one after the
// other on wave 0 for URDF trees, whose reverse sweep keeps adjoint sums in the shared frames).
#pragma once
#include "score_kernel.h"

namespace dcx {

constexpr int kJacMaxAcc = 104;  // D * C + C accumulators per lane that still leave four waves per SIMD
constexpr int kJacMaxD = 24;     // compiled widths
constexpr bool jac_applies(int D, int CC) { return CC > 1 && D <= kJacMaxD && D * CC + CC <= kJacMaxAcc && D + CC <= 38; }

struct LdsPlanJac {
    int q, f, x, g, gq, gq_stride, red, fk, total;
};
__host__ __device__ inline LdsPlanJac lds_plan_jac(int dof, int d_fk, int frame_floats, int acc_floats, int classes) {
    LdsPlanJac p;
    p.q = 0;
    p.f = p.q + ((64 * dof + 3) & ~3);
    p.x = p.f + 64 * frame_floats;
    p.g = p.x + 64 * d_fk;                          // [class][k][64]
    p.gq = p.g + classes * 64 * d_fk;               // [class][64 * dof]
    p.gq_stride = (64 * dof + 3) & ~3;
    p.red = p.gq + classes * p.gq_stride;           // ONE partial row: waves hand over in turn
    p.fk = p.red + acc_floats * 64;
    p.total = p.fk;
    return p;
}

// supports [j0, j1) against this lane's configuration, every class at once; rows broadcast through SGPRs with the
// two-buffer whole-row pipeline of score_kernel.h (wait -> issue next -> consume)
template <int D, int KF, int CC>
__device__ __forceinline__ void sweep_rows_jac(const ScoreArgs& a, const float (&x)[D], int j0, int j1, float (&sc)[CC],
                                               v2f (&gj)[CC][D / 2 + 1], float (&gt)[CC]) {
    using L = RowLayout<D, CC>;
    constexpr int USED = D + CC;
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)a.rows;
    auto load_row = [&](float (&dst)[USED], int j) __attribute__((always_inline)) {
        cfloat_ptr r = rows + (size_t)j * L::RS;
#pragma unroll
        for (int e = 0; e < USED; ++e) dst[e] = r[e];
    };
    auto consume = [&](const float (&r)[USED]) __attribute__((always_inline)) {
        constexpr int NA = DCX_D2_ACCS(D);
        v2f acc[NA], dp[D / 2 + 1];
        float dl = 0.0f;
        acc[0] = v2f{d2_seed<KF>(a), 0.0f};   // (RQ2: constants folded, score_kernel.h sweep_eval)
#pragma unroll
        for (int i = 1; i < NA; ++i) acc[i] = v2f{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k + 1 < D; k += 2) {
            const v2f xv = {x[k], x[k + 1]};
            const v2f rv = {r[k], r[k + 1]};
            dp[k / 2] = xv - rv;
            acc[(k / 2) % NA] = __builtin_elementwise_fma(dp[k / 2], dp[k / 2], acc[(k / 2) % NA]);
        }
#pragma unroll
        for (int i = 1; i < NA; ++i) acc[0] += acc[i];
        float d2 = acc[0].x + acc[0].y;
        if constexpr (D & 1) {
            dl = x[D - 1] - r[D - 1];
            d2 = fmaf(dl, dl, d2);
        }
        float val, g;
        sweep_eval<KF>(d2, a, val, g);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const float w = r[L::W_OFF + c];
            sc[c] = fmaf(w, val, sc[c]);
            const float coef = g * w;
            const v2f c2 = {coef, coef};
#pragma unroll
            for (int k = 0; k + 1 < D; k += 2) gj[c][k / 2] = __builtin_elementwise_fma(c2, dp[k / 2], gj[c][k / 2]);
            if constexpr (D & 1) gt[c] = fmaf(coef, dl, gt[c]);
        }
    };
    if (j0 >= j1) return;
    float bufA[USED], bufB[USED];
    const int jl = j1 - 1;
    load_row(bufA, j0);
    int j = j0;
    for (; j + 1 < j1; j += 2) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        load_row(bufB, j + 1);
        __builtin_amdgcn_sched_barrier(0);
        consume(bufA);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        load_row(bufA, (j + 2 < j1) ? j + 2 : jl);
        __builtin_amdgcn_sched_barrier(0);
        consume(bufB);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (j < j1) consume(bufA);
}

template <int D, int KF, int CC, int MAXT>
__global__ __launch_bounds__(MAXT, (MAXT / 256 > 0 ? MAXT / 256 : 1)) void score_jac_kernel(const ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ACC = CC * D + CC;
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int nb = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
    const int dof = a.dof;
    const LdsPlanJac lp = lds_plan_jac(dof, a.d_fk, a.frame_floats, ACC, CC);
    float* sQ = smem + lp.q;
    float* sF = smem + lp.f;
    float* sX = smem + lp.x;
    float* sRed = smem + lp.red;

    // ---- prologue: as score_kernel ----
    const FkWalk fw = fk_stage_sel(a.fkk, a.fk, a.fk_dwords, a.dh, smem + lp.fk, threadIdx.x, blockDim.x);
    {
        const float* qsrc = a.q + b0 * dof;
        const int n = nb * dof;
        for (int i = threadIdx.x; i < 64 * dof; i += blockDim.x) sQ[i] = qsrc[i < n ? i : (i % dof) + (nb - 1) * dof];
    }
    __syncthreads();
    fk_trig_sel(fw, a.dh, sQ + lane * dof, sF + lane, wave, nw);
    __syncthreads();
    if (wave == 0) fk_chain_sel(fw, a.dh, sQ + lane * dof, sX + lane, sF + lane);
    __syncthreads();
    float x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = (k < a.d_fk) ? sX[k * 64 + lane] : 0.0f;

    // ---- the sweep: this wave's slice, all classes ----
    float sc[CC], gt[CC];
    v2f gj[CC][D / 2 + 1];
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        sc[c] = 0.0f;
        gt[c] = 0.0f;
#pragma unroll
        for (int k = 0; k < D / 2 + 1; ++k) gj[c][k] = v2f{0.0f, 0.0f};
    }
    int j0, j1;
    wave_slice(wave, (int)(blockDim.x >> 6), a.s_chunk, a.s_skew, 0, a.S, j0, j1);   // (the sweep kernel's slices: bit-identical to its one-hot sweeps)
    sweep_rows_jac<D, KF, CC>(a, x, j0, j1, sc, gj, gt);
    lane = fresh_lane();

    // ---- fixed-order fold through ONE LDS row: waves 1 .. nw-1 hand their partial sums to wave 0 in turn ----
    float tot[ACC];
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        tot[c] = sc[c];
#pragma unroll
        for (int k = 0; k < D; ++k)
            tot[CC + c * D + k] = 0.0f + ((k == D - 1 && (D & 1)) ? gt[c] : (k & 1 ? gj[c][k / 2].y : gj[c][k / 2].x));
    }
    for (int w = 1; w < nw; ++w) {
        if (wave == w) {
#pragma unroll
            for (int e = 0; e < ACC; ++e) sRed[e * 64 + lane] = tot[e];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < ACC; ++e) tot[e] += sRed[e * 64 + lane];
        }
        __syncthreads();
    }
    // ---- wave 0: scores out, per-class feature gradients to LDS ----
    float* sG = smem + lp.g;
    if (wave == 0) {
        if (a.score != nullptr && lane < nb) {
#pragma unroll
            for (int c = 0; c < CC; ++c)
                if (c < a.c_out) a.score[(b0 + lane) * a.c_out + c] = tot[c];
        }
#pragma unroll
        for (int c = 0; c < CC; ++c) {
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (k < a.d_fk) sG[(c * a.d_fk + k) * 64 + lane] = tot[CC + c * D + k] * kGradScale<KF>;
        }
    }
    __syncthreads();
    // ---- J^T per class: side by side on waves 0 .. C-1, or one after the other on wave 0 (URDF trees, few waves) ----
    const bool parallel = (nw >= CC) && !fk_is_tree(fw);
    auto finish_class = [&](int c) __attribute__((always_inline)) {
        float* gq = smem + lp.gq + c * lp.gq_stride;
        for (int i = 0; i < dof; ++i) gq[lane * dof + i] = sQ[lane * dof + i];   // the row is built in place of a copy of q
        fk_vjp_sel(fw, a.dh, gq + lane * dof, sF + lane, sG + (size_t)c * a.d_fk * 64 + lane, gq + lane * dof, dof);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        float* gdst = a.grad + b0 * a.grad_stride + (size_t)c * dof;
        const int n = nb * dof;
        for (int i = lane; i < n; i += 64) gdst[(int64_t)(i / dof) * a.grad_stride + (i % dof)] = gq[i];
    };
    if (parallel) {
        if (wave < a.c_out) finish_class(wave);      // (classes beyond the caller's are zero padding: no output row)
    } else if (wave == 0) {
        for (int c = 0; c < a.c_out; ++c) finish_class(c);
    }
}

}  // namespace dcx
