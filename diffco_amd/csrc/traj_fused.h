// traj_fused.h — BASELINE config #5 as ONE persistent launch: n Adam iterations of adam_traj_optimize's loop body
// (reference diffco/optim.py:86-127) on R waypoint paths, one workgroup per path, the iterations looped inside the launch.
//
// dcx_traj_adam_run otherwise enqueues two launches per iteration — the fused score + hinge-gradient sweep over all R*W
// waypoints, then traj_adam_step_kernel (FK again, path terms, J^T, Adam, bookkeeping) — and hands the collision score
// and gradient from one to the other through HBM.  Paths are independent, and a path of W <= 64 waypoints is exactly one
// 64-configuration tile of the sweep, so here a tile is PATH-ALIGNED (lane = waypoint) and everything a path needs stays
// on its CU for the whole run:
//   * the waypoint rows, the Adam moments and the FK frames live in LDS across iterations (HBM sees the path and the
//     moments once at the start and once at the end, plus the rare best-so-far copies);
//   * ONE forward kinematics per iteration serves the collision sweep AND the path-length / max-move terms (the
//     two-launch form computes it twice); neighbouring waypoints' control points are read from the same LDS slab;
//   * the two J^T products (collision gradient, path-term gradient) run side by side on waves 0 and 1;
//   * no launch boundary, no second FK, no col_score / col_grad round trip.
// Every arithmetic expression is the one the two-launch form evaluates, in the same order (same sweep slices, same
// cross-wave fold, the two J^T products kept separate and added as `path + collision`), so for equal slicing the
// results are bit-identical to it (tests/test_gpu_traj.py).
#pragma once
#include "score_kernel.h"

// Squared-distance accumulator chains of the sweep inside the persistent kernel.  It runs at 4 waves per SIMD, where two
// shorter dependent chains beat one by 2 % (36.0 -> 35.2 us per iteration with -DDCX_TRAJ_NACC=2), but a different
// summation order gives up the bit-identity with the two-launch loop that tests/test_gpu_traj.py holds it to: default 0
// (the sweep kernel's own rule).
#ifndef DCX_TRAJ_NACC
#define DCX_TRAJ_NACC 0
#endif

namespace dcx {

constexpr int kTrajFusedMaxIters = 192;  // iterations per launch (the bias corrections travel as kernel arguments)

struct TrajFusedArgs {
    ScoreArgs sc;            // rows, fk, S, s_chunk, dof, d_fk, frame_floats, kind, kp0, kp1 (the sweep's view of the model)
    dcx_traj_state st;
    dcx_traj_opts opt;
    int32_t n_iters;
    int32_t n_points, point_dim, coord_major;
    float bias1[kTrajFusedMaxIters];       // 1 - beta1^t          (host double arithmetic, like launch_traj_adam_step)
    float bias2_sqrt[kTrajFusedMaxIters];  // sqrt(1 - beta2^t)
};

// LDS carve of the persistent kernel (floats)
struct TrajFusedPlan {
    int q, m, v, gqc, gqp, f, x, gc, gp, red, r, fk, total;
};
// d_acc = the compiled feature width D of the sweep (>= d_fk): a fold row holds D + 1 floats per lane
__host__ __device__ inline TrajFusedPlan traj_fused_plan(int dof, int d_fk, int frame_floats, int nw, int d_acc) {
    TrajFusedPlan p;
    const int rows = (64 * dof + 3) & ~3;
    p.q = 0;
    p.m = p.q + rows;
    p.v = p.m + rows;
    p.gqc = p.v + rows;
    p.gqp = p.gqc + rows;
    p.f = p.gqp + rows;
    p.x = p.f + 64 * frame_floats;
    p.gc = p.x + 64 * d_fk;
    p.gp = p.gc + 64 * d_fk;
    p.red = p.gp + 64 * d_fk;
    p.r = p.red + nw * (d_acc + 1) * 64;
    p.fk = p.r + 128;
    p.total = p.fk;
    return p;
}

__device__ __forceinline__ float traj_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Register budget: ONE block per CU is the design point (256 restarts on 256 CUs; the LDS carve is ~90 KB), i.e.
// MAXT / 256 waves per SIMD, so the allocator may use 512 / (MAXT / 256) VGPRs instead of the sweep kernel's 64: at
// 64 the loop-carried state of the iteration (waypoint, moments, path terms) lived in scratch (236 B per lane).
template <int D, int KF, int MAXT, bool XF = false>
__global__ __launch_bounds__(MAXT, (MAXT / 256 > 0 ? MAXT / 256 : 1)) void traj_fused_kernel(const TrajFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ACC = D + 1;
    const int r = blockIdx.x;
    if (a.st.done[r]) return;  // frozen path
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int W = a.st.n_waypoints, dof = a.sc.dof, d_fk = a.sc.d_fk;
    const TrajFusedPlan lp = traj_fused_plan(dof, d_fk, a.sc.frame_floats, nw, D);
    float* sQ = smem + lp.q;
    float* sM = smem + lp.m;
    float* sV = smem + lp.v;
    float* sGQc = smem + lp.gqc;
    float* sGQp = smem + lp.gqp;
    float* sF = smem + lp.f;
    float* sX = smem + lp.x;
    float* sGc = smem + lp.gc;
    float* sGp = smem + lp.gp;
    float* sRed = smem + lp.red;
    float* sR = smem + lp.r;

    const FkWalk fw = fk_stage_sel(a.sc.fkk, a.sc.fk, a.sc.fk_dwords, a.sc.dh, smem + lp.fk, tid, blockDim.x);
    {
        // waypoint rows (lanes past the path replicate its last row, like a ragged tile of the sweep) and Adam moments
        const size_t base = (size_t)r * W * dof;
        const int n = W * dof;
        for (int i = tid; i < 64 * dof; i += blockDim.x) {
            const int ii = i < n ? i : (i % dof) + (W - 1) * dof;
            sQ[i] = a.st.path[base + ii];
            sM[i] = i < n ? a.st.adam_m[base + i] : 0.f;
            sV[i] = i < n ? a.st.adam_v[base + i] : 0.f;
        }
    }
    __syncthreads();

    const int w = lane;  // this lane's waypoint (waves 0 and 1 use it)
    const bool live = w < W;
    const float v2 = a.opt.max_speed * a.opt.max_speed;
    const int pd = a.point_dim;
    // this wave's slice of the supports (the sweep's slicing: wave w takes [w * s_chunk, (w + 1) * s_chunk))
    const int j0 = (wave * a.sc.s_chunk < a.sc.S) ? wave * a.sc.s_chunk : a.sc.S;
    const int j1 = (j0 + a.sc.s_chunk < a.sc.S) ? j0 + a.sc.s_chunk : a.sc.S;
    const bool tree = fk_is_tree(fw);  // its reverse sweep keeps adjoint sums in the frames: one at a time

    int it = 0;
    for (; it < a.n_iters; ++it) {
        // ---- forward kinematics of the current waypoints (once per iteration) -----------------------------------
        fk_trig_sel(fw, a.sc.dh, sQ + lane * dof, sF + lane, wave, nw);
        __syncthreads();
        if (wave == 0) fk_chain_sel(fw, a.sc.dh, sQ + lane * dof, sX + lane, sF + lane);
        __syncthreads();
        float x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = (k < d_fk) ? sX[k * 64 + lane] : 0.0f;

        // ---- collision sweep: score and feature gradient against this wave's supports ---------------------------
        float sc[1] = {0.0f};
        float gx[D];
        const float up[1] = {1.0f};
#pragma unroll
        for (int k = 0; k < D; ++k) gx[k] = 0.0f;
        sweep_rows<D, KF, 1, MODE_GRAD_ROW, XF, DCX_TRAJ_NACC>(a.sc, x, up, j0, j1, sc, gx);
        if (nw > 1) {
            // the sweep's parallel cross-wave fold (score_kernel.h): row 0 first, then 1, 2, ...
            float* mine = sRed + (size_t)wave * ACC * 64 + lane;
            mine[0] = sc[0];
#pragma unroll
            for (int k = 0; k < D; ++k) mine[(1 + k) * 64] = gx[k];
            __syncthreads();
            fold_partial_rows<ACC>(sRed, wave, lane, nw);
            __syncthreads();
        }
        // ---- wave 0: hinge + J^T of the collision gradient;  wave 1: path terms + their J^T ------------------------
        float obj = 0.f, mmv = 0.f, col = 0.f;
        const int pwave = (nw > 1 && !tree) ? 1 : 0;
        if (wave == 0) {
            if (nw > 1) {
                sc[0] = sRed[lane];
#pragma unroll
                for (int k = 0; k < D; ++k) gx[k] = sRed[(1 + k) * 64 + lane];
            }
            const float scale = (sc[0] - a.opt.safety_margin > 0.0f) ? a.opt.w_collision : 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (k < d_fk) sGc[k * 64 + lane] = gx[k] * scale;
            // q row -> gq row in a separate buffer (the two-launch form overwrites the q row; same arithmetic)
            for (int i = 0; i < dof; ++i) sGQc[lane * dof + i] = sQ[lane * dof + i];
            fk_vjp_sel(fw, a.sc.dh, sGQc + lane * dof, sF + lane, sGc + lane, sGQc + lane * dof, dof);
            const float s0 = sc[0] - a.opt.safety_margin;
            if (live && s0 > 0.f) col = s0;
        }
        if (wave == pwave) {
            auto X = [&](int k, int v) { return sX[k * 64 + v]; };
            for (int p = 0; p < a.n_points; ++p) {
                float dn[3] = {0.f, 0.f, 0.f}, dp[3] = {0.f, 0.f, 0.f};
                float n2n = 0.f, n2p = 0.f;
                for (int c = 0; c < pd; ++c) {
                    const int k = a.coord_major ? c * a.n_points + p : p * pd + c;
                    const float xc = live ? X(k, w) : 0.f;
                    if (live && w + 1 < W) { dn[c] = X(k, w + 1) - xc; n2n = fmaf(dn[c], dn[c], n2n); }
                    if (live && w >= 1)    { dp[c] = xc - X(k, w - 1); n2p = fmaf(dp[c], dp[c], n2p); }
                }
                const float mn = n2n - v2, mp = n2p - v2;
                if (live && w + 1 < W) {   // each segment is counted once, by its left waypoint
                    obj += n2n;
                    if (mn > 0.f) mmv += mn;
                }
                const float cn = 2.f * (a.opt.w_diff + (mn > 0.f ? a.opt.w_max_move : 0.f));
                const float cp = 2.f * (a.opt.w_diff + (mp > 0.f ? a.opt.w_max_move : 0.f));
                for (int c = 0; c < pd; ++c) sGp[(a.coord_major ? c * a.n_points + p : p * pd + c) * 64 + lane] = cp * dp[c] - cn * dn[c];
            }
            fk_vjp_sel(fw, a.sc.dh, sQ + lane * dof, sF + lane, sGp + lane, sGQp + lane * dof, dof);
            const float so = traj_wave_sum(obj), sm = traj_wave_sum(mmv);
            if (lane == 0) { sR[0] = so; sR[16] = sm; }
        }
        __syncthreads();
        // ---- wave 0: joint limits, endpoint mask, Adam, loss terms, best-so-far bookkeeping -------------------------
        if (wave == 0) {
            float jl = 0.f, gn2 = 0.f;
            if (live) {
                const bool endpoint = (w == 0) || (w == W - 1);
                const float b1 = a.bias1[it], b2s = a.bias2_sqrt[it];
                for (int i = 0; i < dof; ++i) {
                    const float q = sQ[lane * dof + i];
                    const float lo = a.st.limits[2 * i], hi = a.st.limits[2 * i + 1];
                    float g = sGQp[lane * dof + i] + sGQc[lane * dof + i];
                    if (q < lo) { jl += lo - q; g -= a.opt.w_joint_limit; }
                    if (q > hi) { jl += q - hi; g += a.opt.w_joint_limit; }
                    if (endpoint) g = 0.f;  // p.grad[[0, -1]] = 0 (optim.py:102)
                    gn2 = fmaf(g, g, gn2);
                    float m = sM[lane * dof + i], v = sV[lane * dof + i];
                    m = fmaf(a.opt.beta1, m, (1.f - a.opt.beta1) * g);
                    v = fmaf(a.opt.beta2, v, (1.f - a.opt.beta2) * g * g);
                    const float denom = sqrtf(v) / b2s + a.opt.eps;
                    const float qn = q - (a.opt.lr / b1) * (m / denom);
                    sM[lane * dof + i] = m;
                    sV[lane * dof + i] = v;
                    sQ[lane * dof + i] = qn;
                }
            }
            const float t_jl = traj_wave_sum(jl), t_col = traj_wave_sum(col), t_gn2 = traj_wave_sum(gn2);
            if (lane == 0) {
                // the step kernel's totals: 0 + (one per-wave partial)
                const float tot0 = 0.f + sR[0], tot1 = 0.f + sR[16], tot2 = 0.f + t_jl, tot3 = 0.f + t_col, tot4 = 0.f + t_gn2;
                const float objective = a.opt.w_diff * tot0;
                const float constraint = a.opt.w_collision * tot3 + a.opt.w_max_move * tot1 + a.opt.w_joint_limit * tot2;
                const float loss = objective + constraint;
                const float gnorm = sqrtf(tot4);
                float* st = a.st.stats + (size_t)r * 8;
                st[0] = loss; st[1] = objective; st[2] = constraint; st[3] = gnorm; st[4] = tot3; st[5] = tot1; st[6] = tot2;
                st[7] = 0.f;
                int flags = 0;
                if (loss < a.st.lowest_loss[r]) {  // optim.py:107-112 (solution = p AFTER the step)
                    a.st.lowest_loss[r] = loss;
                    a.st.lowest_obj[r] = objective;
                    flags |= 1;
                }
                if (constraint <= a.opt.valid_tol) {  // optim.py:113-118
                    if (objective < a.st.best_valid_obj[r]) {
                        a.st.best_valid_obj[r] = objective;
                        flags |= 2;
                    }
                    if (gnorm < a.opt.grad_tol) flags |= 4;  // optim.py:126-127: the path stops here
                }
                a.st.steps[r] += 1;
                sR[96] = __int_as_float(flags);
            }
        }
        __syncthreads();
        const int flags = __float_as_int(sR[96]);
        if (flags & 3) {
            float* lo = a.st.lowest_path + (size_t)r * W * dof;
            float* bv = a.st.best_valid_path + (size_t)r * W * dof;
            for (int i = tid; i < W * dof; i += blockDim.x) {
                const float v = sQ[i];
                if (flags & 1) lo[i] = v;
                if (flags & 2) bv[i] = v;
            }
        }
        // lanes past the path follow its last row (their FK feeds nothing, but keep them finite and in step)
        if (tid < dof) {
            for (int l = W; l < 64; ++l) sQ[l * dof + tid] = sQ[(W - 1) * dof + tid];
        }
        __syncthreads();
        if (flags & 4) {
            if (tid == 0) a.st.done[r] = 1;
            ++it;
            break;
        }
    }
    (void)it;
    // ---- state back to HBM ------------------------------------------------------------------------------------------
    {
        const size_t base = (size_t)r * W * dof;
        for (int i = tid; i < W * dof; i += blockDim.x) {
            a.st.path[base + i] = sQ[i];
            a.st.adam_m[base + i] = sM[i];
            a.st.adam_v[base + i] = sV[i];
        }
    }
}

}  // namespace dcx
