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

#ifdef DCX_TIMING
#define DCX_TTS(slot)                                                                             \
    do {                                                                                          \
        if (a.sc.ts && blockIdx.x == 0 && blockIdx.y == 0 && it == 2 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 16) \
            a.sc.ts[(slot) * 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();             \
    } while (0)
#else
#define DCX_TTS(slot) do { } while (0)
#endif

// Arithmetic shared by the persistent kernel below and traj_adam_step_kernel (traj_kernels.hip), which tests hold to
// identical bits: every product and sum here rounds on its own, as the separate torch operations of the reference do
// (optim.py:86-103).  Without the pragma the compiler fuses mul + add pairs into fma wherever its DAG combiner sees one,
// which depends on the code around the expression - the two kernels would agree only by luck.
__device__ __forceinline__ float traj_excess(float n2, float max_speed) {
#pragma clang fp contract(off)
    return n2 - max_speed * max_speed;
}
__device__ __forceinline__ float traj_path_grad(float cp, float dp, float cn, float dn) {
#pragma clang fp contract(off)
    return cp * dp - cn * dn;
}
__device__ __forceinline__ float traj_adam_q(float q, float lr, float bias1, float m, float denom) {
#pragma clang fp contract(off)
    return q - (lr / bias1) * (m / denom);
}
__device__ __forceinline__ float traj_constraint(float w_col, float t_col, float w_mm, float t_mm, float w_jl, float t_jl) {
#pragma clang fp contract(off)
    return w_col * t_col + w_mm * t_mm + w_jl * t_jl;
}

constexpr int kTrajFusedMaxIters = 192;  // iterations per launch (the bias corrections travel as kernel arguments)

struct TrajFusedArgs {
    ScoreArgs sc;            // rows, fk, S, s_chunk, dof, d_fk, frame_floats, kind, kp0, kp1 (the sweep's view of the model)
    dcx_traj_state st;
    dcx_traj_opts opt;
    int32_t n_iters;
    int32_t n_points, point_dim, coord_major;
    // cluster form (traj_fused_kernel<..., CL = true>, gridDim.y = ys): a path's supports split over ys workgroups
    int32_t ys, s_super;            // block y sweeps supports [y * s_super, (y + 1) * s_super), its waves s_chunk each
    unsigned long long* exch;       // [n_paths][2 (iteration parity)][ys][D + 1, or CC + 2 D for several classes][64] (value, tag) words, see traj_exchange
    uint32_t tag_base;              // tags of this launch: tag_base + iteration + 1 (unique per launch on this buffer)
    int32_t cl_across;              // 1: grid (ys, n_paths) - consecutive workgroup ids = the ys members of a path, which the
                                    // dispatcher deals out to DIFFERENT XCDs; 0: grid (n_paths, ys) - with n_paths a multiple of
                                    // 8 the members of a path land on ONE XCD and exchange through its L2 (see traj_exchange)
    float margin_c[8];              // CC > 1 (traj_fused_kernel<..., CC>): the per-class safety margins (opt.safety_margin is the one of CC == 1)
    float bias1[kTrajFusedMaxIters];       // 1 - beta1^t          (host double arithmetic, like launch_traj_adam_step)
    float bias2_sqrt[kTrajFusedMaxIters];  // sqrt(1 - beta2^t)
};

// LDS carve of the persistent kernel (floats)
struct TrajFusedPlan {
    int q, m, v, gqc, gqp, f, x, gc, gp, red, r, jl, jt, fk, total;
};
// d_acc = the compiled feature width D of the sweep (>= d_fk): a fold row holds D + 1 floats per lane;  n_pt > 0: the
// several-wave J^T's scratch, 15 columns per point step (9 rotation, 3 + 3 for the collision and the path gradient)
__host__ __device__ inline TrajFusedPlan traj_fused_plan(int dof, int d_fk, int frame_floats, int nw, int d_acc, int n_pt = 0, int cc = 1) {
    TrajFusedPlan p;
    const int rows = (64 * dof + 3) & ~3;
    p.x = 0;
    p.q = p.x + 64 * d_fk;
    p.m = p.q + rows;
    p.v = p.m + rows;
    p.gqc = p.v + rows;
    p.gqp = p.gqc + rows;
    p.f = p.gqp + rows;
    p.gc = p.f + 64 * frame_floats;
    p.gp = p.gc + 64 * d_fk;
    p.red = p.gp + 64 * d_fk;
    p.r = p.red + nw * (d_acc + cc) * 64;   // cc: class scores in front of the feature gradient
    p.jl = p.r + 128;                 // [2 dof][64]: per-joint limit excess and gradient entry (Adam on several waves)
    p.jt = p.jl + 2 * dof * 64;
    p.fk = p.jt + 15 * n_pt * 64;
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
//
// SGPRs (round 3): TrajFusedArgs is 2.4 KB of kernel arguments.  With every phase of the iteration reading them from the
// one by-value parameter the allocator kept dozens live across the whole loop and parked them in VGPR lanes - 2129
// v_readlane / v_writelane in the kernel, most of them in the lone-wave phases, which pay ~6 cycles for each
// (tools/lone_wave_ubench.hip).  Every phase below therefore reads what it needs AFRESH from the kernarg segment
// (reload_kernargs: scalar-cache hits) and re-derives its LDS pointers; nothing but the loop counter, the lane's own
// indices and the sweep's operands crosses the sweep.
struct TrajLds {
    float *sQ, *sM, *sV, *sGQc, *sGQp, *sF, *sX, *sGc, *sGp, *sRed, *sR, *sJl, *sJ, *sFk;
};
template <int D, int CC = 1, class A>
__device__ __forceinline__ TrajLds traj_lds(float* smem, const A& b, int nw) {
    const bool jt = b.sc.jt_waves != 0;
    const TrajFusedPlan lp = traj_fused_plan(b.sc.dof, b.sc.d_fk, b.sc.frame_floats, nw, D, jt ? b.sc.dh.n_pt : 0, CC);
    TrajLds l;
    l.sQ = smem + lp.q; l.sM = smem + lp.m; l.sV = smem + lp.v; l.sGQc = smem + lp.gqc; l.sGQp = smem + lp.gqp;
    l.sF = smem + lp.f; l.sX = smem + lp.x; l.sGc = smem + lp.gc; l.sGp = smem + lp.gp; l.sRed = smem + lp.red;
    l.sR = smem + lp.r; l.sJl = smem + lp.jl; l.sJ = smem + lp.jt; l.sFk = smem + lp.fk;
    return l;
}
// sR (128 floats): [0] sum of segment lengths^2, [16] max-move excess, [32 .. 32 + 2 dof) joint limits (staged once),
// [17] hinge excess, [96] flags of the iteration; the path's records, read once at the start of the launch, kept here while
// it iterates and written back once at its end (round 4: as global read-modify-writes of one lane they put a memory round
// trip in front of a barrier every iteration; in the cluster form every workgroup needs its own copy anyway - all must
// take the same decisions): [18] lowest loss so far, [19] best valid objective so far, [21] objective at the lowest loss,
// [22] steps taken (int bits), [97 .. 104] the loss terms of the last step (`stats`);  [20] != 0: an exchange gave up
constexpr int kTrajLim = 32, kTrajFlags = 96, kTrajLowest = 18, kTrajBestValid = 19, kTrajAbort = 20, kTrajLowestObj = 21,
              kTrajSteps = 22, kTrajStats = 97;

// ---- the cluster form's exchange (round 4) ---------------------------------------------------------------------------
// One workgroup per path leaves a 32-restart shard (BASELINE config #5 on 8 GPUs) on 32 of 256 CUs, each spending 3/4 of
// an iteration in a sweep that 8 CUs could share.  In the cluster form ys workgroups (gridDim.y) own a path TOGETHER: each
// keeps the whole state of the path in its LDS and runs the whole iteration - FK, path terms, both J^T, Adam, bookkeeping -
// redundantly and identically, but sweeps only its 1 / ys of the supports.  Once per iteration the ys partial rows (score +
// feature gradient per waypoint: D + 1 floats x 64 lanes) are exchanged ALL TO ALL through global memory, so every
// workgroup ends up with the same totals and the iteration needs no second hand-over and no leader:
//   * a value travels as ONE 8-byte word (value, tag) written with an agent-scope store (write-through to where the
//     other XCDs read); 8-byte stores are single-copy atomic, so a reader that sees the tag sees the value: no drain, no
//     arrival counter, no fence - one store and one round of polling loads per iteration;
//   * tag = tag_base + iteration + 1 is unique per launch and iteration on this buffer (the host hands out tag_base), so
//     nothing is ever reset; rows alternate between two slots by iteration parity: a workgroup can be at most one
//     exchange ahead of its slowest peer, which is then still reading the OTHER slot;
//   * wave w of every workgroup owns accumulator w (w, w + nw, ...): it folds it over the block's nw partial rows
//     (row 0 first, like fold_partial_rows), publishes it, polls the ys copies and adds them in the order y = 0, 1, ... -
//     the sums a split launch of the sweep kernel forms (score_kernel.h), so for equal slicing the cluster form is
//     bit-identical to the two-launch loop as well.  No barrier inside the exchange.
//   Tried and not kept: the path terms and phase R1 of J^T BETWEEN publishing and polling instead of in front of the sweep
//   (they do not need the totals).  With twelve waves polling beside it the path-terms wave took 7.4 k cycles instead of
//   5 k and became the critical path: 12.65 -> 12.85 us per iteration at 32 paths (profiles/r04_traj_cluster.txt).
// The ys workgroups of a path must be resident together: the host launches the grid cooperatively (n_paths * ys <= CUs).
// A poll that sees nothing for ~1 s gives up: the launch ends with stats[r][7] = -1, the path and its moments as the
// launch found them (the loss records may have been touched).
constexpr int kTrajPollLimit = 1 << 19;
__device__ __forceinline__ unsigned long long traj_pack(float v, uint32_t tag) {
    return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
// first half: fold this wave's accumulators over the block's rows and publish them
// (E0, E1: the accumulators of this exchange - all of them with one class; with several the class scores [0, CC) travel after the
// first sweep and the feature gradient [CC, CC + D) after the second, through disjoint words of the same slot)
// XS / XO: words per workgroup in the exchange rows and the offset of accumulator 0 there (several classes: the gradient of a
// failed speculation's second sweep travels through words of its own, see the kernel)
template <int ACC, int E0 = 0, int E1 = ACC, int XS = ACC, int XO = 0>
__device__ __forceinline__ void traj_exchange_publish(const float* sRed, unsigned long long* slot /* this path's rows of this parity, + lane */,
                                                      uint32_t tag, int y, int wave, int lane, int nw) {
    auto fold_pub = [&](auto nwc) __attribute__((always_inline)) {
        constexpr int NWC = decltype(nwc)::value;
        const int nwr = NWC > 0 ? NWC : nw;
        for (int e = E0 + wave; e < E1; e += nwr) {
            float v;
            if constexpr (NWC > 0) {
                float r[NWC];
#pragma unroll
                for (int w = 0; w < NWC; ++w) r[w] = sRed[((size_t)w * ACC + e) * 64 + lane];
                v = r[0];
#pragma unroll
                for (int w = 1; w < NWC; ++w) v += r[w];
            } else {
                v = sRed[e * 64 + lane];
                for (int w = 1; w < nw; ++w) v += sRed[((size_t)w * ACC + e) * 64 + lane];
            }
            __hip_atomic_store(slot + ((size_t)y * XS + XO + e) * 64, traj_pack(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (nw == 16) fold_pub(std::integral_constant<int, 16>{});
    else if (nw == 8) fold_pub(std::integral_constant<int, 8>{});
    else if (nw == 4) fold_pub(std::integral_constant<int, 4>{});
    else fold_pub(std::integral_constant<int, 0>{});
}
// second half: poll the ys copies of each of this wave's accumulators; totals to row 0 of the scratch (only this wave touches
// accumulator e's slots)
template <int ACC, int E0 = 0, int E1 = ACC, int XS = ACC, int XO = 0>
__device__ __forceinline__ bool traj_exchange_collect(float* sRed, const unsigned long long* slot, uint32_t tag, int ys, int wave, int lane,
                                                      int nw) {
    bool ok = true;
    for (int e = E0 + wave; e < E1; e += nw) {
        float tot = 0.0f;
        int spins = 0;
        for (;;) {
            tot = 0.0f;
            bool all = true;
            for (int y0 = 0; y0 < ys; y0 += 8) {
                unsigned long long w[8];
#pragma unroll
                for (int v = 0; v < 8; ++v)
                    if (y0 + v < ys) w[v] = __hip_atomic_load(slot + ((size_t)(y0 + v) * XS + XO + e) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int v = 0; v < 8; ++v)
                    if (y0 + v < ys) {
                        all = all && ((uint32_t)(w[v] >> 32) == tag);
                        tot += __uint_as_float((uint32_t)w[v]);
                    }
            }
            if (__builtin_amdgcn_ballot_w64(!all) == 0) break;
            if (++spins > kTrajPollLimit) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        sRed[e * 64 + lane] = tot;
    }
    return ok;
}

// The path terms of one lane's waypoint: length / max-move gradient with respect to its control points -> sGp, and the
// wave's objective / max-move sums -> sR[0], sR[16].  They read the features in LDS only, so the persistent kernel runs
// them BEFORE the sweep, on a wave whose latency the other waves' sweeps hide (round 3; after the fold they cost 5 k
// exposed cycles per iteration: profiles/r03_traj_phase.txt).
template <typename A>
__device__ __forceinline__ void traj_path_terms(const A& b, const TrajLds& L, int lane) {
    const int w = lane;
    const int W = b.st.n_waypoints, pd = b.point_dim, n_points = b.n_points;
    const bool live = w < W;
    float obj = 0.f, mmv = 0.f;
        const float ms = b.opt.max_speed;
        const float w_diff = b.opt.w_diff, w_mm = b.opt.w_max_move;
        const bool has_n = live && w + 1 < W, has_p = live && w >= 1;
        const int wn = has_n ? w + 1 : w, wp = has_p ? w - 1 : w;  // a missing neighbour reads the lane's own point
        auto one = [&](int kx, int ky, int kz, int npd) __attribute__((always_inline)) {
            // npd coordinates of one control point at feature columns kx, ky, kz
            const int kk[3] = {kx, ky, kz};
            float dn[3] = {0.f, 0.f, 0.f}, dp[3] = {0.f, 0.f, 0.f};
            float n2n = 0.f, n2p = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c < npd) {
                    const float xc = live ? L.sX[kk[c] * 64 + w] : 0.f;
                    if (has_n) { dn[c] = L.sX[kk[c] * 64 + wn] - xc; n2n = fmaf(dn[c], dn[c], n2n); }
                    if (has_p) { dp[c] = xc - L.sX[kk[c] * 64 + wp]; n2p = fmaf(dp[c], dp[c], n2p); }
                }
            }
            const float mn = traj_excess(n2n, ms), mp = traj_excess(n2p, ms);
            if (has_n) {   // each segment is counted once, by its left waypoint
                obj += n2n;
                if (mn > 0.f) mmv += mn;
            }
            const float cn = 2.f * (w_diff + (mn > 0.f ? w_mm : 0.f));
            const float cp = 2.f * (w_diff + (mp > 0.f ? w_mm : 0.f));
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < npd) L.sGp[kk[c] * 64 + lane] = traj_path_grad(cp, dp[c], cn, dn[c]);
        };
        if (pd == 3 && !b.coord_major) {
            for (int p = 0; p < n_points; ++p) one(3 * p, 3 * p + 1, 3 * p + 2, 3);
        } else {
            for (int p = 0; p < n_points; ++p) {
                const int k0 = b.coord_major ? p : p * pd, st = b.coord_major ? n_points : 1;
                one(k0, k0 + st, k0 + 2 * st, pd);
            }
        }
        const float so = traj_wave_sum(obj), sm = traj_wave_sum(mmv);
        if (lane == 0) { L.sR[0] = so; L.sR[16] = sm; }
}

// CC > 1 (round 6): a multi-class checker (MultiDiffCo.rbf_score -> [W, C]) under the reference's per-class margins,
//   collision = sum_w sum_c clamp(score_wc - margin_c, 0)          (optim.py:88-89 with safety_margin [C]: scripts/2d_trajopt.py:94-102,
//                                                                   scripts/active.py:35, 65)
// The gradient's upstream w_collision * 1[score_wc > margin_c] is only known once ALL supports have been seen, so an iteration
// sweeps twice: class scores first (MODE_SCORE; fold / exchange of the CC score accumulators), then the feature gradient with that
// upstream per lane (MODE_GRAD_UP; fold / exchange of the D gradient accumulators) - the arithmetic, slicing and fold order of
// dcx_score followed by dcx_score_hinge_grad_mc's second launch, which is what the three-launch loop of dcx_traj_adam_run_mc runs
// where this kernel is not compiled (bit-identical for equal slicing, tests/test_gpu_traj.py).  Fold rows: [c][64] scores, then
// [k][64] gradient (the sweep kernel's layout).
// SPECULATION (round 6): between two Adam steps the hinge's indicator rarely changes, so an iteration first sweeps ONCE with the
// previous iteration's upstream - MODE_GRAD_UP accumulates the class scores beside the gradient - folds / exchanges scores and
// gradient together and compares the indicator the new scores give with the one it used: equal on every lane and the iteration
// is done with one sweep; else the gradient is swept again with the right upstream.  An iteration speculates only if the
// indicator did not move in the one before (a two-sweep iteration sees that for free), so a path whose indicators keep flipping -
// random restarts far from convergence - keeps the two sweeps and pays nothing for failed attempts.  Every wave - and every workgroup of a cluster - sees the same totals, so all take the same branch.
// The gradient an iteration ends with is always the sweep with the indicator of ITS OWN scores, and the scores are the
// same fma chains in both sweep modes: the results do not depend on whether a speculation held (bit-identical to the
// three-launch loop, tests/test_gpu_multiclass_optim.py).  Exchange rows of the cluster form: CC + 2 D words per workgroup -
// [0, CC + D) for the first exchange(s) of an iteration, [CC + D, CC + 2 D) for the gradient after a failed speculation, whose
// peers may still be reading the first.
template <int D, int KF, int MAXT, bool XF = false, bool CL = false, int CC = 1>
__global__ __launch_bounds__(MAXT, (MAXT / 256 > 0 ? MAXT / 256 : 1)) void traj_fused_kernel(const TrajFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ACC = D + CC;
    constexpr int XS = CC > 1 ? CC + 2 * D : ACC;   // exchange words per workgroup (see SPECULATION above)
    const int r = (CL && a.cl_across) ? blockIdx.y : blockIdx.x;
    const int ycl = CL ? (a.cl_across ? (int)blockIdx.x : (int)blockIdx.y) : 0;   // this workgroup's place among the path's ys
    if (a.st.done[r]) return;  // frozen path (cluster form: the ys workgroups of a path all see the same flag - it is only
                               // ever written by a workgroup that leaves, and none leaves before all have passed here)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    {
        const int W = a.st.n_waypoints, dof = a.sc.dof;
        const TrajLds L = traj_lds<D, CC>(smem, a, nw);
        (void)fk_stage_sel(a.sc.fkk, a.sc.fk, a.sc.fk_dwords, a.sc.dh, L.sFk, tid, blockDim.x);
        // waypoint rows (lanes past the path replicate its last row, like a ragged tile of the sweep) and Adam moments
        const size_t base = (size_t)r * W * dof;
        const int n = W * dof;
        for (int i = tid; i < 64 * dof; i += blockDim.x) {
            const int ii = i < n ? i : (i % dof) + (W - 1) * dof;
            L.sQ[i] = a.st.path[base + ii];
            L.sM[i] = i < n ? a.st.adam_m[base + i] : 0.f;
            L.sV[i] = i < n ? a.st.adam_v[base + i] : 0.f;
        }
        if (tid < 2 * dof) L.sR[kTrajLim + tid] = a.st.limits[tid];  // the joint limits never change: read once
        if (tid == 0) {
            L.sR[kTrajLowest] = a.st.lowest_loss[r];
            L.sR[kTrajBestValid] = a.st.best_valid_obj[r];
            L.sR[kTrajLowestObj] = a.st.lowest_obj[r];
            L.sR[kTrajSteps] = __int_as_float(a.st.steps[r]);
            L.sR[kTrajAbort] = 0.0f;
        }
        if (tid < 8) L.sR[kTrajStats + tid] = a.st.stats[(size_t)r * 8 + tid];
    }
    __syncthreads();

    const int w = lane;  // this lane's waypoint
    float up_prev[CC];   // several classes: the upstream the last iteration ended with, and whether this one may speculate on it
    bool spec = false;
#pragma unroll
    for (int c = 0; c < CC; ++c) up_prev[c] = 0.0f;
    int it = 0, n_iters;
    {
        const auto& b = reload_kernargs<TrajFusedArgs>();
        n_iters = b.n_iters;
    }
    for (; it < n_iters; ++it) {
        float x[D];
        int j0, j1;
        ScoreArgs sa;  // the sweep's view of the model: only what sweep_rows reads
        {
            // ---- forward kinematics of the current waypoints (once per iteration) -----------------------------------
            const auto& b = reload_kernargs<TrajFusedArgs>();
            const TrajLds L = traj_lds<D, CC>(smem, b, nw);
            const int dof = b.sc.dof, d_fk = b.sc.d_fk;
            DhArgs dh;
            DCX_COPY_DH(dh, b.sc.dh);
            FkWalk fw;
            fw.fkk = b.sc.fkk;
            fw.g = b.sc.fk;
            fw.fk = (fk_cptr)(uintptr_t)(uint32_t)(uintptr_t)L.sFk;
            fw.dh = (dh_cptr)(uintptr_t)(uint32_t)(uintptr_t)L.sFk;
            DCX_TTS(0);
            fk_trig_sel(fw, dh, L.sQ + lane * dof, L.sF + lane, wave, nw);
            __syncthreads();
            DCX_TTS(1);
            if (b.sc.jt_waves) dh2_chain_rows_sel(fw.dh, dh, L.sX + lane, L.sF + lane, wave);
            else if (wave == 0) fk_chain_sel(fw, dh, L.sQ + lane * dof, L.sX + lane, L.sF + lane);
            __syncthreads();
            DCX_TTS(2);
            // What needs the features / frames but not the collision gradient goes in front of the sweep, where the other
            // waves' sweeps hide its latency: the path terms on wave 1, phase R1 of J^T on waves 2 ..
            if (b.sc.jt_waves && nw > 1) {
                if (wave == 1) traj_path_terms(b, L, lane);
                else if (wave >= 2) dh2_vjp_r1_sel(fw.dh, dh, L.sF + lane, L.sJ + lane, wave - 2, 15);
            }
            if (d_fk == D) {
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = L.sX[k * 64 + lane];
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = (k < d_fk) ? L.sX[k * 64 + lane] : 0.0f;
            }
            // this wave's slice of the supports (the sweep's slicing: wave w takes [w * s_chunk, (w + 1) * s_chunk))
            if constexpr (CL) {  // ... of this workgroup's share [y * s_super, (y + 1) * s_super) (score_kernel's split slicing)
                const int ybase = ycl * b.s_super;
                const int yend = (ybase + b.s_super < b.sc.S) ? ybase + b.s_super : b.sc.S;
                wave_slice(wave, nw, b.sc.s_chunk, b.sc.s_skew, ybase, yend, j0, j1);
            } else {
                wave_slice(wave, nw, b.sc.s_chunk, b.sc.s_skew, 0, b.sc.S, j0, j1);
            }
            if constexpr (XF) {  // the expanded form works on centred data (score_kernel.h)
                cfloat_ptr cen = (cfloat_ptr)(uintptr_t)b.sc.centre;
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] -= cen[k];
            }
            sa.rows = b.sc.rows;
            sa.kind = b.sc.kind;
            sa.kp0 = b.sc.kp0;
            sa.kp1 = b.sc.kp1;
        }
        // ---- collision sweep: score and feature gradient against this wave's supports -------------------------------
        float sc[CC];
        float gx[D];
        bool grad_folded = false, regrad = false, grad_zero = false;   // (several classes, see SPECULATION)
#pragma unroll
        for (int k = 0; k < D; ++k) gx[k] = 0.0f;
#pragma unroll
        for (int c = 0; c < CC; ++c) sc[c] = 0.0f;
        if constexpr (CC == 1) {
            const float up[1] = {1.0f};
            sweep_rows<D, KF, 1, MODE_GRAD_ROW, XF, DCX_TRAJ_NACC>(sa, x, up, j0, j1, sc, gx);
        } else {
            float up[CC];
            bool resweep = true;   // does the gradient still need a sweep with the indicator of this iteration's scores?
            bool moved = false;    // did the indicator change against the last iteration's?
            auto totals_to_upstream = [&](const auto& b, const TrajLds& L) __attribute__((always_inline)) {
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    const float t = L.sRed[c * 64 + lane];
                    up[c] = (c < b.sc.c_out && t - b.margin_c[c < 8 ? c : 7] > 0.0f) ? b.opt.w_collision : 0.0f;
                }
            };
            // (an all-zero upstream - no class over its margin on any waypoint of the path, the state of a path in free space - is not
            // worth a gradient sweep at all: the scores alone are swept, and only if they put an indicator up is the gradient)
            bool prev_nonzero = false;
#pragma unroll
            for (int c = 0; c < CC; ++c) prev_nonzero = prev_nonzero || (up_prev[c] != 0.0f);
            const bool spec_full = spec && __builtin_amdgcn_ballot_w64(prev_nonzero) != 0;
            if (spec_full) {
                // ONE sweep with the last iteration's upstream: class scores and gradient together
#pragma unroll
                for (int c = 0; c < CC; ++c) up[c] = up_prev[c];
                sweep_rows<D, KF, CC, MODE_GRAD_UP, XF, DCX_TRAJ_NACC>(sa, x, up, j0, j1, sc, gx);
                const auto& b = reload_kernargs<TrajFusedArgs>();
                const TrajLds L = traj_lds<D, CC>(smem, b, nw);
                float* mine = L.sRed + (size_t)wave * ACC * 64 + lane;
#pragma unroll
                for (int c = 0; c < CC; ++c) mine[c * 64] = sc[c];
#pragma unroll
                for (int k = 0; k < D; ++k) mine[(CC + k) * 64] = gx[k];
                __syncthreads();
                if constexpr (CL) {
                    unsigned long long* slot = b.exch + ((size_t)(r * 2 + (it & 1)) * b.ys) * XS * 64 + lane;
                    const uint32_t tag = b.tag_base + (uint32_t)it + 1u;
                    traj_exchange_publish<ACC, 0, ACC, XS, 0>(L.sRed, slot, tag, ycl, wave, lane, nw);
                    if (!traj_exchange_collect<ACC, 0, ACC, XS, 0>(L.sRed, slot, tag, b.ys, wave, lane, nw)) L.sR[kTrajAbort] = 1.0f;
                } else {
                    fold_partial_rows<ACC, 0, ACC>(L.sRed, wave, lane, nw);
                }
                __syncthreads();
                totals_to_upstream(b, L);
                bool differs = false;
#pragma unroll
                for (int c = 0; c < CC; ++c) differs = differs || (up[c] != up_prev[c]);
                resweep = __builtin_amdgcn_ballot_w64(differs) != 0;   // the same on every wave (and workgroup): they read the same totals
                moved = resweep;
            } else {
                // the class scores first ...
#pragma unroll
                for (int c = 0; c < CC; ++c) up[c] = 0.0f;
                sweep_rows<D, KF, CC, MODE_SCORE, XF, DCX_TRAJ_NACC>(sa, x, up, j0, j1, sc, gx);
                const auto& b = reload_kernargs<TrajFusedArgs>();
                const TrajLds L = traj_lds<D, CC>(smem, b, nw);
                float* mine = L.sRed + (size_t)wave * ACC * 64 + lane;
#pragma unroll
                for (int c = 0; c < CC; ++c) mine[c * 64] = sc[c];
                __syncthreads();
                if constexpr (CL) {
                    unsigned long long* slot = b.exch + ((size_t)(r * 2 + (it & 1)) * b.ys) * XS * 64 + lane;
                    const uint32_t tag = b.tag_base + (uint32_t)it + 1u;
                    traj_exchange_publish<ACC, 0, CC, XS, 0>(L.sRed, slot, tag, ycl, wave, lane, nw);
                    if (!traj_exchange_collect<ACC, 0, CC, XS, 0>(L.sRed, slot, tag, b.ys, wave, lane, nw)) L.sR[kTrajAbort] = 1.0f;
                } else {
                    fold_partial_rows<ACC, 0, CC>(L.sRed, wave, lane, nw);
                }
                __syncthreads();
                totals_to_upstream(b, L);
                // (was the indicator the same as one iteration ago?  Only then is the next iteration worth a speculation)
                bool differs = false;
#pragma unroll
                for (int c = 0; c < CC; ++c) differs = differs || (up[c] != up_prev[c]);
                moved = __builtin_amdgcn_ballot_w64(differs) != 0;
                bool nonzero = false;
#pragma unroll
                for (int c = 0; c < CC; ++c) nonzero = nonzero || (up[c] != 0.0f);
                resweep = __builtin_amdgcn_ballot_w64(nonzero) != 0;
                grad_zero = !resweep;
            }
            if (resweep) {
                // ... then the gradient of w_collision * sum_c clamp(score_c - margin_c, 0) with its upstream per lane and class (the
                // score totals stay in row 0 of the scratch: this sweep's partial rows only use the gradient columns)
#pragma unroll
                for (int c = 0; c < CC; ++c) sc[c] = 0.0f;
#pragma unroll
                for (int k = 0; k < D; ++k) gx[k] = 0.0f;
                sweep_rows<D, KF, CC, MODE_GRAD_UP, XF, DCX_TRAJ_NACC, false, true>(sa, x, up, j0, j1, sc, gx);   // (NS: the scores are known)
            }
            grad_folded = !resweep;             // a speculation that held (or no gradient at all): the totals are in row 0 already
            regrad = spec_full && resweep;      // a speculation that failed: its second gradient travels through words of its own
#pragma unroll
            for (int c = 0; c < CC; ++c) up_prev[c] = up[c];
            spec = !moved;                      // speculate only behind an iteration whose indicator stood still: a path whose
                                                // indicators keep moving stays with two sweeps and pays nothing for the attempt
        }
        DCX_TTS(3);
        {
            const auto& b = reload_kernargs<TrajFusedArgs>();
            const TrajLds L = traj_lds<D, CC>(smem, b, nw);
            const int W = b.st.n_waypoints, dof = b.sc.dof, d_fk = b.sc.d_fk;
            const bool live = w < W;
            const bool jt = b.sc.jt_waves != 0;
            DhArgs dh;
            DCX_COPY_DH(dh, b.sc.dh);
            FkWalk fw;
            fw.fkk = b.sc.fkk;
            fw.g = b.sc.fk;
            fw.fk = (fk_cptr)(uintptr_t)(uint32_t)(uintptr_t)L.sFk;
            fw.dh = (dh_cptr)(uintptr_t)(uint32_t)(uintptr_t)L.sFk;
            const bool tree = fk_is_tree(fw);  // its reverse sweep keeps adjoint sums in the frames: one at a time
            float col = 0.f;
            const int pwave = (nw > 1 && !tree) ? 1 : 0;
            auto path_terms = [&]() __attribute__((always_inline)) { traj_path_terms(b, L, lane); };
            constexpr int E0 = CC > 1 ? CC : 0;   // (several classes: the scores' totals already sit in row 0)
            if (CC > 1 && grad_zero) {
                // no class over its margin anywhere on the path: the gradient's totals are zero (what the sweep would have summed:
                // every coefficient is g x 0)
                for (int e = wave; e < D; e += nw) L.sRed[(CC + e) * 64 + lane] = 0.0f;
                __syncthreads();
            }
            if ((CL || nw > 1 || CC > 1) && !grad_folded) {
                // the sweep's parallel cross-wave fold (score_kernel.h): row 0 first, then 1, 2, ...
                float* mine = L.sRed + (size_t)wave * ACC * 64 + lane;
                if constexpr (CC == 1) mine[0] = sc[0];
#pragma unroll
                for (int k = 0; k < D; ++k) mine[(CC + k) * 64] = gx[k];
                __syncthreads();
                DCX_TTS(4);
                if constexpr (CL) {
                    unsigned long long* slot = b.exch + ((size_t)(r * 2 + (it & 1)) * b.ys) * XS * 64 + lane;
                    const uint32_t tag = b.tag_base + (uint32_t)it + 1u;
                    if (CC > 1 && regrad) {
                        traj_exchange_publish<ACC, E0, ACC, XS, D>(L.sRed, slot, tag, ycl, wave, lane, nw);
                        if (!traj_exchange_collect<ACC, E0, ACC, XS, D>(L.sRed, slot, tag, b.ys, wave, lane, nw)) L.sR[kTrajAbort] = 1.0f;
                    } else {
                        traj_exchange_publish<ACC, E0, ACC, XS, 0>(L.sRed, slot, tag, ycl, wave, lane, nw);
                        if (!traj_exchange_collect<ACC, E0, ACC, XS, 0>(L.sRed, slot, tag, b.ys, wave, lane, nw)) L.sR[kTrajAbort] = 1.0f;
                    }
                } else {
                fold_partial_rows<ACC, E0, ACC>(L.sRed, wave, lane, nw);
                }
                DCX_TTS(5);
                DCX_TTS(6);  // (the path terms and phase R1 of J^T ran in front of the sweep)
                __syncthreads();
                DCX_TTS(7);
            }
            if (jt) {
                // ---- both J^T products on several waves (fk_device.h): l = R^T g per point step for the collision gradient
                // (the folded totals x the hinge factor, read in place) and for the path gradient; then the two wrench
                // recurrences side by side, chain by chain: collision on waves 0 (, 1), path on waves 2 (, 3) ----------
                float scale = 1.0f;   // (several classes: the hinge went into the second sweep's upstream)
                if constexpr (CC == 1) {
                    const float s0 = L.sRed[lane] - b.opt.safety_margin;
                    scale = (s0 > 0.0f) ? b.opt.w_collision : 0.0f;
                    if (wave == 0 && live && s0 > 0.f) col = s0;
                } else if (wave == 0 && live) {
#pragma unroll
                    for (int c = 0; c < CC; ++c) {
                        const float e = L.sRed[c * 64 + lane] - b.margin_c[c < 8 ? c : 7];
                        if (c < b.sc.c_out && e > 0.f) col += e;
                    }
                }
                const int half = nw >> 1;
                if (wave < half) dh2_vjp_r1b(fw.dh, dh, L.sRed + CC * 64 + lane, scale, L.sJ + lane, L.sGQc + lane * dof, dof, wave, half, 15, 9);
                else dh2_vjp_r1b(fw.dh, dh, L.sGp + lane, 1.0f, L.sJ + lane, L.sGQp + lane * dof, dof, wave - half, nw - half, 15, 12);
                __syncthreads();
                DCX_TTS(8);
                if (wave < 2) dh2_vjp_r2_sel(fw.dh, dh, L.sF + lane, L.sJ + lane, L.sGQc + lane * dof, wave, 15, 9);
                else dh2_vjp_r2_sel(fw.dh, dh, L.sF + lane, L.sJ + lane, L.sGQp + lane * dof, wave - 2, 15, 12);
            } else {
                // ---- wave 0: hinge + J^T of the collision gradient;  wave 1: path terms + their J^T --------------------
                if (wave == 0) {
                    if (nw > 1 || CC > 1) {
#pragma unroll
                        for (int c = 0; c < CC; ++c) sc[c] = L.sRed[c * 64 + lane];
#pragma unroll
                        for (int k = 0; k < D; ++k) gx[k] = L.sRed[(CC + k) * 64 + lane];
                    }
                    const float scale = CC > 1 ? 1.0f : ((sc[0] - b.opt.safety_margin > 0.0f) ? b.opt.w_collision : 0.0f);
#pragma unroll
                    for (int k = 0; k < D; ++k)
                        if (k < d_fk) L.sGc[k * 64 + lane] = gx[k] * scale;
                    // q row -> gq row in a separate buffer (the two-launch form overwrites the q row; same arithmetic)
                    for (int i = 0; i < dof; ++i) L.sGQc[lane * dof + i] = L.sQ[lane * dof + i];
                    fk_vjp_sel(fw, dh, L.sGQc + lane * dof, L.sF + lane, L.sGc + lane, L.sGQc + lane * dof, dof);
                    if constexpr (CC == 1) {
                        const float s0 = sc[0] - b.opt.safety_margin;
                        if (live && s0 > 0.f) col = s0;
                    } else if (live) {
#pragma unroll
                        for (int c = 0; c < CC; ++c) {
                            const float e = sc[c] - b.margin_c[c < 8 ? c : 7];
                            if (c < b.sc.c_out && e > 0.f) col += e;
                        }
                    }
                }
                if (wave == pwave) {
                    path_terms();
                    fk_vjp_sel(fw, dh, L.sQ + lane * dof, L.sF + lane, L.sGp + lane, L.sGQp + lane * dof, dof);
                }
            }
            if (wave == 0) {  // the lane's hinge excess, for the loss terms below
                const float t_col = traj_wave_sum(col);
                if (lane == 0) L.sR[17] = t_col;
            }
            DCX_TTS(9);
        }
        __syncthreads();
        DCX_TTS(10);
        int flags;
        {
            // ---- joint limits, endpoint mask, Adam: joint i on wave i % nw (the joints are independent; the two sums that
            // run over them - limit excess, |g|^2 - are put together afterwards in the order i = 0, 1, ... a single wave would
            // use, so the loss terms do not change by a bit) ----------------------------------------------------------------
            const auto& b = reload_kernargs<TrajFusedArgs>();
            const TrajLds L = traj_lds<D, CC>(smem, b, nw);
            const int W = b.st.n_waypoints, dof = b.sc.dof;
            const bool live = w < W;
            const bool endpoint = (w == 0) || (w == W - 1);
            float* sJl = L.sJl;
            for (int i = wave; i < dof; i += nw) {
                float jl_i = 0.f, g2_i = 0.f;
                if (live) {
                    const float b1 = b.bias1[it], b2s = b.bias2_sqrt[it];
                    const float q = L.sQ[lane * dof + i];
                    const float lo = L.sR[kTrajLim + 2 * i], hi = L.sR[kTrajLim + 2 * i + 1];
                    float g = L.sGQp[lane * dof + i] + L.sGQc[lane * dof + i];
                    if (q < lo) { jl_i = lo - q; g -= b.opt.w_joint_limit; }
                    if (q > hi) { jl_i += q - hi; g += b.opt.w_joint_limit; }
                    if (endpoint) g = 0.f;  // p.grad[[0, -1]] = 0 (optim.py:102)
                    g2_i = g;
                    float m = L.sM[lane * dof + i], v = L.sV[lane * dof + i];
                    m = fmaf(b.opt.beta1, m, (1.f - b.opt.beta1) * g);
                    v = fmaf(b.opt.beta2, v, (1.f - b.opt.beta2) * g * g);
                    const float denom = sqrtf(v) / b2s + b.opt.eps;
                    const float qn = traj_adam_q(q, b.opt.lr, b1, m, denom);
                    L.sM[lane * dof + i] = m;
                    L.sV[lane * dof + i] = v;
                    L.sQ[lane * dof + i] = qn;
                }
                sJl[(2 * i) * 64 + lane] = jl_i;
                sJl[(2 * i + 1) * 64 + lane] = g2_i;
            }
        }
        __syncthreads();
        {
            const auto& b = reload_kernargs<TrajFusedArgs>();
            const TrajLds L = traj_lds<D, CC>(smem, b, nw);
            const int dof = b.sc.dof;
            if (wave == 0) {
                const float* sJl = L.sJl;
                float jl = 0.f, gn2 = 0.f;
                for (int i = 0; i < dof; ++i) {
                    // the one-wave form's order: a joint below its lower limit adds lo - q, above its upper limit q - hi
                    jl += sJl[(2 * i) * 64 + lane];
                    const float g = sJl[(2 * i + 1) * 64 + lane];
                    gn2 = fmaf(g, g, gn2);
                }
                const float t_jl = traj_wave_sum(jl), t_gn2 = traj_wave_sum(gn2);
                if (lane == 0) {
                    // the step kernel's totals: 0 + (one per-wave partial)
                    const float tot0 = 0.f + L.sR[0], tot1 = 0.f + L.sR[16], tot2 = 0.f + t_jl, tot3 = 0.f + L.sR[17], tot4 = 0.f + t_gn2;
                    const float objective = b.opt.w_diff * tot0;
                    const float constraint = traj_constraint(b.opt.w_collision, tot3, b.opt.w_max_move, tot1, b.opt.w_joint_limit, tot2);
                    const float loss = objective + constraint;
                    const float gnorm = sqrtf(tot4);
                    float* st = L.sR + kTrajStats;   // (to global memory once, after the last iteration)
                    st[0] = loss; st[1] = objective; st[2] = constraint; st[3] = gnorm; st[4] = tot3; st[5] = tot1; st[6] = tot2;
                    st[7] = 0.f;
                    int fl = 0;
                    if (loss < L.sR[kTrajLowest]) {  // optim.py:107-112 (solution = p AFTER the step)
                        L.sR[kTrajLowest] = loss;
                        L.sR[kTrajLowestObj] = objective;
                        fl |= 1;
                    }
                    if (constraint <= b.opt.valid_tol) {  // optim.py:113-118
                        if (objective < L.sR[kTrajBestValid]) {
                            L.sR[kTrajBestValid] = objective;
                            fl |= 2;
                        }
                        if (gnorm < b.opt.grad_tol) fl |= 4;  // optim.py:126-127: the path stops here
                    }
                    L.sR[kTrajSteps] = __int_as_float(__float_as_int(L.sR[kTrajSteps]) + 1);
                    L.sR[kTrajFlags] = __int_as_float(fl);
                }
            }
        }
        __syncthreads();
        {
            const auto& b = reload_kernargs<TrajFusedArgs>();
            const TrajLds L = traj_lds<D, CC>(smem, b, nw);
            const int W = b.st.n_waypoints, dof = b.sc.dof;
            flags = __float_as_int(L.sR[kTrajFlags]);
            if constexpr (CL) {
                if (L.sR[kTrajAbort] != 0.0f) flags = 8;  // an exchange gave up in this workgroup: leave, state untouched
            }
            if ((flags & 3) && (!CL || ycl == 0)) {
                float* lo = b.st.lowest_path + (size_t)r * W * dof;
                float* bv = b.st.best_valid_path + (size_t)r * W * dof;
                for (int i = tid; i < W * dof; i += blockDim.x) {
                    const float v = L.sQ[i];
                    if (flags & 1) lo[i] = v;
                    if (flags & 2) bv[i] = v;
                }
            }
            DCX_TTS(11);
            // lanes past the path follow its last row (their FK feeds nothing, but keep them finite and in step)
            for (int i = W * dof + tid; i < 64 * dof; i += blockDim.x) L.sQ[i] = L.sQ[(W - 1) * dof + (i % dof)];
        }
        __syncthreads();
        if constexpr (CL) {
            if (flags & 8) {
                if (tid == 0) a.st.stats[(size_t)r * 8 + 7] = -1.0f;
                return;
            }
        }
        if (flags & 4) {
            if (tid == 0 && (!CL || ycl == 0)) a.st.done[r] = 1;
            ++it;
            break;
        }
    }
    (void)it;
    if constexpr (CL) {
        if (ycl != 0) return;  // workgroup 0 of the cluster writes the state back
    }
    // ---- state back to HBM ------------------------------------------------------------------------------------------
    {
        const auto& b = reload_kernargs<TrajFusedArgs>();
        const TrajLds L = traj_lds<D, CC>(smem, b, nw);
        const int W = b.st.n_waypoints, dof = b.sc.dof;
        const size_t base = (size_t)r * W * dof;
        for (int i = tid; i < W * dof; i += blockDim.x) {
            b.st.path[base + i] = L.sQ[i];
            b.st.adam_m[base + i] = L.sM[i];
            b.st.adam_v[base + i] = L.sV[i];
        }
        if (tid < 8) b.st.stats[(size_t)r * 8 + tid] = L.sR[kTrajStats + tid];
        if (tid == 0) {
            b.st.lowest_loss[r] = L.sR[kTrajLowest];
            b.st.lowest_obj[r] = L.sR[kTrajLowestObj];
            b.st.best_valid_obj[r] = L.sR[kTrajBestValid];
            b.st.steps[r] = __float_as_int(L.sR[kTrajSteps]);
        }
    }
}

}  // namespace dcx
