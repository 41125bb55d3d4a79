// traj_kernels.hip — one fused Adam step on R waypoint paths.
//
// Batched restatement of the loop body of the reference's adam_traj_optimize (optim.py:86-127): the collision
// term (score and hinge gradient) comes from the fused score kernel; everything else of the step — the FK of
// the waypoints, the path-length and max-move terms that couple neighbouring waypoints, the joint-limit term,
// J^T, endpoint masking, the Adam update and the best-so-far bookkeeping — happens here, in one launch, with
// no host synchronisation (the reference syncs every iteration through .data.numpy(), optim.py:107-118).
//
// Mapping: one block per path, one lane per waypoint (W <= 1024); a wave owns 64 consecutive waypoints and
// keeps their control points in its own LDS slab, so a waypoint reads its neighbours' control points from LDS.
#include <cmath>

#include "dcx_internal.h"

namespace dcx {
namespace {

struct TrajArgs {
    const FkProg* fk;
    dcx_traj_state st;
    dcx_traj_opts opt;
    int32_t step;
    int32_t dof, d_fk, n_points, point_dim, frame_floats;
    int32_t coord_major;  // features laid out [point_dim][n_points] (DCX_FK_TREE, t_coord_major) instead of [n_points][point_dim]
    float bias1, bias2_sqrt;  // 1 - beta1^t, sqrt(1 - beta2^t)
    int32_t n_class;          // columns of col_score: [R*W, n_class] (several classes: a MultiDiffCo score under per-class margins)
    float margin_c[8];        // ... and their margins (n_class == 1: opt.safety_margin)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(1024) void traj_adam_step_kernel(const TrajArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int r = blockIdx.x;
    if (a.st.done[r]) return;  // frozen path
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int W = a.st.n_waypoints, dof = a.dof, D = a.d_fk;
    const int w = tid;                       // this lane's waypoint
    const bool live = w < W;

    // LDS carve: q rows [nw*64][dof] | gq rows | per-wave slabs X, G, F | reduction scratch | program (variable size)
    float* sQ = smem;
    float* sGQ = sQ + nw * 64 * dof;
    float* sX = sGQ + nw * 64 * dof;
    float* sG = sX + nw * 64 * D;
    float* sF = sG + nw * 64 * D;
    float* sR = sF + nw * 64 * a.frame_floats;  // [6][16] partial sums + flags
    float* sP = sR + 128;

    const fk_cptr fk = stage_fk_prog(a.fk, sP, tid, blockDim.x);
    const float* path = a.st.path + (size_t)r * W * dof;
    for (int i = tid; i < nw * 64 * dof; i += blockDim.x) sQ[i] = path[i < W * dof ? i : (i % dof) + (W - 1) * dof];
    __syncthreads();

    float* myQ = sQ + (wave * 64 + lane) * dof;
    float* myX = sX + wave * 64 * D + lane;
    float* myG = sG + wave * 64 * D + lane;
    float* myF = sF + wave * 64 * a.frame_floats + lane;
    fk_forward_trig(fk, myQ, myF, 0, 1);
    fk_forward_chain(fk, myQ, myX, myF);
    __syncthreads();

    // control point coordinate k of waypoint v
    auto X = [&](int k, int v) { return sX[(v >> 6) * 64 * D + k * 64 + (v & 63)]; };

    // ---- path-length and max-move terms: gradient w.r.t. this waypoint's control points ---------------
    const float ms = a.opt.max_speed;
    float obj = 0.f, mmv = 0.f;
    const int pd = a.point_dim;
    for (int p = 0; p < a.n_points; ++p) {
        float dn[3] = {0.f, 0.f, 0.f}, dp[3] = {0.f, 0.f, 0.f};
        float n2n = 0.f, n2p = 0.f;
        for (int c = 0; c < pd; ++c) {
            const int k = a.coord_major ? c * a.n_points + p : p * pd + c;
            const float xc = live ? X(k, w) : 0.f;
            if (live && w + 1 < W) { dn[c] = X(k, w + 1) - xc; n2n = fmaf(dn[c], dn[c], n2n); }
            if (live && w >= 1)    { dp[c] = xc - X(k, w - 1); n2p = fmaf(dp[c], dp[c], n2p); }
        }
        const float mn = traj_excess(n2n, ms), mp = traj_excess(n2p, ms);
        if (live && w + 1 < W) {   // each segment is counted once, by its left waypoint
            obj += n2n;
            if (mn > 0.f) mmv += mn;
        }
        const float cn = 2.f * (a.opt.w_diff + (mn > 0.f ? a.opt.w_max_move : 0.f));
        const float cp = 2.f * (a.opt.w_diff + (mp > 0.f ? a.opt.w_max_move : 0.f));
        for (int c = 0; c < pd; ++c) myG[(a.coord_major ? c * a.n_points + p : p * pd + c) * 64] = traj_path_grad(cp, dp[c], cn, dn[c]);
    }
    // J^T of that gradient (per lane; frames of this lane are in its slab)
    float* myGQ = sGQ + (wave * 64 + lane) * dof;
    fk_vjp(fk, myQ, myF, myG, myGQ);

    // ---- joint limits, collision gradient, endpoint mask, Adam ----------------------------------------
    float jl = 0.f, gn2 = 0.f, col = 0.f;
    if (live) {
        const size_t base = ((size_t)r * W + w) * dof;
        const bool endpoint = (w == 0) || (w == W - 1);
        // collision term of this waypoint: sum_c clamp(score_c - margin_c, 0) (optim.py:88-89; one class: one term)
        for (int c = 0; c < a.n_class; ++c) {
            const float sc = a.st.col_score[((size_t)r * W + w) * a.n_class + c] - a.margin_c[c];
            if (sc > 0.f) col += sc;
        }
        for (int i = 0; i < dof; ++i) {
            const float q = myQ[i];
            const float lo = a.st.limits[2 * i], hi = a.st.limits[2 * i + 1];
            float g = myGQ[i] + a.st.col_grad[base + i];
            if (q < lo) { jl += lo - q; g -= a.opt.w_joint_limit; }
            if (q > hi) { jl += q - hi; g += a.opt.w_joint_limit; }
            if (endpoint) g = 0.f;  // p.grad[[0, -1]] = 0 (optim.py:102)
            gn2 = fmaf(g, g, gn2);
            float m = a.st.adam_m[base + i], v = a.st.adam_v[base + i];
            m = fmaf(a.opt.beta1, m, (1.f - a.opt.beta1) * g);
            v = fmaf(a.opt.beta2, v, (1.f - a.opt.beta2) * g * g);
            const float denom = sqrtf(v) / a.bias2_sqrt + a.opt.eps;
            const float qn = traj_adam_q(q, a.opt.lr, a.bias1, m, denom);
            a.st.adam_m[base + i] = m;
            a.st.adam_v[base + i] = v;
            a.st.path[base + i] = qn;
            myQ[i] = qn;  // keep the new row for the bookkeeping copies below
        }
    }

    // ---- block sums -> loss terms ------------------------------------------------------------------------
    float part[5] = {obj, mmv, jl, col, gn2};
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float s = wave_sum(part[t]);
        if (lane == 0) sR[t * 16 + wave] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float tot[5];
        for (int t = 0; t < 5; ++t) {
            float s = 0.f;
            for (int k = 0; k < nw; ++k) s += sR[t * 16 + k];
            tot[t] = s;
        }
        const float objective = a.opt.w_diff * tot[0];
        const float constraint = traj_constraint(a.opt.w_collision, tot[3], a.opt.w_max_move, tot[1], a.opt.w_joint_limit, tot[2]);
        const float loss = objective + constraint;
        const float gnorm = sqrtf(tot[4]);
        float* st = a.st.stats + (size_t)r * 8;
        st[0] = loss; st[1] = objective; st[2] = constraint; st[3] = gnorm; st[4] = tot[3]; st[5] = tot[1]; st[6] = tot[2];
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
            if (gnorm < a.opt.grad_tol) a.st.done[r] = 1;  // optim.py:126-127
        }
        a.st.steps[r] += 1;
        sR[96] = __int_as_float(flags);
    }
    __syncthreads();
    const int flags = __float_as_int(sR[96]);
    if (flags) {
        float* lo = a.st.lowest_path + (size_t)r * W * dof;
        float* bv = a.st.best_valid_path + (size_t)r * W * dof;
        for (int i = tid; i < W * dof; i += blockDim.x) {
            const float v = sQ[i];
            if (flags & 1) lo[i] = v;
            if (flags & 2) bv[i] = v;
        }
    }
}

}  // namespace

size_t traj_lds_bytes(const dcx_fk_desc& fk, int nw) {
    const int d_fk = fk.n_points * fk.point_dim;
    return sizeof(float) * (fk_prog_floats(fk) + 2 * nw * 64 * fk.dof + 2 * nw * 64 * d_fk + nw * 64 * fk_frame_floats(fk) + 128);
}

hipError_t launch_traj_adam_step(const FkProg* fk_dev, const dcx_fk_desc& fk, const dcx_traj_state& st,
                                 const dcx_traj_opts& opt, int step, hipStream_t stream, int n_class, const float* margin_c) {
    if (st.n_paths == 0) return hipSuccess;
    TrajArgs a;
    a.n_class = n_class;
    for (int c = 0; c < 8; ++c) a.margin_c[c] = (margin_c && c < n_class) ? margin_c[c] : opt.safety_margin;
    a.fk = fk_dev;
    a.st = st;
    a.opt = opt;
    a.step = step;
    a.dof = fk.dof;
    a.d_fk = fk.n_points * fk.point_dim;
    a.n_points = fk.n_points;
    a.point_dim = fk.point_dim;
    a.coord_major = (fk.kind == DCX_FK_TREE && fk.t_coord_major) ? 1 : 0;
    a.frame_floats = fk_frame_floats(fk);
    a.bias1 = (float)(1.0 - pow((double)opt.beta1, (double)step));
    a.bias2_sqrt = (float)sqrt(1.0 - pow((double)opt.beta2, (double)step));
    const int nw = (st.n_waypoints + 63) / 64;
    const size_t lds = traj_lds_bytes(fk, nw);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)traj_adam_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    traj_adam_step_kernel<<<dim3((unsigned)st.n_paths), dim3(64 * nw), lds, stream>>>(a);
    return hipGetLastError();
}

// ---- escape from collision: the update half of dcx_escape_adam (include/dcx.h) ------------------------------------------------
// The loop of OptimSampler.optim_escape (scripts/escape.py:19-38) with its decisions kept on the device: the fused sweep has
// written score [B, C] and grad [B, dof] of THIS step (gradient of sum_c score_c); one lane = one configuration decides
// whether its loop goes on, records, takes the Adam step (the arithmetic of traj_adam_step_kernel above) and wraps.
//   steps[loop][0]   evaluations of dist_est so far      steps[loop][1]   Adam steps taken so far
// A loop that goes on takes one step per evaluation, a loop that stops has evaluated once more than it stepped: "stopped" is
// steps[loop][0] != steps[loop][1], and no other flag is kept (joint form: one loop, index 0).
// Step 0 INITIALISES: every loop is alive there, its counters and Adam moments are taken as zero without being read, so the
// caller's buffers need no memset, and a loop's final record is written where the loop stops or takes the call's last step
// (no pass afterwards): a three-step escape is six launches.
namespace {

__device__ __forceinline__ float escape_wrap2pi(float q) {  // utils.py:51-52 on fp32 tensors: (pi + q) % (2 pi) - pi, Python's %
    const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
    float r = fmodf(pi + q, two_pi);
    if (r != 0.f && r < 0.f) r += two_pi;
    return r - pi;
}

// joint form, before the update: ONE workgroup sums score - margin over the whole batch (escape.py:26) and takes the loop's
// decision, so that the update launch only reads it
// the whole batch's excess (escape.py:26), summed in double by one workgroup of 1024 lanes; every lane returns the sum
__device__ __forceinline__ double escape_total_excess(const EscapeArgs& a, double* part) {
    double acc = 0.0;
    const int64_t n = a.B * a.C;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const float mg = a.margin ? a.margin[e % a.C] : 0.f;
        acc += (double)(a.score[e] - mg);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    return part[0];
}

// a loop's final configuration into the slot behind its last record (escape.py:37): written by the lane that sees the loop
// stop, or take the call's last step - no pass over the batch afterwards
__device__ __forceinline__ void escape_final_record(const EscapeArgs& a, int64_t b, int updates) {
    if (!a.history) return;
    const int slot = a.record_freq > 0 ? (updates + a.record_freq - 1) / a.record_freq : 0;
    float* h = a.history + ((int64_t)slot * a.B + b) * a.dof;
    for (int k = 0; k < a.dof; ++k) h[k] = a.q[b * a.dof + k];
}

// configuration b (row i of the sweep's batch) of a loop that goes on: record, Adam step (the arithmetic of
// traj_adam_step_kernel), wrap
__device__ __forceinline__ void escape_row_step(const EscapeArgs& a, int64_t b, int64_t i, int updates) {
    const int dof = a.dof;
    float* q = a.q + b * dof;
    if (a.history && a.record_freq > 0 && a.step % a.record_freq == 0) {
        float* h = a.history + ((int64_t)(a.step / a.record_freq) * a.B + b) * dof;
        for (int k = 0; k < dof; ++k) h[k] = q[k];
    }
    for (int k = 0; k < dof; ++k) {
        const float g = a.grad[i * dof + k];
        float m = 0.f, v = 0.f;
        if (a.step > 0) { m = a.adam_m[b * dof + k]; v = a.adam_v[b * dof + k]; }
        m = fmaf(a.beta1, m, (1.f - a.beta1) * g);
        v = fmaf(a.beta2, v, (1.f - a.beta2) * g * g);
        const float denom = sqrtf(v) / a.bias2_sqrt + a.eps;
        float qn = traj_adam_q(q[k], a.lr, a.bias1, m, denom);
        if ((a.wrap_mask >> k) & 1ull) qn = escape_wrap2pi(qn);
        a.adam_m[b * dof + k] = m;
        a.adam_v[b * dof + k] = v;
        q[k] = qn;
        if (a.qa) a.qa[i * dof + k] = qn;   // the dense copy the next sweep reads
    }
    if (a.last) escape_final_record(a, b, updates);
}

// joint form, before the update: ONE workgroup sums score - margin over the whole batch and takes the loop's decision, so that
// the update launch only reads it
__global__ __launch_bounds__(1024) void escape_decide_kernel(const EscapeArgs a) {
    __shared__ double part[1024];
    const int ev = a.step > 0 ? a.steps[0] : 0, up = a.step > 0 ? a.steps[1] : 0;
    if (ev != up) return;  // workgroup-uniform: stopped in an earlier step
    const double excess = escape_total_excess(a, part);
    if (threadIdx.x == 0) {
        a.steps[0] = ev + 1;
        a.steps[1] = up + (excess > 0.0 ? 1 : 0);  // the update launch behind this one takes the step
    }
}

// joint form, B <= 1024: decision and update in the same workgroup (same sums, same steps as the two launches); the launch
// sizes the workgroup by the batch (one wave for the usual single configuration)
__global__ __launch_bounds__(1024) void escape_joint_small_kernel(const EscapeArgs a) {
    __shared__ double part[1024];
    const int ev = a.step > 0 ? a.steps[0] : 0, up = a.step > 0 ? a.steps[1] : 0;
    if (ev != up) return;
    const double excess = escape_total_excess(a, part);
    __syncthreads();                       // everybody has read steps[] and part[0]
    if (threadIdx.x == 0) {
        a.steps[0] = ev + 1;
        a.steps[1] = up + (excess > 0.0 ? 1 : 0);
    }
    if ((int64_t)threadIdx.x < a.B) {
        if (excess > 0.0) escape_row_step(a, threadIdx.x, threadIdx.x, up + 1);
        else escape_final_record(a, threadIdx.x, up);     // the loop stops here
    }
}

// joint form of a handful of configurations (B * dof and B * C <= 64: the usual single configuration): ONE wave, lane = one
// coordinate.  Everything the step may need is requested up front (one round trip instead of counters -> scores -> moments in
// turn: a lone wave pays ~1 us per dependent global load), the excess is a wave reduction, no LDS, no barrier.  Same decision
// and the same Adam arithmetic per coordinate as the kernels above.
__global__ __launch_bounds__(64) void escape_joint_wave_kernel(const EscapeArgs a) {
    const int l = threadIdx.x;
    const int nE = (int)a.B * a.dof, nS = (int)a.B * a.C;
    int ev = 0, up = 0;
    if (a.step > 0) { ev = a.steps[0]; up = a.steps[1]; }
    const float sc = l < nS ? a.score[l] - (a.margin ? a.margin[l % a.C] : 0.f) : 0.f;
    float g = 0.f, m = 0.f, v = 0.f, q = 0.f;
    if (l < nE) {
        g = a.grad[l];
        q = a.q[l];
        if (a.step > 0) { m = a.adam_m[l]; v = a.adam_v[l]; }
    }
    if (ev != up) return;                  // stopped in an earlier step
    double ex = (double)sc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ex += __shfl_xor(ex, o, 64);
    const bool go = ex > 0.0;
    if (l == 0) {
        a.steps[0] = ev + 1;
        a.steps[1] = up + (go ? 1 : 0);
    }
    if (l >= nE) return;
    const int rf = a.record_freq;
    auto slot_of = [&](int updates) { return rf > 0 ? (updates + rf - 1) / rf : 0; };
    if (!go) {                             // the loop stops here: its final record (escape.py:37)
        if (a.history) a.history[(int64_t)slot_of(up) * nE + l] = q;
        return;
    }
    if (a.history && rf > 0 && a.step % rf == 0) a.history[(int64_t)(a.step / rf) * nE + l] = q;
    m = fmaf(a.beta1, m, (1.f - a.beta1) * g);
    v = fmaf(a.beta2, v, (1.f - a.beta2) * g * g);
    const float denom = sqrtf(v) / a.bias2_sqrt + a.eps;
    float qn = traj_adam_q(q, a.lr, a.bias1, m, denom);
    if ((a.wrap_mask >> (l % a.dof)) & 1ull) qn = escape_wrap2pi(qn);
    a.adam_m[l] = m;
    a.adam_v[l] = v;
    a.q[l] = qn;
    if (a.last && a.history) a.history[(int64_t)slot_of(up + 1) * nE + l] = qn;
}

// lane i of the sweep's batch is configuration `row` of the caller's (a.idx: the loops still running after a compaction)
__global__ __launch_bounds__(256) void escape_update_kernel(const EscapeArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_act) return;
    const int64_t b = a.idx ? a.idx[i] : i;
    int up;
    if (a.joint) {
        // escape_decide_kernel ran in front of this launch: it counted this evaluation, and this step's Adam step iff the loop
        // goes on (the loop is a step behind its evaluations from the moment it stops)
        const int ev = a.steps[0];
        up = a.steps[1];
        if (ev != a.step + 1) return;                 // stopped in an earlier step
        if (up != a.step + 1) {                       // stops here
            escape_final_record(a, b, up);
            return;
        }
    } else {
        const int ev = a.step > 0 ? a.steps[2 * b] : 0;
        up = a.step > 0 ? a.steps[2 * b + 1] : 0;
        if (ev != up) return;
        float excess = 0.f;
        for (int c = 0; c < a.C; ++c) excess += a.score[i * a.C + c] - (a.margin ? a.margin[c] : 0.f);
        a.steps[2 * b] = ev + 1;
        a.steps[2 * b + 1] = up + (excess > 0.f ? 1 : 0);
        if (excess <= 0.f) {
            escape_final_record(a, b, up);            // stops here
            return;
        }
        up += 1;
    }
    escape_row_step(a, b, i, up);
}

// the loops still running, taken out of [0, n_in) into a dense list (order as the waves arrive: a configuration's arithmetic
// does not depend on its place in the batch) with their current configurations side by side for the next sweeps
__global__ __launch_bounds__(256) void escape_compact_kernel(const EscapeArgs a, const int32_t* idx_in, int64_t n_in, int32_t* idx_out,
                                                             float* qa_out, int32_t* count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int64_t b = 0;
    bool alive = false;
    if (i < n_in) {
        b = idx_in ? idx_in[i] : i;
        alive = a.steps[2 * b] == a.steps[2 * b + 1];
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(alive);
    int base = 0;
    if (lane == 0 && mask) base = atomicAdd(count, __builtin_popcountll(mask));
    base = __builtin_amdgcn_readfirstlane(base);
    if (alive) {
        const int j = base + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        idx_out[j] = (int32_t)b;
        for (int k = 0; k < a.dof; ++k) qa_out[(int64_t)j * a.dof + k] = a.q[b * a.dof + k];
    }
}

}  // namespace

hipError_t launch_escape_step(EscapeArgs a, int step, bool last, hipStream_t stream) {
    a.step = step;
    a.last = last ? 1 : 0;
    a.bias1 = (float)(1.0 - pow((double)a.beta1, (double)(step + 1)));
    a.bias2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, (double)(step + 1)));
    if (a.joint && a.B * a.dof <= 64 && a.B * a.C <= 64) {   // the usual call (one configuration): one wave, one round trip
        escape_joint_wave_kernel<<<1, 64, 0, stream>>>(a);
        return hipGetLastError();
    }
    if (a.joint && a.B <= 1024) {   // decision and update in one workgroup, one launch
        const int64_t n = a.B * a.C > a.B ? a.B * a.C : a.B;
        escape_joint_small_kernel<<<1, n <= 64 ? 64 : n <= 256 ? 256 : 1024, 0, stream>>>(a);
        return hipGetLastError();
    }
    if (a.joint) {
        escape_decide_kernel<<<1, 1024, 0, stream>>>(a);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    escape_update_kernel<<<dim3((unsigned)((a.n_act + 255) / 256)), 256, 0, stream>>>(a);
    return hipGetLastError();
}

hipError_t launch_escape_compact(const EscapeArgs& a, const int32_t* idx_in, int64_t n_in, int32_t* idx_out, float* qa_out,
                                 int32_t* count, hipStream_t stream) {
    escape_compact_kernel<<<dim3((unsigned)((n_in + 255) / 256)), 256, 0, stream>>>(a, idx_in, n_in, idx_out, qa_out, count);
    return hipGetLastError();
}


}  // namespace dcx
