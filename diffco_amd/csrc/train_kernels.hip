// train_kernels.hip — the kernel-perceptron trainer as ONE persistent workgroup.
//
// Restates DiffCo.train_perceptron (kernel_perceptrons.py:98-137) and the multi-class variant
// (deprecated/MultiDiffCo.py:50-83): an inherently sequential loop — one argmin per iteration — whose body is
// tiny (N-element reductions, one lazily filled kernel row K(x_i, X), an axpy on the hypothesis).  The reference
// runs it as ~10 torch ops per iteration on the host (848 iterations = 0.38 s in its notebook).  Here the whole
// loop runs inside one launch: 1024 threads of one workgroup stride over the N samples, the state (y, gains,
// hypothesis, the lazily filled N x N kernel matrix) stays in HBM/L2, and nothing returns to the host until the
// loop ends.  With 288 GB of HBM the dense N x N matrix is affordable far beyond the reference's 10 000-sample
// "move it to the CPU to save VRAM" threshold (kernel_perceptrons.py:90-94, 151-155).
//
// Arithmetic mirrors the torch expressions (separate multiply and add where torch has two ops) so that the
// sequence of argmin choices, and therefore the support set, matches the reference's.
#include "dcx_internal.h"

// The reference's updates are separate torch ops (hypothesis += delta * K[i]: one rounding for the product, one for the
// sum), and the argmin sequence depends on those roundings.  HIP's __fmul_rn / __fadd_rn are plain operators compiled
// with the header's contraction state, so hipcc still fuses them into an fma (the generic kernel did, until round 2:
// its hypothesis differed from the register-resident kernels' in the last bit).  Contraction is therefore switched off
// for this translation unit and the updates use the helpers below; the kernel functions' own fmaf calls are explicit
// and stay.
#pragma clang fp contract(off)
namespace dcx {
namespace {
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
}  // namespace
}  // namespace dcx

namespace dcx {
namespace {

struct TrainArgs {
    const float* feats;  // [N, D] transformed samples
    const float* y;      // [N, C]  +-1 labels
    float* gains;        // [N, C]  in/out
    float* hypo;         // [N, C]  in/out
    float* K;            // [N, N]  in/out; a row is filled when first needed; K[i,i] == 0 means "not filled"
    int32_t* info;       // [2] out: iterations used, converged flag
    int32_t N, D, C, max_iter;
    int32_t kind;
    float kp0, kp1, beta;
};

struct Best {
    float v;
    int i;
};
// torch.min / torch.max semantics: a NaN wins over every number (a diverged run keeps walking like the reference's does,
// instead of leaving the search without a valid index), the first index wins among equals
__device__ __forceinline__ bool takes_min(float bv, int bi, float av, int ai) {
    const bool bn = bv != bv, an = av != av;
    return (bn && !an) || (bn == an && (bv < av || ((bv == av || bn) && bi < ai)));
}
__device__ __forceinline__ bool takes_max(float bv, int bi, float av, int ai) {
    const bool bn = bv != bv, an = av != av;
    return (bn && !an) || (bn == an && (bv > av || ((bv == av || bn) && bi < ai)));
}
__device__ __forceinline__ Best better_min(Best a, Best b) { return takes_min(b.v, b.i, a.v, a.i) ? b : a; }
__device__ __forceinline__ Best better_max(Best a, Best b) { return takes_max(b.v, b.i, a.v, a.i) ? b : a; }

// one step of the wave reduction through DPP (a lane that receives nothing keeps comparing with itself)
template <bool IS_MIN, int CTRL, int ROW_MASK>
__device__ __forceinline__ Best dpp_best(Best m) {
    const Best o{__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m.v), __builtin_bit_cast(int, m.v), CTRL, ROW_MASK, 0xF, false)),
                 __builtin_amdgcn_update_dpp(m.i, m.i, CTRL, ROW_MASK, 0xF, false)};
    return IS_MIN ? better_min(m, o) : better_max(m, o);
}
template <bool IS_MIN>
__device__ Best block_best(Best mine, Best* sB, int tid) {
    // four DPP steps inside each row of 16 lanes, two row broadcasts, v_readlane of lane 63: the choice (extreme value, lowest
    // index on ties) does not depend on the order of the comparisons.  (Six __shfl_xor steps of two words each went through
    // the LDS crossbar on the dependent chain of every iteration.)
    mine = dpp_best<IS_MIN, 0xB1, 0xF>(mine);    // quad_perm [1, 0, 3, 2]
    mine = dpp_best<IS_MIN, 0x4E, 0xF>(mine);    // quad_perm [2, 3, 0, 1]
    mine = dpp_best<IS_MIN, 0x141, 0xF>(mine);   // row_half_mirror
    mine = dpp_best<IS_MIN, 0x140, 0xF>(mine);   // row_mirror
    mine = dpp_best<IS_MIN, 0x142, 0xA>(mine);   // row_bcast15 into rows 1 and 3
    mine = dpp_best<IS_MIN, 0x143, 0xC>(mine);   // row_bcast31 into rows 2 and 3
    mine = Best{__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.v), 63)), __builtin_amdgcn_readlane(mine.i, 63)};
    __syncthreads();
    if ((tid & 63) == 0) sB[tid >> 6] = mine;
    __syncthreads();
    Best r = sB[0];
    const int nw = blockDim.x >> 6;
    for (int w = 1; w < nw; ++w) r = IS_MIN ? better_min(r, sB[w]) : better_max(r, sB[w]);
    return r;  // every thread computes the same result
}

__device__ float kernel_value(const TrainArgs& a, float d2) {
    ScoreArgs k{};
    k.kind = a.kind;
    k.kp0 = a.kp0;
    k.kp1 = a.kp1;
    float val, g;
    if (a.kind == DCX_K_RQ && a.kp1 == 2.0f) {
        kernel_eval<KF_RQ2>(d2, k, val, g);
    } else if (a.kind == DCX_K_POLY && a.kp0 == 1.0f) {
        val = (d2 > 0.f ? d2 * __builtin_amdgcn_rsqf(d2) : 0.f) / a.kp1;
    } else {
        kernel_eval<KF_GEN>(d2, k, val, g);
        if (a.kind == DCX_K_POLY && d2 == 0.f) val = 0.f;
    }
    return val;
}

// one perceptron step for label column c; returns true when the column has converged
__device__ bool class_step(const TrainArgs& a, int c, float* sX, Best* sB, int* sCnt) {
    const int tid = threadIdx.x, NT = blockDim.x, N = a.N, C = a.C;
    // 1. the worst margin (first index on ties, like torch.min)
    Best mine{INFINITY, 0x7fffffff};
    for (int j = tid; j < N; j += NT) {
        const float m = a.y[(size_t)j * C + c] * a.hypo[(size_t)j * C + c];
        mine = better_min(mine, Best{m, j});
    }
    const Best worst = block_best<true>(mine, sB, tid);
    const int i = worst.i;
    float* Ki = a.K + (size_t)i * N;
    // 2. fill row i of the kernel matrix on first use (k(x, x) != 0 marks a filled row)
    if (Ki[i] == 0.0f) {
        for (int k = tid; k < a.D; k += NT) sX[k] = a.feats[(size_t)i * a.D + k];
        __syncthreads();
        for (int j = tid; j < N; j += NT) {
            const float* xj = a.feats + (size_t)j * a.D;
            float d2 = 0.f;
            for (int k = 0; k < a.D; ++k) {
                const float dl = sX[k] - xj[k];
                d2 = fmaf(dl, dl, d2);
            }
            const float kv = kernel_value(a, d2);
            Ki[j] = kv;
            a.K[(size_t)j * N + i] = kv;  // and column i, like the reference's K[:, i] = K[i] (kernel_perceptrons.py:117-119)
        }
        __syncthreads();
    }
    const float kii = Ki[i];
    if (worst.v <= 0.0f) {
        // 3. margin violated: move sample i onto its target (beta scales the positive target)
        const float yi = a.y[(size_t)i * C + c], hi = a.hypo[(size_t)i * C + c];
        // beta^((1+y)/2) * y (kernel_perceptrons.py:121); exact shortcuts for the usual +-1 labels
        const float target = (yi == 1.0f ? a.beta : yi == -1.0f ? 1.0f : powf(a.beta, 0.5f * (1.0f + yi))) * yi;
        const float step = div_rn(sub_rn(target, hi), kii);
        __syncthreads();  // everyone has read hypo[i] before it changes
        for (int j = tid; j < N; j += NT) {
            const size_t o = (size_t)j * C + c;
            a.hypo[o] = add_rn(a.hypo[o], mul_rn(step, Ki[j]));
        }
        if (tid == 0) a.gains[(size_t)i * C + c] = add_rn(a.gains[(size_t)i * C + c], step);
        __syncthreads();
        return false;
    }
    // 4. all margins positive: retire a support that is classified correctly without its own contribution
    Best cand{-INFINITY, 0x7fffffff};
    int nnz = 0;
    for (int j = tid; j < N; j += NT) {
        const size_t o = (size_t)j * C + c;
        const float g = a.gains[o];
        float mm = 0.0f;
        if (g != 0.0f) {
            ++nnz;
            mm = mul_rn(a.y[o], sub_rn(a.hypo[o], mul_rn(g, a.K[(size_t)j * N + j])));
        }
        cand = better_max(cand, Best{mm, j});
    }
    const Best top = block_best<false>(cand, sB, tid);
    // count of active supports
    for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) sCnt[tid >> 6] = nnz;
    __syncthreads();
    int total = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += sCnt[w];
    if (top.v > 0.0f && total > 1) {
        const int jx = top.i;
        const float gj = a.gains[(size_t)jx * C + c];
        const float* Kj = a.K + (size_t)jx * N;
        __syncthreads();
        for (int j = tid; j < N; j += NT) {
            const size_t o = (size_t)j * C + c;
            a.hypo[o] = sub_rn(a.hypo[o], mul_rn(gj, Kj[j]));
        }
        if (tid == 0) a.gains[(size_t)jx * C + c] = 0.0f;
        __syncthreads();
        return false;
    }
    return true;
}

__global__ __launch_bounds__(1024) void perceptron_kernel(const TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                                         // [D]
    Best* sB = reinterpret_cast<Best*>(sX + ((a.D + 3) & ~3));  // [16]
    int* sCnt = reinterpret_cast<int*>(sB + 16);              // [16]
    int it = 0;
    bool converged = false;
    if (a.C == 1) {
        for (; it < a.max_iter; ++it) {
            if (class_step(a, 0, sX, sB, sCnt)) { converged = true; break; }
        }
    } else {
        unsigned done_mask = 0;  // a column that converged once stays flagged (deprecated/MultiDiffCo.py:52, 75-80)
        for (; it < a.max_iter; ++it) {
            for (int c = 0; c < a.C; ++c)
                if (class_step(a, c, sX, sB, sCnt)) done_mask |= 1u << c;
            if (done_mask == (1u << a.C) - 1u) { converged = true; break; }
        }
    }
    if (threadIdx.x == 0) {
        a.info[0] = it;
        a.info[1] = converged ? 1 : 0;
    }
}

// ---- register-resident variant (single class, N <= 1024 * EPT) ---------------------------------------------
// Each thread owns EPT samples (j = tid + e*NT) and keeps their margin, signed gain and kernel diagonal in
// registers for the whole training run; per iteration only the selected kernel row crosses memory (computed and
// stored on first use, re-loaded from the N x N matrix afterwards).  Everything else is register arithmetic plus
// two block reductions, so an iteration costs a few microseconds instead of several passes over global arrays.
// FL: the features sit in LDS for the whole run (row stride D | 1: consecutive samples in different banks) - the two L2 round
// trips of a first-use iteration (sample i's features, then every thread's own samples') become LDS reads
template <int EPT, int NT, bool FL = false>
__global__ __launch_bounds__(NT) void perceptron_reg_kernel(const TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                                         // [D]
    Best* sB = reinterpret_cast<Best*>(sX + ((a.D + 3) & ~3));  // [16]
    int* sCnt = reinterpret_cast<int*>(sB + 16);              // [16]
    float* sS = reinterpret_cast<float*>(sCnt + 16);          // [4] broadcast slots
    float* sFe = sS + 4;                                      // FL: [N][D | 1]
    const int tid = threadIdx.x, N = a.N;
    const int fstride = FL ? (a.D | 1) : a.D;
    const float* feats = FL ? sFe : a.feats;
    if constexpr (FL) {
        for (int e = tid; e < N * a.D; e += NT) {
            const int j = e / a.D, k = e - j * a.D;
            sFe[j * fstride + k] = a.feats[e];
        }
        __syncthreads();
    }
    {
        // The margin form below needs y in {-1, +1}.  Any other label (0/1 labels, y = 0) takes the generic loop,
        // which computes y*h and beta^((1+y)/2)*y with the actual y like the reference does.
        int bad = 0;
        for (int j = tid; j < N; j += NT) bad |= (a.y[j] != 1.0f && a.y[j] != -1.0f);
        if (__syncthreads_or(bad)) {
            int it = 0;
            bool converged = false;
            for (; it < a.max_iter; ++it)
                if (class_step(a, 0, sX, sB, sCnt)) { converged = true; break; }
            if (tid == 0) { a.info[0] = it; a.info[1] = converged ? 1 : 0; }
            return;
        }
    }
    // Per-sample state in "margin form": m = y*h and yg = y*g.  With y in {-1,+1} every update below is the
    // reference's update multiplied by an exact sign, so the roundings (and the argmin sequence) are identical,
    // and the label itself shrinks to one bit.
    float m[EPT], yg[EPT], dg[EPT];
    unsigned ypos = 0;  // bit e set <=> y_j > 0
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = tid + e * NT;
        const bool in = j < N;
        const float yj = in ? a.y[j] : 1.0f;
        if (yj > 0.f) ypos |= 1u << e;
        m[e] = in ? yj * a.hypo[j] : INFINITY;
        yg[e] = in ? yj * a.gains[j] : 0.0f;
        dg[e] = in ? a.K[(size_t)j * N + j] : 0.0f;
    }
    auto ysign = [&](int e) { return (ypos >> e) & 1u ? 1.0f : -1.0f; };
    int it = 0;
    bool converged = false;
    for (; it < a.max_iter; ++it) {
        // 1. worst margin (first index on ties)
        Best mine{INFINITY, 0x7fffffff};
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int j = tid + e * NT;
            if (j < N) mine = better_min(mine, Best{m[e], j});
        }
        const Best worst = block_best<true>(mine, sB, tid);
        const int i = worst.i;
        // owner broadcasts y_i and K_ii (h_i = y_i * m_i)
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (tid + e * NT == i) { sS[1] = ysign(e); sS[2] = dg[e]; }
        __syncthreads();
        const float yi = sS[1], hi = yi * worst.v;
        float kii = sS[2];
        float* Ki = a.K + (size_t)i * N;
        const bool violated = worst.v <= 0.0f;
        float krow[EPT <= 4 ? EPT : 1];   // EPT <= 4: the row's values this thread just computed (no reload behind the store)
        bool fresh = false;
        if (kii == 0.0f && EPT <= 4) {
            // 2. first use of row i, few samples per thread: computed once, stored in the N x N matrix (row and column), and
            //    kept in registers for the update below - the reload of a value just written is an L2 round trip per iteration
            for (int k = tid; k < a.D; k += NT) sX[k] = feats[(size_t)i * fstride + k];
            __syncthreads();
            fresh = true;
#pragma unroll
            for (int e = 0; e < (EPT <= 4 ? EPT : 1); ++e) {
                const int j = tid + e * NT;
                krow[e] = 0.0f;
                if (j < N) {
                    const float* xj = feats + (size_t)j * fstride;
                    float d2 = 0.f;
                    for (int k = 0; k < a.D; ++k) {
                        const float dl = sX[k] - xj[k];
                        d2 = fmaf(dl, dl, d2);
                    }
                    const float kv = kernel_value(a, d2);
                    krow[e] = kv;
                    Ki[j] = kv;
                    a.K[(size_t)j * N + i] = kv;
                    if (j == i) sS[3] = kv;
                }
            }
            __syncthreads();
            kii = sS[3];
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (tid + e * NT == i) dg[e] = kii;
        } else if (kii == 0.0f) {
            // 2. first use of row i: compute it once (rolled loop: the kernel function may be a powf/logf body)
            //    and store it in the N x N matrix
            for (int k = tid; k < a.D; k += NT) sX[k] = a.feats[(size_t)i * a.D + k];
            __syncthreads();
#pragma unroll 1
            for (int j = tid; j < N; j += NT) {
                const float* xj = a.feats + (size_t)j * a.D;
                float d2 = 0.f;
                for (int k = 0; k < a.D; ++k) {
                    const float dl = sX[k] - xj[k];
                    d2 = fmaf(dl, dl, d2);
                }
                const float kv = kernel_value(a, d2);
                Ki[j] = kv;
                a.K[(size_t)j * N + i] = kv;  // column i as well (the reference fills K[i, :] and K[:, i] together)
                if (j == i) sS[3] = kv;
            }
            __syncthreads();
            kii = sS[3];
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (tid + e * NT == i) dg[e] = kii;
        }
        if (violated) {
            // 3. margin violated: move sample i onto its target:  h += step * K_i,  g_i += step
            const float target = (yi > 0.f ? a.beta : 1.0f) * yi;
            const float step = div_rn(sub_rn(target, hi), kii);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int j = tid + e * NT;
                if (j < N) m[e] = add_rn(m[e], ysign(e) * mul_rn(step, (EPT <= 4 && fresh) ? krow[e < (EPT <= 4 ? EPT : 1) ? e : 0] : Ki[j]));
                if (j == i) yg[e] = add_rn(yg[e], ysign(e) * step);
            }
            __syncthreads();  // sS is rewritten next iteration
            continue;
        }
        // 4. all margins positive: retire a support that is classified correctly without its own contribution
        Best cand{-INFINITY, 0x7fffffff};
        int nnz = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int j = tid + e * NT;
            if (j < N) {
                float mm = 0.0f;
                if (yg[e] != 0.0f) {
                    ++nnz;
                    mm = sub_rn(m[e], mul_rn(yg[e], dg[e]));
                }
                cand = better_max(cand, Best{mm, j});
            }
        }
        const Best top = block_best<false>(cand, sB, tid);
        for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) sCnt[tid >> 6] = nnz;
        __syncthreads();
        int total = 0;
        for (int w = 0; w < NT / 64; ++w) total += sCnt[w];
        if (top.v > 0.0f && total > 1) {
            const int jx = top.i;
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (tid + e * NT == jx) sS[0] = ysign(e) * yg[e];   // g_jx
            __syncthreads();
            const float gj = sS[0];
            const float* Kj = a.K + (size_t)jx * N;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int j = tid + e * NT;
                if (j < N) m[e] = sub_rn(m[e], ysign(e) * mul_rn(gj, Kj[j]));
                if (j == jx) yg[e] = 0.0f;
            }
            __syncthreads();
            continue;
        }
        converged = true;
        break;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = tid + e * NT;
        if (j < N) { a.hypo[j] = ysign(e) * m[e]; a.gains[j] = ysign(e) * yg[e]; }
    }
    if (tid == 0) {
        a.info[0] = it;
        a.info[1] = converged ? 1 : 0;
    }
}

// ---- several workgroups (single class, -1 / +1 labels, N beyond one workgroup's registers) -----------------------
// The register-resident loop spread over G workgroups on G CUs.  Workgroup g owns samples j = g * NT + tid + e * G * NT
// and keeps their margin, signed gain and kernel diagonal in registers; per iteration each workgroup reduces its own
// candidates, posts one 16-byte record, and a grid-wide barrier (arrival counter in device memory, agent-scope release /
// acquire) turns the G records into the same global choice everywhere — the argmin with the lowest index on ties, i.e.
// exactly the sequence of the one-workgroup kernel.  The kernel matrix needs no cross-workgroup visibility: when
// sample i is first selected every workgroup fills ITS part of row i (and of column i), and later only reads entries it
// wrote itself.  One barrier per violated-margin iteration, two per retire iteration.
// The launch is cooperative (all G workgroups resident), and a barrier that does not complete within ~2 s sets an abort
// flag that every workgroup honours (info[1] = -1) instead of hanging the device.
struct GridRec {
    float v;
    int i;
    float a0, a1;
};
struct GridSync {
    unsigned int counter;   // arrivals, monotonically increasing
    unsigned int abort;
    unsigned int bad_labels;  // some label is neither -1 nor +1: workgroup 0 runs the generic loop, the others leave
    unsigned int pad[1];
    GridRec rec[2][128];    // double-buffered by barrier parity
};
constexpr int kGridMaxWg = 128;
constexpr int kGridNT = 256, kGridEPT = 4;  // 1024 samples per workgroup

// every thread calls; returns false when the run was aborted
__device__ __forceinline__ bool grid_barrier(GridSync* gs, unsigned& n_done, int tid) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    ++n_done;
    if (tid == 0) {
        const unsigned target = n_done * gridDim.x;
        __hip_atomic_fetch_add(&gs->counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        while (__hip_atomic_load(&gs->counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (__hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (wall_clock64() - t0 > 200000000ull) {
                __hip_atomic_store(&gs->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return __hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

__device__ __forceinline__ void post_rec(GridRec* slot, float v, int i, float a0, float a1) {
    __hip_atomic_store(&slot->v, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->i, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->a0, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->a1, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0 folds the G records (best by v with the lowest index on ties; a1 summed when SUM_A1) -> sR[0]; caller syncs
template <bool IS_MIN, bool SUM_A1>
__device__ __forceinline__ void fold_recs(const GridRec* recs, int G, GridRec* sR, int tid) {
    if (tid < 64) {
        GridRec best{IS_MIN ? INFINITY : -INFINITY, 0x7fffffff, 0.f, 0.f};
        float sum = 0.f;
        for (int g = tid; g < G; g += 64) {
            GridRec r;
            r.v = __hip_atomic_load(&recs[g].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.i = __hip_atomic_load(&recs[g].i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.a0 = __hip_atomic_load(&recs[g].a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.a1 = __hip_atomic_load(&recs[g].a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sum += r.a1;
            const bool take = IS_MIN ? takes_min(r.v, r.i, best.v, best.i) : takes_max(r.v, r.i, best.v, best.i);
            if (take) best = r;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            GridRec r{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64), __shfl_xor(best.a0, o, 64), __shfl_xor(best.a1, o, 64)};
            sum += __shfl_xor(sum, o, 64);
            const bool take = IS_MIN ? takes_min(r.v, r.i, best.v, best.i) : takes_max(r.v, r.i, best.v, best.i);
            if (take) best = r;
        }
        if (SUM_A1) best.a1 = sum;
        if (tid == 0) sR[0] = best;
    }
}

template <int EPT, int NT>
__global__ __launch_bounds__(NT) void perceptron_grid_kernel(const TrainArgs a, GridSync* gs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                                           // [D]
    Best* sB = reinterpret_cast<Best*>(sX + ((a.D + 3) & ~3));  // [16]
    int* sCnt = reinterpret_cast<int*>(sB + 16);                // [16]
    GridRec* sR = reinterpret_cast<GridRec*>(sCnt + 16);        // [1]
    const int tid = threadIdx.x, N = a.N, G = gridDim.x;
    const int base = blockIdx.x * NT + tid, stride = G * NT;
    float m[EPT], yg[EPT], dg[EPT];
    unsigned ypos = 0;
    unsigned n_bar = 0;
    {
        // The margin form below needs y in {-1, +1}; the check happens HERE, on the device (round 3: the host used to read
        // the labels back and synchronise the caller's stream for it - the one entry point of the library that waited).  Any
        // other label (0 / 1 labels, y = 0): one grid-wide agreement, then workgroup 0 runs the generic loop, which
        // evaluates the reference's expressions on y itself, and the other workgroups leave.
        int bad = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int j = base + e * stride;
            if (j < N) bad |= (a.y[j] != 1.0f && a.y[j] != -1.0f);
        }
        if (__syncthreads_or(bad) && tid == 0) __hip_atomic_store(&gs->bad_labels, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool alive = grid_barrier(gs, n_bar, tid);
        if (!alive || __hip_atomic_load(&gs->bad_labels, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (blockIdx.x != 0) return;
            int it = 0;
            bool converged = false;
            if (alive)
                for (; it < a.max_iter; ++it)
                    if (class_step(a, 0, sX, sB, sCnt)) { converged = true; break; }
            if (tid == 0) { a.info[0] = it; a.info[1] = alive ? (converged ? 1 : 0) : -1; }
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = base + e * stride;
        const bool in = j < N;
        const float yj = in ? a.y[j] : 1.0f;
        if (yj > 0.f) ypos |= 1u << e;
        m[e] = in ? yj * a.hypo[j] : INFINITY;
        yg[e] = in ? yj * a.gains[j] : 0.0f;
        dg[e] = in ? a.K[(size_t)j * N + j] : 0.0f;
    }
    auto ysign = [&](int e) { return (ypos >> e) & 1u ? 1.0f : -1.0f; };
    int it = 0;
    int converged = 0;
    for (; it < a.max_iter; ++it) {
        // 1. worst margin: this workgroup's, then everyone's
        Best mine{INFINITY, 0x7fffffff};
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int j = base + e * stride;
            if (j < N) mine = better_min(mine, Best{m[e], j});
        }
        const Best wg = block_best<true>(mine, sB, tid);
        GridRec* recs = gs->rec[n_bar & 1];
        if (wg.i == 0x7fffffff) {
            if (tid == 0) post_rec(&recs[blockIdx.x], INFINITY, 0x7fffffff, 0.f, 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (base + e * stride == wg.i) post_rec(&recs[blockIdx.x], wg.v, wg.i, ysign(e), dg[e]);  // the owner posts y_i, K_ii
        }
        if (!grid_barrier(gs, n_bar, tid)) { converged = -1; break; }
        fold_recs<true, false>(recs, G, sR, tid);
        __syncthreads();
        const GridRec worst = sR[0];
        const int i = worst.i;
        const float yi = worst.a0, hi = yi * worst.v;
        float kii = worst.a1;
        float* Ki = a.K + (size_t)i * N;
        const bool violated = worst.v <= 0.0f;
        float krow[EPT];
        bool fresh = false;
        if (kii == 0.0f) {
            // 2. first use of row i: every workgroup fills its own part of the row and of the column
            for (int k = tid; k < a.D; k += NT) sX[k] = a.feats[(size_t)i * a.D + k];
            __syncthreads();
            fresh = true;   // (the values stay in registers for the update below: no reload behind the store, see perceptron_reg_kernel)
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int j = base + e * stride;
                krow[e] = 0.0f;
                if (j < N) {
                    const float* xj = a.feats + (size_t)j * a.D;
                    float d2 = 0.f;
                    for (int k = 0; k < a.D; ++k) {
                        const float dl = sX[k] - xj[k];
                        d2 = fmaf(dl, dl, d2);
                    }
                    const float kv = kernel_value(a, d2);
                    krow[e] = kv;
                    Ki[j] = kv;
                    a.K[(size_t)j * N + i] = kv;
                }
            }
            kii = kernel_value(a, 0.0f);  // d2 of sample i against itself is an exact zero
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (base + e * stride == i) dg[e] = kii;
        }
        if (violated) {
            // 3. margin violated: h += step * K_i, g_i += step
            const float target = (yi > 0.f ? a.beta : 1.0f) * yi;
            const float step = div_rn(sub_rn(target, hi), kii);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int j = base + e * stride;
                if (j < N) m[e] = add_rn(m[e], ysign(e) * mul_rn(step, fresh ? krow[e] : Ki[j]));
                if (j == i) yg[e] = add_rn(yg[e], ysign(e) * step);
            }
            __syncthreads();  // sR / sX are rewritten next iteration
            continue;
        }
        // 4. all margins positive: retire a support that is classified correctly without its own contribution
        Best cand{-INFINITY, 0x7fffffff};
        int nnz = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int j = base + e * stride;
            if (j < N) {
                float mm = 0.0f;
                if (yg[e] != 0.0f) {
                    ++nnz;
                    mm = sub_rn(m[e], mul_rn(yg[e], dg[e]));
                }
                cand = better_max(cand, Best{mm, j});
            }
        }
        const Best top = block_best<false>(cand, sB, tid);
        for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) sCnt[tid >> 6] = nnz;
        __syncthreads();
        int wg_nnz = 0;
        for (int w = 0; w < NT / 64; ++w) wg_nnz += sCnt[w];
        recs = gs->rec[n_bar & 1];
        if (top.i == 0x7fffffff) {
            if (tid == 0) post_rec(&recs[blockIdx.x], -INFINITY, 0x7fffffff, 0.f, 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (base + e * stride == top.i) post_rec(&recs[blockIdx.x], top.v, top.i, ysign(e) * yg[e], (float)wg_nnz);  // g_jx, count
        }
        if (!grid_barrier(gs, n_bar, tid)) { converged = -1; break; }
        fold_recs<false, true>(recs, G, sR, tid);
        __syncthreads();
        const GridRec best = sR[0];
        if (best.v > 0.0f && best.a1 > 1.0f) {
            const int jx = best.i;
            const float gj = best.a0;
            const float* Kj = a.K + (size_t)jx * N;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int j = base + e * stride;
                if (j < N) m[e] = sub_rn(m[e], ysign(e) * mul_rn(gj, Kj[j]));
                if (j == jx) yg[e] = 0.0f;
            }
            __syncthreads();
            continue;
        }
        converged = 1;
        break;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = base + e * stride;
        if (j < N) { a.hypo[j] = ysign(e) * m[e]; a.gains[j] = ysign(e) * yg[e]; }
    }
    if (tid == 0 && blockIdx.x == 0) {
        a.info[0] = it;
        a.info[1] = converged;
    }
}

}  // namespace

int perceptron_grid_workgroups(int N) { return (N + kGridNT * kGridEPT - 1) / (kGridNT * kGridEPT); }

hipError_t launch_perceptron(int kind, float kp0, float kp1, float beta, const float* feats, const float* y, float* gains,
                             float* hypo, float* K, int32_t* info, int N, int D, int C, int max_iter, bool sign_labels,
                             bool grid, hipStream_t st) {
    TrainArgs a;
    a.feats = feats; a.y = y; a.gains = gains; a.hypo = hypo; a.K = K; a.info = info;
    a.N = N; a.D = D; a.C = C; a.max_iter = max_iter; a.kind = kind; a.kp0 = kp0; a.kp1 = kp1; a.beta = beta;
    const size_t lds = sizeof(float) * ((D + 3) & ~3) + 16 * sizeof(Best) + 16 * sizeof(int) + 4 * sizeof(float) + sizeof(GridRec);
    // (a stream that is being captured takes the one-workgroup kernels: stream-ordered allocation and a cooperative launch
    // cannot be recorded, and a refused one could invalidate the capture before the fallback below runs)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusActive;
    }
    if (grid && cap == hipStreamCaptureStatusNone && sign_labels && C == 1 && perceptron_grid_workgroups(N) <= kGridMaxWg) {
        // several workgroups: a stream-ordered scratch for the records and the arrival counter, a cooperative launch
        // (if the cooperative launch is refused - partition mode, resources - the one-workgroup kernels below take over)
        GridSync* gs = nullptr;
        hipError_t e = hipMallocAsync((void**)&gs, sizeof(GridSync), st);
        if (e == hipSuccess) {
            e = hipMemsetAsync(gs, 0, sizeof(GridSync), st);
            void* params[] = {(void*)&a, (void*)&gs};
            if (e == hipSuccess)
                e = hipLaunchCooperativeKernel((const void*)perceptron_grid_kernel<kGridEPT, kGridNT>,
                                               dim3(perceptron_grid_workgroups(N)), dim3(kGridNT), params, (unsigned)lds, st);
            const hipError_t f = hipFreeAsync(gs, st);
            if (e == hipSuccess) return f;
        }
        (void)hipGetLastError();
    }
    if (sign_labels && C == 1 && N <= 768) {
        // an active-learning round trains on a few hundred samples: four waves reduce and synchronise faster than sixteen that
        // mostly hold nothing (N = 300: 0.37 -> 0.32 ms for 58 iterations, N = 640: 0.84 -> 0.78; level at 1000).  Same sequence.
        const size_t lds_fl = lds + sizeof(float) * (size_t)N * (D | 1);
        if (lds_fl <= 150 * 1024) {   // the features ride in LDS (N = 640, D = 24: 64 KB)
            auto kern = perceptron_reg_kernel<4, 256, true>;
            if (lds_fl > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fl) != hipSuccess) {
                (void)hipGetLastError();
                perceptron_reg_kernel<4, 256><<<dim3(1), dim3(256), lds, st>>>(a);
            } else {
                kern<<<dim3(1), dim3(256), lds_fl, st>>>(a);
            }
        } else {
            perceptron_reg_kernel<4, 256><<<dim3(1), dim3(256), lds, st>>>(a);
        }
    } else if (sign_labels && C == 1 && N <= 1024 * 4) {
        const size_t lds_fl = lds + sizeof(float) * (size_t)N * (D | 1);
        auto kern = perceptron_reg_kernel<4, 1024, true>;
        if (lds_fl <= 150 * 1024 && (lds_fl <= 64 * 1024 || hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fl) == hipSuccess)) {
            kern<<<dim3(1), dim3(1024), lds_fl, st>>>(a);   // (the features in LDS while they fit: N = 1500 at D = 24)
        } else {
            (void)hipGetLastError();
            perceptron_reg_kernel<4, 1024><<<dim3(1), dim3(1024), lds, st>>>(a);
        }
    } else if (sign_labels && C == 1 && N <= 512 * 20) {   // 8 waves = 2 per SIMD: 256 VGPRs per lane hold 20 samples' state
        perceptron_reg_kernel<20, 512><<<dim3(1), dim3(512), lds, st>>>(a);
    } else {
        perceptron_kernel<<<dim3(1), dim3(1024), lds, st>>>(a);
    }
    return hipGetLastError();
}

}  // namespace dcx
