// solve_body.h — the body of solve_kernels.hip, compiled once per workgroup size (DCX_SOLVE_NT = 256 and 512, each inside its
// own namespace): everything below is written against kNT / kTJ.  No include guard on purpose.

constexpr int kNT = DCX_SOLVE_NT;   // threads per workgroup
constexpr int kTJ = kNT / 32;       // trailing columns per group: one lane per (column, row of the block)
static_assert(kNT == 256 || kNT == 512, "the register budget below is written for 4 or 8 waves");
constexpr int kMaxNb = 32;

// columns per panel: 64 doubles per thread hold ceil(m / kNT) rows of nb columns
__host__ __device__ inline int nbt_for(int m) { return m <= 2 * kNT ? 32 : m <= 4 * kNT ? 16 : m <= 8 * kNT ? 8 : 4; }
__host__ __device__ inline int nb_for(int m) { return nbt_for(m) < m ? nbt_for(m) : m; }


// every thread calls; false when the run was aborted
__device__ __forceinline__ bool grid_barrier(SolveSync* gs, unsigned& n_done, int tid) {
    if (gridDim.x == 1) {
        __syncthreads();
        return true;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    ++n_done;
    if (tid == 0) {
        const unsigned target = n_done * gridDim.x;
        __hip_atomic_fetch_add(&gs->counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        while (__hip_atomic_load(&gs->counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
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

// ---- the panel: rows k0..n, columns k0..k0+nb, factorised in REGISTERS by one workgroup ----------------------------------
// Thread t keeps the rows that START as panel rows t, t + 256, ... (R = 64 / NB of them) as NB doubles each: 64 doubles per
// thread whatever the shape.  A row never leaves its thread: LAPACK's "rows c and p change places" is a relabelling - each
// slot carries the panel position `pos` its content currently stands at, and a swap exchanges two labels.  Per column:
//   * the pivot search is a wave reduction and ONE LDS atomic per wave, ds_max_u64, of a single 64-bit key = |value|'s bit pattern with the low 12 mantissa bits
//     replaced by 4095 - position (non-negative doubles order like their bit patterns; among candidates equal to 1e-12 the
//     lowest position wins, as idamax's first maximum does), one word per panel column so nothing is ever reset;
//   * the owner of the pivot row publishes it through LDS; every thread applies the rank-1 update to its rows below c with
//     compile-time register indices.  The multipliers are value x (1 / pivot), the reciprocal from v_rcp_f64 and two
//     Newton steps (LAPACK's getf2 scales by the reciprocal too).
// Two __syncthreads and ~100 instructions per column; no LDS traffic in the update.  The columns are a fold expression, not
// a loop (a loop would index the register array dynamically), and the row selection is per-lane data flow (`if (pos[r] ==
// p)`, never `if (r == rp)`: the optimiser turns such chains into v[rp] and the array into scratch).
// (Round 4, first forms: the panel in LDS, a ds_read / fma / ds_write chain per element - 4.6 us per column at n = 438;
//  registers with a three-word shuffle reduction and IEEE division - 1.36 us.)
// Leaves L (unit lower, multipliers) and U in W in LAPACK's row order, and the pivots' net row movement in gs.
template <int NB, int R>
struct PanelRegs {
    double v[R][NB];   // this thread's rows
    int pos[R];        // the panel position each of them stands at (m or more: not a row)
};

// the largest 32-bit value of a wave, in every lane: four DPP steps inside each row of 16 lanes, two row broadcasts, one
// v_readlane (~10 cycles a step; a __shfl_xor tree goes through the LDS crossbar, ~100 cycles a step, and sits on the
// dependent chain of every panel column)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned x) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xF, false);
    return o > x ? o : x;
}
__device__ __forceinline__ unsigned wave_umax(unsigned x) {
    x = dpp_umax<0xB1, 0xF>(x);    // quad_perm [1, 0, 3, 2]
    x = dpp_umax<0x4E, 0xF>(x);    // quad_perm [2, 3, 0, 1]
    x = dpp_umax<0x141, 0xF>(x);   // row_half_mirror
    x = dpp_umax<0x140, 0xF>(x);   // row_mirror
    x = dpp_umax<0x142, 0xA>(x);   // row_bcast15 into rows 1 and 3
    x = dpp_umax<0x143, 0xC>(x);   // row_bcast31 into rows 2 and 3: lane 63 has seen all 64
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ unsigned long long wave_umax64(unsigned long long k) {
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned mh = wave_umax(hi);
    const unsigned ml = wave_umax(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}

__device__ __forceinline__ double read_lane(double x, int lane) {   // lane uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}

__device__ __forceinline__ double fast_reciprocal(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// column C of the panel (C a compile-time constant: every register index below is one)
// (Tried: ONE barrier per column - every wave publishes its own best candidate's key AND row before the barrier, everybody
//  picks the winner behind it.  Slower, 0.81 -> 0.96 ms at n = 438: now every wave issues the 32 - C row writes in front of the
//  barrier, where only the one owner did behind the first.)
template <int NB, int C>
__device__ __forceinline__ void panel_column(SolveSync* gs, PanelRegs<NB, 64 / NB>& g, int k0, int m, double* sRowP,
                                             unsigned long long* sKey, int tid) {
    constexpr int R = 64 / NB;
    const unsigned row_lds = (unsigned)(size_t)sRowP;   // the LDS byte address (the low half of the flat one)
    DCX_PTS_DECL;
    unsigned long long mine = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long key =
            (static_cast<unsigned long long>(__double_as_longlong(fabs(g.v[r][C]))) & ~0xFFFull) | (unsigned)(4095 - g.pos[r]);
        if (g.pos[r] >= C && g.pos[r] < m) mine = key > mine ? key : mine;
    }
    mine = wave_umax64(mine);
    if ((tid & 63) == 0 && mine != 0) atomicMax(&sKey[C], mine);   // (256 same-address atomics would serialise: 2.5 us)
    __syncthreads();
    DCX_PTS(0);
    const unsigned long long best = sKey[C];
    const int p = best != 0 ? 4095 - (int)(best & 0xFFF) : C;
    // (an all-zero column: the key is the position alone; NaN: larger than every number, and not > 0 either)
    if (tid == 0 && !(__longlong_as_double((long long)(best & ~0xFFFull)) > 0.0) && gs->info == 0) gs->info = k0 + C + 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (g.pos[r] == p) {   // (per lane: one slot of one thread)
            // one ds_write_b64 per value, straight from its register pair (the compiler pairs them into ds_write2_b64 and
            // first copies every operand into an aligned quad: four v_mov per write on the wave everybody waits for)
#pragma unroll
            for (int cc = C; cc < NB; ++cc)
                asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(row_lds), "v"(g.v[r][cc]), "n"(cc * 8) : "memory");
        }
        g.pos[r] = g.pos[r] == p ? C : (g.pos[r] == C ? p : g.pos[r]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");   // (the writes above are invisible to the compiler's counters)
    __syncthreads();
    DCX_PTS(1);
    double up[NB];
#pragma unroll
    for (int cc = C; cc < NB; ++cc) up[cc] = sRowP[cc];
    const double rinv = up[C] != 0.0 ? fast_reciprocal(up[C]) : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (g.pos[r] > C && g.pos[r] < m) {
            const double l = g.v[r][C] * rinv;
            g.v[r][C] = l;
#pragma unroll
            for (int cc = C + 1; cc < NB; ++cc) g.v[r][cc] -= l * up[cc];
        }
    }
    DCX_PTS(2);
    // (the next pivot row is written to LDS behind the next pivot search's barrier: every read above is done by then)
}

template <int NB, int... Cs>
__device__ __forceinline__ void panel_columns(SolveSync* gs, PanelRegs<NB, 64 / NB>& g, int k0, int nb, int m, double* sRowP,
                                              unsigned long long* sKey, int tid, std::integer_sequence<int, Cs...>) {
    // (nb < NB only in the last panel of a matrix; the test is uniform)
    ((Cs < nb ? panel_column<NB, Cs>(gs, g, k0, m, sRowP, sKey, tid) : (void)0), ...);
}

template <int NB>
__device__ __forceinline__ void factor_panel(SolveSync* gs, int buf, double* W, size_t ld, int n, int k0, int nb, double* sRowP,
                                             unsigned long long* sKey, int* sCnt, int tid) {
    constexpr int R = 64 / NB;
    const int m = n - k0;
    PanelRegs<NB, R> g;
    DCX_PTS_DECL;
    // every load unconditional, from a clamped address, all 64 issued before the first is looked at - and opaque: left to
    // itself the compiler sinks each one under `i < m` again, 64 branches, each waiting for its own load (15 us per panel
    // at n = 438, 51 us at n = 2000)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + kNT * r;
        g.pos[r] = i;
#pragma unroll
        for (int cc = 0; cc < NB; ++cc) g.v[r][cc] = W[(size_t)(k0 + (cc < nb ? cc : 0)) * ld + k0 + (i < m ? i : 0)];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + kNT * r;
#pragma unroll
        for (int cc = 0; cc < NB; ++cc) {
            asm volatile("" : "+v"(g.v[r][cc]));
            g.v[r][cc] = (i < m && cc < nb) ? g.v[r][cc] : 0.0;
        }
    }
    if (tid == 0) *sCnt = 0;
    if (tid < 32) sKey[tid] = 0;
    __syncthreads();
    DCX_PTS(3);
    panel_columns<NB>(gs, g, k0, nb, m, sRowP, sKey, tid, std::make_integer_sequence<int, NB>{});
    DCX_PTS(6);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + kNT * r, at = g.pos[r];
        if (i < m) {
#pragma unroll
            for (int cc = 0; cc < NB; ++cc)
                if (cc < nb) W[(size_t)(k0 + cc) * ld + k0 + at] = g.v[r][cc];
            // the swaps' net effect: the source row of each top row, and the rows below that received another row's content
            if (at < nb) gs->top_src[buf][at] = k0 + i;
            else if (at != i) {
                const int slot = atomicAdd(sCnt, 1);
                gs->low_dst[buf][slot] = k0 + at;
                gs->low_src[buf][slot] = k0 + i;
            }
        }
    }
    DCX_PTS(4);
    __syncthreads();
    if (tid == 0) gs->n_low[buf] = *sCnt;
    DCX_PTS(5);
}

// rows i >= k0 + nb of one column group: W[i][j..] -= L[i][0..nb) . U[0..nb)[j..]
// RB rows per thread and pass (32 / NB, at most 4): all their loads - NB of L and 8 of the group per row - are issued before
// the first product, so a narrow panel (NB = 8 or 4, the large matrices) has as many bytes in flight as a wide one.
template <int NB>
__device__ __forceinline__ void update_rows(const SolveArgs& a, int k0, int nb, int j0, int nj, const double* sU, int tid) {
    constexpr int RB0 = NB >= 32 ? 1 : NB == 16 ? 2 : 4, RB = RB0 * kTJ > 32 ? 32 / kTJ : RB0;
    constexpr int CH = 32 / kTJ;   // rows of U per chunk: 32 doubles
    const size_t ld = (size_t)a.ld;
    const double* Lp = a.W + (size_t)k0 * ld;
    double* Cj = a.W + (size_t)j0 * ld;
#pragma unroll 1
    for (int i0 = k0 + nb + tid; i0 < a.n; i0 += kNT * RB) {
        asm volatile("" ::: "memory");   // U stays in LDS: hoisted out of this loop it would be 2 * NB * 8 registers
        double l[RB][NB], acc[RB][kTJ];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int i = i0 + kNT * rb < a.n ? i0 + kNT * rb : i0;   // (a row past the end repeats the first; never stored)
#pragma unroll
            for (int c = 0; c < NB; ++c) l[rb][c] = Lp[(size_t)(c < nb ? c : 0) * ld + i];   // (columns >= nb meet zero rows of U)
#pragma unroll
            for (int jj = 0; jj < kTJ; ++jj) acc[rb][jj] = Cj[(size_t)(jj < nj ? jj : 0) * ld + i];
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (c % CH == 0) {
                // a few rows of U at a time: the loads may not rise above this point and the sums may not sink below it
                // (left alone the scheduler reads all NB x 8 values first: 512 registers for NB = 32)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                    for (int o = 0; o < kTJ; o += 8) DCX_PIN8(acc[rb], o);
                }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                for (int jj = 0; jj < kTJ; ++jj) acc[rb][jj] -= l[rb][c] * sU[c * kTJ + jj];
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int i = i0 + kNT * rb;
            if (i < a.n) {
#pragma unroll
                for (int jj = 0; jj < kTJ; ++jj)
                    if (jj < nj) Cj[(size_t)jj * ld + i] = acc[rb][jj];
            }
        }
    }
}

// one group of <= 8 trailing columns: row movement, forward substitution with L11, rank-nb update
__device__ __forceinline__ void trailing_group(const SolveArgs& a, int k0, int nb, int j0, int nj, const double* sL, double* sU, const int* sTop,
                               const int* sLowDst, const int* sLowSrc, int n_low, int tid) {
    const size_t ld = (size_t)a.ld;
    const int jj = tid >> 5, d = tid & 31;   // (column of the group, row of the block)
    const bool on = jj < nj && d < nb;
    double* colp = a.W + (size_t)(j0 + (jj < nj ? jj : 0)) * ld;
    double top = on ? colp[sTop[d]] : 0.0;
    // the displaced rows below the block: item e = (column, pair)
    double lowv = 0.0;
    const int e_col = tid / (n_low > 0 ? n_low : 1), e_pair = tid - e_col * (n_low > 0 ? n_low : 1);
    const bool low_on = n_low > 0 && e_col < nj;
    if (low_on) lowv = a.W[(size_t)(j0 + e_col) * ld + sLowSrc[e_pair]];
    __syncthreads();
    if (low_on) a.W[(size_t)(j0 + e_col) * ld + sLowDst[e_pair]] = lowv;
    // forward substitution: lane (jj, d) owns u[d] of column jj; two columns per wave
    const int half = (tid & 63) & 32;
    for (int c = 0; c < nb; ++c) {
        const double u0 = read_lane(top, c), u1 = read_lane(top, 32 + c);   // (v_readlane: a shuffle would go through LDS)
        const double uc = half ? u1 : u0;
        if (d > c) top -= sL[c * 33 + d] * uc;
    }
    sU[d * kTJ + jj] = on ? top : 0.0;   // zero rows / columns beyond nb / nj: update_rows multiplies them in
    if (on) colp[k0 + d] = top;
    __syncthreads();
    switch (nb > 16 ? 32 : nb > 8 ? 16 : nb > 4 ? 8 : 4) {
        case 32: update_rows<32>(a, k0, nb, j0, nj, sU, tid); break;
        case 16: update_rows<16>(a, k0, nb, j0, nj, sU, tid); break;
        case 8: update_rows<8>(a, k0, nb, j0, nj, sU, tid); break;
        default: update_rows<4>(a, k0, nb, j0, nj, sU, tid); break;
    }
    __syncthreads();
}

// the 32 dependent steps of a diagonal block, unknown C = 31 - K solved at step K (a fold: u[] keeps static indices)
template <int C>
__device__ __forceinline__ void back_step(double& v, const double (&u)[32], double rdiag, int nbk, int d, int half) {
    if (C < nbk) {   // uniform
        const double mine = v * rdiag;   // (the solved value, where d == C)
        const double x0 = read_lane(mine, C), x1 = read_lane(mine, 32 + C);
        const double xc = half ? x1 : x0;
        v = d == C ? xc : (d < C ? v - u[C] * xc : v);
    }
}
template <int... Ks>
__device__ __forceinline__ void back_chain(double& v, const double (&u)[32], double rdiag, int nbk, int d, int half,
                                           std::integer_sequence<int, Ks...>) {
    (back_step<31 - Ks>(v, u, rdiag, nbk, d, half), ...);
}

// U x = y for every right-hand side, blockwise from the bottom; one workgroup.  sD [32][33]: the diagonal block of U.
// Per block of 32 unknowns: lane (jj, d) owns x[d] of right-hand side jj, keeps ITS column of the block (U[d][c], all c) and
// 1 / U[d][d] in registers, and the 32 dependent steps exchange the solved value by v_readlane (a shuffle through LDS
// would put ~100 cycles on each step); then every thread takes rows above the block and subtracts U[i][block] x with the
// 32 loads of a row in flight together.  (First form: conditional loads one at a time, a division and two LDS reads on
// the chain: 9.3 us per block.)
__device__ __forceinline__ void back_substitute(const SolveArgs& a, double* sD, double* sX, int tid) {
    const int n = a.n;
    const size_t ld = (size_t)a.ld;
    const int jj = tid >> 5, d = tid & 31, half = (tid & 63) & 32;
    for (int r0 = 0; r0 < a.nrhs; r0 += kTJ) {
        const int nr = a.nrhs - r0 < kTJ ? a.nrhs - r0 : kTJ;
        double* y = a.W + (size_t)(n + r0 + (jj < nr ? jj : 0)) * ld;
        for (int kb = (n - 1) / 32 * 32; kb >= 0; kb -= 32) {
            const int nbk = n - kb < 32 ? n - kb : 32;
            constexpr int DQ = 1024 / kNT;
            double dv[DQ];
#pragma unroll
            for (int q = 0; q < DQ; ++q) {   // (unconditional loads from clamped addresses, all in flight)
                const int e = tid + kNT * q, c = e >> 5, r = e & 31;
                dv[q] = a.W[(size_t)(kb + (c < nbk ? c : 0)) * ld + kb + (r < nbk ? r : 0)];
            }
            const bool on = jj < nr && d < nbk;
            double v = y[kb + (d < nbk ? d : 0)];
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                const int e = tid + kNT * q, c = e >> 5, r = e & 31;
                asm volatile("" : "+v"(dv[q]));
                sD[c * 33 + r] = (c < nbk && r <= c) ? dv[q] : 0.0;
            }
            asm volatile("" : "+v"(v));
            v = on ? v : 0.0;
            __syncthreads();
            double u[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) u[c] = sD[c * 33 + d];
            const double rdiag = sD[d * 33 + d] != 0.0 ? fast_reciprocal(sD[d * 33 + d]) : 0.0;
            back_chain(v, u, rdiag, nbk, d, half, std::make_integer_sequence<int, 32>{});
            sX[d * kTJ + jj] = on ? v : 0.0;
            if (on) a.X[(size_t)(kb + d) * a.nrhs + r0 + jj] = (float)v;
            __syncthreads();
            // y[0..kb) -= U[0..kb, kb..kb+nbk) x
            for (int i = tid; i < kb; i += kNT) {
                double uu[32];
#pragma unroll
                for (int c = 0; c < 32; ++c) uu[c] = a.W[(size_t)(kb + (c < nbk ? c : 0)) * ld + i];   // (x is zero beyond nbk)
                double acc[kTJ];
#pragma unroll
                for (int q = 0; q < kTJ; ++q) acc[q] = a.W[(size_t)(n + r0 + (q < nr ? q : 0)) * ld + i];
#pragma unroll
                for (int c = 0; c < 32; ++c) {
                    if (c % (32 / kTJ) == 0) {
#pragma unroll
                        for (int o = 0; o < kTJ; o += 8) DCX_PIN8(acc, o);
                    }
#pragma unroll
                    for (int q = 0; q < kTJ; ++q) acc[q] -= uu[c] * sX[c * kTJ + q];
                }
#pragma unroll
                for (int q = 0; q < kTJ; ++q)
                    if (q < nr) a.W[(size_t)(n + r0 + q) * ld + i] = acc[q];
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kNT) void lu_solve_kernel(const SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sL = smem;                                 // [32 * 33] L11 of the step (and the transposition tile, the diagonal
                                                       //            block of the back substitution)
    double* sU = sL + 32 * 33;                         // [32 * kTJ]
    double* sRow = sU + 32 * kTJ;                      // [32 (+ 32 spare)] the pivot row of a panel column
    int* sTop = reinterpret_cast<int*>(sRow + 64);     // [32]
    int* sLowDst = sTop + 32;                          // [32]
    int* sLowSrc = sLowDst + 32;                       // [32]
    int* sCnt = sLowSrc + 32;                          // [1 (+ 3 spare)]
    unsigned long long* sKey = reinterpret_cast<unsigned long long*>(sCnt + 4);   // [32] pivot keys, one per panel column
    const int tid = threadIdx.x, n = a.n, ncol = a.n + a.nrhs, G = gridDim.x;
    const size_t ld = (size_t)a.ld;
    unsigned n_bar = 0;
    // ---- W <- [A | B] transposed into column-major fp64: 32 x 32 tiles through LDS ---------------------------------------
    {
        float* sT = reinterpret_cast<float*>(sL);   // [32][33]
        const int tx = tid & 31, ty = tid >> 5;
        const int tiles_r = (n + 31) / 32, tiles_c = (ncol + 31) / 32;
        for (int t = blockIdx.x; t < tiles_r * tiles_c; t += G) {
            const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
            for (int rr = ty; rr < 32; rr += kNT / 32) {
                const int r = r0 + rr, c = c0 + tx;
                float v = 0.0f;
                if (r < n && c < ncol) v = c < n ? a.A[(size_t)r * n + c] : a.B[(size_t)r * a.nrhs + (c - n)];
                sT[rr * 33 + tx] = v;
            }
            __syncthreads();
            for (int cc = ty; cc < 32; cc += kNT / 32) {
                const int r = r0 + tx, c = c0 + cc;
                if (r < n && c < ncol) a.W[(size_t)c * ld + r] = (double)sT[tx * 33 + cc];
            }
            __syncthreads();
        }
    }
    DCX_STS(0);
    bool alive = grid_barrier(a.gs, n_bar, tid);
    DCX_STS(1);
    // ---- block steps, with a look-ahead of one panel ----------------------------------------------------------------------
    // Step s applies panel s to the trailing columns.  The groups that hold the NEXT panel's columns are updated first (one
    // workgroup each), workgroup 0 waits for exactly those - a counter, not a barrier - and then factorises that panel while
    // the other workgroups are still updating the rest: ONE grid barrier per step, and the
    // panel's dependent chain - 0.7 - 1 us per column on one wave per SIMD - runs beside the trailing update instead of in
    // front of it.  Every element still sees the same fused multiply-adds in
    // the same order.  The panels' row-movement lists are double-buffered by step parity.
    auto factor = [&](int k0, int nb, int buf) __attribute__((always_inline)) {
        switch (nbt_for(n - k0)) {
            case 32: factor_panel<32>(a.gs, buf, a.W, ld, n, k0, nb, sRow, sKey, sCnt, tid); break;
            case 16: factor_panel<16>(a.gs, buf, a.W, ld, n, k0, nb, sRow, sKey, sCnt, tid); break;
            case 8: factor_panel<8>(a.gs, buf, a.W, ld, n, k0, nb, sRow, sKey, sCnt, tid); break;
            default: factor_panel<4>(a.gs, buf, a.W, ld, n, k0, nb, sRow, sKey, sCnt, tid); break;
        }
    };
    // (tried first: workgroup 0 updating the next panel's columns itself, group after group - at n = 438 four groups in a row
    // cost more than the overlap returns, 0.87 -> 1.17 ms; with the counter both forms gain: 0.88 -> 0.81 ms at n = 438,
    // 3.04 -> 2.68 at 1000, 10.6 -> 9.0 at 2000.)  kLookAhead = false is the two-barrier step, kept for A/B.
    constexpr bool kLookAhead = true;
    // (iteration 0 applies nothing and factorises panel 0; iteration s >= 1 applies panel s - 1 from list buffer (s - 1) & 1
    // and factorises panel s into buffer s & 1; the one call site keeps the four unrolled panel bodies in the binary once)
    int k0 = 0, nb = 0, step = 0;
    unsigned panel_expected = 0;
    while (alive) {
        DCX_STS(8 * (step + 1));
        const int jt = k0 + nb;   // first trailing column (jt <= n < ncol: the right-hand sides are always there)
        const int nb1 = jt < n ? nb_for(n - jt) : 0;          // the next panel
        if (nb > 0) {
            const int buf = (step - 1) & 1;
            // this step's L11 and row movement into LDS
            for (int e = tid; e < nb * 32; e += kNT) {
                const int c = e >> 5, d = e & 31;
                sL[c * 33 + d] = (d < nb && d > c) ? a.W[(size_t)(k0 + c) * ld + k0 + d] : 0.0;
            }
            const int n_low = a.gs->n_low[buf];
            if (tid < 32) {
                sTop[tid] = tid < nb ? a.gs->top_src[buf][tid] : 0;
                sLowDst[tid] = tid < n_low ? a.gs->low_dst[buf][tid] : 0;
                sLowSrc[tid] = tid < n_low ? a.gs->low_src[buf][tid] : 0;
            }
            __syncthreads();
            const int groups = (ncol - jt + kTJ - 1) / kTJ;
            const int gp = (nb1 + kTJ - 1) / kTJ;             // the next panel sits in the first gp groups
            auto group = [&](int g) __attribute__((always_inline)) {
                const int j0 = jt + g * kTJ;
                trailing_group(a, k0, nb, j0, ncol - j0 < kTJ ? ncol - j0 : kTJ, sL, sU, sTop, sLowDst, sLowSrc, n_low, tid);
            };
            if (G == 1 || nb1 == 0 || !kLookAhead) {            // shared out evenly
                for (int g = blockIdx.x; g < groups; g += G) group(g);
            } else if (G - 1 >= gp) {
                // look-ahead: workgroups 1 .. gp update the next panel's columns FIRST, one group each, and say so; workgroup 0
                // waits for those gp groups only (not for a grid barrier) and factorises while everybody else goes on
                if (blockIdx.x == 0) {
                    panel_expected += (unsigned)gp;
                    if (tid == 0) {
                        const unsigned long long t0 = wall_clock64();
                        while (__hip_atomic_load(&a.gs->panel_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < panel_expected) {
                            __builtin_amdgcn_s_sleep(1);
                            if (__hip_atomic_load(&a.gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                            if (wall_clock64() - t0 > 200000000ull) {
                                __hip_atomic_store(&a.gs->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                    }
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                } else {
                    const int w = (int)blockIdx.x - 1;
                    if (w < gp) {
                        group(w);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        __syncthreads();
                        if (tid == 0) __hip_atomic_fetch_add(&a.gs->panel_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // the other groups start behind them (group gp + r goes to workgroup (gp + r) mod (G - 1)): a workgroup that
                    // did a panel group gets another one only when the groups wrap around the grid
                    for (int r = ((w - gp) % (G - 1) + (G - 1)) % (G - 1); r < groups - gp; r += G - 1) group(gp + r);
                }
            } else {                                              // (a grid smaller than a panel: workgroup 0 takes them itself)
                if (blockIdx.x == 0) for (int g = 0; g < gp && g < groups; ++g) group(g);
                else for (int g = gp + (int)blockIdx.x - 1; g < groups; g += G - 1) group(g);
            }
        }
        DCX_STS(8 * (step + 1) + 1);
        if (!kLookAhead && nb > 0 && nb1 > 0 && G > 1) {   // every workgroup's part of the next panel's columns must have landed
            alive = grid_barrier(a.gs, n_bar, tid);
            if (!alive) break;
        }
        if (blockIdx.x == 0 && nb1 > 0) factor(jt, nb1, step & 1);
        k0 = jt;
        nb = nb1;
        ++step;
        DCX_STS(8 * step + 2);
        alive = grid_barrier(a.gs, n_bar, tid);
        DCX_STS(8 * step + 3);
        if (nb == 0) break;
    }
    DCX_STS(2);
    if (blockIdx.x == 0) {
        if (alive) back_substitute(a, sL, sU, tid);
        __syncthreads();
        DCX_STS(3);
        if (tid == 0) {
            a.info[0] = alive ? a.gs->info : -1;
            a.info[1] = (int)n_bar;
        }
    }
}


size_t lds_bytes() { return sizeof(double) * (32 * 33 + 32 * kTJ + 64) + sizeof(int) * (96 + 4) + sizeof(unsigned long long) * 32 + 16; }

// cooperative on up to n_cu workgroups; one workgroup when that is refused, asked for, or the stream is being captured
hipError_t launch(const SolveArgs& a, int n_cu, bool one_workgroup, bool capturing, hipStream_t st) {
    const size_t lds = lds_bytes();
    int G = (a.n + a.nrhs + kTJ - 1) / kTJ;
    if (G > n_cu) G = n_cu;
    if (G < 1) G = 1;
    if (!one_workgroup && !capturing && G > 1) {
        SolveArgs copy = a;
        void* params[] = {(void*)&copy};
        const hipError_t e = hipLaunchCooperativeKernel((const void*)lu_solve_kernel, dim3(G), dim3(kNT), params, (unsigned)lds, st);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
    }
    lu_solve_kernel<<<dim3(1), dim3(kNT), lds, st>>>(a);
    return hipGetLastError();
}
