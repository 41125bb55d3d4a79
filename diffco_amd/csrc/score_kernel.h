// score_kernel.h — the fused hot path: FK -> pairwise kernel block -> weight contraction ->
// analytic gradient -> FK vjp, one launch.
//
// Replaces (paths under /root/reference/diffco): DiffCo.score / score_original
// kernel_perceptrons.py:359-370, DiffCo.poly_score :309-319, MultiDiffCo.score / rbf_score
// deprecated/MultiDiffCo.py:118-123,156-169, DiffCoBeta.rbf_score deprecated/DiffCoBeta.py:173-181,
// the kernels kernel.py:17-29,49-57,73-79, and the autograd pass the optimisers run through
// them (optim.py:101, 211-216).
//
// Mapping onto CDNA4
//   * one lane  = one configuration.  Its D features, D gradient accumulators and C score
//     accumulators stay in VGPRs for the whole support sweep (K[B,S] is never materialised).
//   * one wave  = 64 configurations x one contiguous slice of the supports.  The slice is
//     wave-uniform, so a support row (D coords + C weights [+ row-sum]) is fetched with
//     scalar loads (s_load_dwordx4/x8/x16 through the constant cache) into SGPRs and used as
//     the scalar operand of the VALU ops: the sweep issues no vector-memory and no LDS
//     instruction at all.  The path is fp32-VALU bound, and this layout
//     spends VALU slots only on the algorithmic sub/fma/rsq work.
//   * one block = the same 64 configurations x NW waves, each wave on its own support slice
//     (fills the chip when B is small); partial sums meet in LDS, every wave folds its share.
//   * prologue / epilogue: q rows staged coalesced through LDS, FK chain evaluated per lane with
//     frames kept in LDS (fk_device.h; DH arms: on several waves), J^T applied to the feature
//     gradient, gradient rows staged back through LDS for a coalesced store.
// Two forms of the pair body.  DIRECT: differences (x - s), squared, scaled by the coefficient - every
// kernel function, every input.  EXPANDED (XF, the default where it applies): d2 = |x|^2 + |s|^2 - 2 x.s
// and gX = x sum(c) - sum(c s) on data CENTRED on the support centroid, with a direct-form correction
// for near pairs where the kernel is not smooth (Polyharmonic(1)) and no correction where it is
// (RQKernel(2), FK-centred features only) - see "the expanded form" below.  This is NOT the reference's
// unguarded |x|^2+|s|^2-2x.s cdist GEMM, which is what puts its own fp32 result ~1e-5 off on raw
// coordinates: raw-input models with a sharp kernel (config #4) stay direct.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "fk_device.h"

namespace dcx {

// kernel-function specialisations compiled into the sweep
enum { KF_RQ2 = 0,   // RQKernel with p == 2 (the reference default, kernel.py:13)
       KF_POLY1 = 1, // Polyharmonic(k=1) (the reference's inference kernel, collision_checkers.py:206)
       KF_GEN = 2 }; // everything else, selected by a wave-uniform switch

// gradient modes
enum { MODE_SCORE = 0,    // score only
       MODE_GRAD_ROW = 1, // score + grad with the per-row folded weight (C == 1, or upstream == ones)
       MODE_GRAD_UP = 2 };// score + grad with an explicit upstream[b, :] (C > 1)

struct ScoreArgs {
    const float* rows;        // [S][RS] support rows: D coords, CC weights, (CC>1: sum of weights), |s|^2, pad
    const float* centre;      // expanded-form launches (XF kernels): the vector [D] that `rows` has been shifted by (the support
                              // centroid for features an FK transform produced, zero for raw inputs); the kernel shifts the
                              // features by the same vector; |s - c|^2 sits in the rows' last column
    const FkProg* fk;         // device copy of the compiled FK program
    const float* q;           // [B][dof]
    const float* upstream;    // [B][C] or null
    float* score;             // [B][C] or null
    float* grad;              // [B][dof] or null
    int64_t B;
    int32_t S;                // active supports
    int32_t s_chunk;          // supports per wave slice
    int32_t dof;
    int32_t d_fk;             // features the FK writes (<= D; the rest are zero padding)
    int32_t frame_floats;     // per-lane LDS floats for FK frames
    int32_t kind;             // DCX_K_* (used by KF_GEN)
    int32_t c_out;            // classes the CALLER has (<= the compiled CC: 3 runs as 4, 6 and 7 as 8 with zero weight columns -
                              // dcx_api.hip compiled_classes): row stride of upstream / score, and the columns written
    int32_t one_hot;          // MODE_GRAD_UP: >= 0 selects upstream = e_{one_hot} (Jacobian rows); -1 = use upstream[]
    int32_t nz;               // MODE_GRAD_UP, > 1: gridDim.z = nz classes in ONE launch, block z takes upstream = e_z and writes
                              // grad + z * dof (all Jacobian rows at once: the per-class sweeps of a small batch run side by side)
    int64_t grad_stride;      // floats between consecutive configurations' gradient rows (dof, or C*dof for jac)
    float* partial;           // split launch: per (tile, y) partial sums [(tile*ys + y)][ACC][64]; null = finish in-kernel
    const unsigned short* aplanes;  // XM sweep: the centred supports as bf16 planes laid out as MFMA A operands (xm_applies)
    int32_t xm;               // 1: the launch takes the distance of the expanded form from the matrix cores (score_kernel<..., XM>)
    unsigned int* tile_done;  // split launch: per-tile arrival counters (zero between launches).  Non-null: the LAST of a
                              // tile's ys blocks to arrive adds the partial rows and finishes in this launch; null: a
                              // second launch (score_finish_kernel) does
    unsigned long long* pwords;  // split launch, owner-polls hand-over (round 4): per (tile, y >= 1) rows of 8-byte (value, tag)
                              // words [(tile*ys + y)][ACC][64], all zero between launches; non-null selects that protocol
    uint32_t ptag;            // ... the tag of THIS launch's words (never 0; the host numbers the launches that use a buffer)
    int32_t* giveup;          // ... host-visible flag an owner sets when its peers never published (results NaN; the API refuses
                              // further launches of the model: dcx_api.hip run_score)
    int32_t ys;               // support super-chunks (gridDim.y); block y sweeps [y*s_super, (y+1)*s_super)
    int32_t s_super;
    int32_t red_slots;        // LDS rows for the cross-wave fold: nw (all waves write, fold in parallel) or 1 (waves
                              // take turns through one row: wide shapes, where nw rows would cost occupancy)
#ifdef DCX_TIMING
    unsigned long long* ts;   // developer builds: [32 slots][16 waves] cycle stamps of block (ts_block,0)
    unsigned int ts_block;
#endif
    float kp0, kp1;           // kernel parameters
    int32_t mfma;             // 1: the launch uses the MFMA form of the gradient fold (score_kernel<..., MF = true>)
    int32_t xf;               // 1: the launch uses the expanded form of the sweep (score_kernel<..., XF = true>)
    int32_t fkk;              // DCX_FK_DH only: 1 = the FK walks read the program with scalar loads (fk_*_dh_k), 2 = the step
                              // table (fk_device.h dh2_*; `dh` below); 0 = every kind: FkProg interpreted from its LDS copy
    int32_t fk_dwords;        // dwords of FkProg the transform uses (staged into LDS; the host knows it: no dependent load)
    DhArgs dh;                // fkk == 2: the step table's control part (counts and masks live in SGPRs)
    int32_t jt_rows;          // fkk == 2, chains of <= kDhUnroll steps, >= 2 waves per chain: the chain runs row-split
    int32_t jt_waves;         // jt_rows and the block folds in parallel with room in its scratch rows for 12 columns per
                              // point step (+ 1): J^T runs on several waves (fk_device.h dh2_vjp_r1_sel / r1b / r2_sel)
    int32_t hinge;            // 1 (C == 1): gradient of weight * clamp(score - margin, 0) instead of the score's;  2 (MODE_GRAD_UP,
                              // C > 1): `upstream` holds this batch's SCORES (an earlier score-only launch) and the sweep's upstream
                              // is hinge_weight * 1[score_c - hinge_margin_c[c] > 0]: the gradient of
                              // weight * sum_c clamp(score_c - margin_c, 0) (optim.py:88-89 on a MultiDiffCo, scripts/active.py:65)
    float hinge_margin, hinge_weight;
    float hinge_margin_c[8];  // hinge == 2: the per-class margins (DCX_MAX_C)
    int32_t s_skew;           // 16-wave blocks: per mille of s_chunk by which the slices of a block's wave GROUPS differ (wave_slice below)
    int32_t qt;               // 1: the quarter-tile form (score_kernel<..., QT>): 16 configurations per block, rows from LDS
    int32_t qt_off;           // ... float offset of the rows' copy inside the block's LDS
    int32_t qt_per;           // ... rows per slice (kQtSlices * nw slices; a slice starts four banks behind the one before)
    int32_t qt_per_g[4];      // ... per wave group (waves 4 g .. 4 g + 3: 16 slices): equal to qt_per, or skewed like wave_slice's shares
                              //     (sum over the groups = (nw / 4) qt_per: the same rows, the same LDS)
    int32_t qt_scr;           // ... float offset of J^T's own scratch columns (phase R1 runs beside the sweep's tail)
    int32_t qt_front;         // ... rows copied into LDS before the first barrier (the rest during the FK chain)
};

// floats a multi-class row's stride is rounded up to (developer A/B: 16 / 32 put every row on its own 64 / 128-byte lines)
#ifndef DCX_ROW_ALIGN_MULTI
#define DCX_ROW_ALIGN_MULTI 4
#endif
template <int D, int CC>
struct RowLayout {
    static constexpr int W_OFF = D;
    static constexpr int WSUM_OFF = D + CC;                   // only present when CC > 1
    static constexpr int SS_OFF = D + CC + (CC > 1 ? 1 : 0);  // |s|^2 (float64 sum rounded once; the expanded-form sweep)
    static constexpr int AL = CC > 1 ? DCX_ROW_ALIGN_MULTI : 4;
    static constexpr int RS = (SS_OFF + 1 + AL - 1) / AL * AL;
};

// This wave's slice [j0, j1) of its block's supports [ybase, yend).  Equal slices of s_chunk rows - unless skew > 0 (16-wave blocks):
// the SIMD's arbiter issues oldest-first, so the four waves a block has on a SIMD do not advance together - the stamps of one block
// of the headline launch (tools/phase_timing.py) have waves 0-3 leave the sweep after 22 k cycles, 4-7 after 30 k, 8-11 after 41 k and
// 12-15 after 51 k, the last group running nearly alone (and a lone wave uses a third of a SIMD's issue slots) while the first waits
// at the barrier.  With skew the groups take unequal shares of the block's rows (packed per mille, below; the rules: dcx_api.hip
// skew_rule / skew8_rule), so that the groups finish closer together.  The slices still tile [ybase, ybase + 16 s_chunk) in wave
// order: the fold's order of the sums is unchanged, the sums themselves are those of the new slices.
__device__ __forceinline__ void wave_slice(int wave, int nw, int s_chunk, int skew, int ybase, int yend, int& j0, int& j1) {
    int start = wave * s_chunk, len = s_chunk;
    if (skew > 0 && nw == 16) {
        // skew = w0 | w1 << 10 | w2 << 20: the per-mille shares of the block's 16 s_chunk rows that wave groups 0, 1, 2 take (group 3:
        // the rest); a group's four waves share its rows equally, lengths even
        const int w0 = skew & 1023, w1 = (skew >> 10) & 1023, w2 = (skew >> 20) & 1023;
        const int l0 = (s_chunk * w0 / 250) & ~1, l1 = (s_chunk * w1 / 250) & ~1, l2 = (s_chunk * w2 / 250) & ~1;
        int l3 = 4 * s_chunk - l0 - l1 - l2;
        l3 = l3 > 0 ? l3 : 0;
        const int g = wave >> 2, q = wave & 3;
        len = (g == 0) ? l0 : (g == 1) ? l1 : (g == 2) ? l2 : l3;
        start = 4 * ((g > 0 ? l0 : 0) + (g > 1 ? l1 : 0) + (g > 2 ? l2 : 0)) + q * len;
    } else if (skew < 0 && nw == 8) {
        // 8-wave blocks (two wave groups): -skew = the per-mille share of the block's 8 s_chunk rows that waves 0-3 take
        int l0 = (s_chunk * (-skew) / 500) & ~1;
        l0 = l0 < 2 * s_chunk ? l0 : 2 * s_chunk;
        const int g = wave >> 2, q = wave & 3;
        len = g ? 2 * s_chunk - l0 : l0;
        start = (g ? 4 * l0 : 0) + q * len;
    }
    j0 = (ybase + start < yend) ? ybase + start : yend;
    j1 = (j0 + len < yend) ? j0 + len : yend;
}

// arrival counters of a split launch sit one per 128-byte line: 256 blocks bumping 64 counters inside one line serialise
// at the memory-side atomic unit (~12 ns each; the count took 2.5 k cycles instead of ~0.7 k)
constexpr int kCounterStride = 32;

typedef const __attribute__((address_space(4))) float* cfloat_ptr;
typedef float v2f __attribute__((ext_vector_type(2)));

// Developer-only phase timing (build with -DDCX_TIMING): block (0,0) records s_memtime at checkpoints into
// a device symbol that tools/phase_timing.py reads back.  Not compiled into the shipped library.
#ifdef DCX_TIMING
#define DCX_TS(slot)                                                                              \
    do {                                                                                          \
        if (a.ts && blockIdx.x == a.ts_block && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 16) \
            a.ts[(slot) * 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();               \
    } while (0)
// per-block stamps (wave 0 of every block, y == 0): k = 0 start, 1 sweep start, 2 sweep end, 3 end
#define DCX_TSB(k)                                                                                 \
    do {                                                                                           \
        const unsigned bl__ = blockIdx.x + gridDim.x * blockIdx.y;                                \
        if (a.ts && bl__ < 4096 && threadIdx.x == 0) {                                             \
            unsigned long long t__ = __builtin_readcyclecounter();                                 \
            if ((k) == 0) {                                                                        \
                unsigned hw__, xcc__;                                                              \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__));                 \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__));               \
                a.ts[512 + bl__ * 4 + 3] = ((unsigned long long)xcc__ << 32) | hw__;               \
                a.ts[512 + bl__ * 4 + 0] = t__;                                                    \
            } else if ((k) == 3) {                                                                 \
                a.ts[512 + bl__ * 4 + 2] = t__;                                                    \
            } else {                                                                               \
                a.ts[512 + bl__ * 4 + 1] = t__;                                                    \
            }                                                                                      \
        }                                                                                          \
    } while (0)
#else
#define DCX_TS(slot) do { } while (0)
#define DCX_TSB(k) do { } while (0)
#endif

// Code-generation choices of the sweep, each A/B-measured on MI355X against the alternatives (the variants lived
// behind a build macro during the round; profiles/r01_sweep_variants.txt has the numbers, DESIGN.md the reasoning):
//   * differences, squared distance and gradient accumulators as packed pairs end to end (v_pk_add / v_pk_fma);
//   * an EXPLICIT software pipeline of the scalar row loads (wait -> issue next -> consume) instead of the
//     compiler's schedule (15-20 % slower everywhere), four rows in flight for narrow rows (two were 11 % slower at
//     small batches), fewer for wide rows (below).
// four-row pipeline limits (SGPRs in flight; see PARTS below), measured: C > 1 holds more scalars of its own (C = 5,
// D = 12: 72 row SGPRs park 96 lane moves per 4 rows, two rows in flight +22..45 %); C = 1: D = 21 (88) is 9..21 %
// faster with two rows in flight; D = 16 / 18 (68 / 76) sit on the edge — whether the compiler parks there changes
// with unrelated code (a 13 % swing at D = 18) — so they take the two-row pipeline as well
// (profiles/r01_sweep_variants.txt, tools/check_sgpr_parking.py)
#ifndef DCX_P0_MAX_MULTI
#define DCX_P0_MAX_MULTI 56
#endif
#ifndef DCX_P0_MAX_SINGLE
#define DCX_P0_MAX_SINGLE 56
#endif
// independent accumulator pairs for the squared distance (see pair())
#ifndef DCX_D2_MULTI
#define DCX_D2_MULTI 1
#endif
#define DCX_D2_ACCS(D) (!DCX_D2_MULTI ? 1 : (D) >= 32 ? 4 : (D) >= 16 ? 2 : 1)
// narrow one-class rows in the direct form: the two rows of a pipeline stage share every packed instruction (see pair2)
#ifndef DCX_PAIR2
#define DCX_PAIR2 1
#endif
#ifndef DCX_PAIR2_MAX_D
#define DCX_PAIR2_MAX_D 8
#endif
// ... and are fetched together: two adjacent rows = ONE scalar load, one address per stage (see the P2 pipeline)
// Round 6: the rows such a sweep reads are stored PAIR-INTERLEAVED (dcx_api.hip rows_p2: rows 2i and 2i + 1 as (r0_0, r1_0, r0_1, r1_1,
// ..., w0, w1, ...), an odd count padded with a zero-weight row), so the operand pair (r0_k, r1_k) of every packed instruction IS
// an aligned SGPR pair of the stage's one scalar load: the two s_mov per feature that used to build it (6 of the 8.9 SALU
// instructions per pair at config #4, profiles/pmc_cfg4.json - one scalar unit serves a CU's four SIMDs) are gone.  Slices start
// on even rows (the host rounds s_chunk / s_super up), a slice that ends on the model's odd last row runs into the padding row.
#ifndef DCX_PAIR2_LOADS
#define DCX_PAIR2_LOADS 1
#endif
// rows fetched in whole groups of four floats by the four-row pipeline (see load_row)
#ifndef DCX_LOAD_GROUPS
#define DCX_LOAD_GROUPS 1
#endif

// value K(d2) and g with dK/dx = g * (x - s)
// CLAMPED: the caller guarantees d2 >= 1e-30 (the expanded-form sweep clamps at its near threshold)
template <int KF, bool CLAMPED = false>
__device__ __forceinline__ void kernel_eval(float d2, const ScoreArgs& a, float& val, float& g) {
    if constexpr (KF == KF_RQ2) {
        // (1 + gamma/2 d2)^-2 ;  dK/dd2 = -gamma (1 + gamma/2 d2)^-3
        const float t = fmaf(0.5f * a.kp0, d2, 1.0f);
        const float u = __builtin_amdgcn_rcpf(t);
        val = u * u;
        g = (-2.0f * a.kp0) * (val * u);
    } else if constexpr (KF == KF_POLY1) {
        // r (1/eps folded into the row weights); sub-gradient 0 at r == 0 because delta == 0 there
        const float d2c = CLAMPED ? d2 : fmaxf(d2, 1e-30f);
        const float ri = __builtin_amdgcn_rsqf(d2c);
        val = d2c * ri;
        g = ri;
    } else {
        if (a.kind == DCX_K_RQ) {
            const float p = a.kp1;
            const float t = fmaf(a.kp0 / p, d2, 1.0f);
            val = powf(t, -p);
            g = (-2.0f * a.kp0) * val * __builtin_amdgcn_rcpf(t);
        } else if (a.kind == DCX_K_POLY) {
            const int k = (int)a.kp0;
            const float ie = 1.0f / a.kp1;
            const float d2c = fmaxf(d2, 1e-30f);
            const float ri = __builtin_amdgcn_rsqf(d2c);
            const float r = d2c * ri;
            float rk2 = (k >= 2) ? 1.0f : ri;  // r^(k-2)
            for (int i = 2; i < k; ++i) rk2 *= r;
            if (k & 1) {
                val = rk2 * d2c * ie;
                g = (float)k * rk2 * ie;
            } else {
                const float lg = 0.5f * logf(d2c);
                val = rk2 * d2c * lg * ie;
                g = rk2 * fmaf((float)k, lg, 1.0f) * ie;
            }
        } else {  // DCX_K_MQ
            const float ie2 = 1.0f / (a.kp0 * a.kp0);
            const float v2 = fmaf(d2, ie2, 1.0f);
            const float rv = __builtin_amdgcn_rsqf(v2);
            val = v2 * rv;
            g = rv * ie2;
        }
    }
}

// ---- RQKernel(p = 2) inside the sweeps: constants folded (round 4) --------------------------------------------------------
// K = (1 + gamma/2 d2)^-2 = (2/gamma)^2 / t^2 with t = d2 + 2/gamma, and dK/dx = -2 gamma (1 + gamma/2 d2)^-3 (x - s) =
// -4 (2/gamma)^2 / t^3 (x - s).  dcx_model_create stores the row weights of an RQ2 model as w (2/gamma)^2; the squared-
// distance accumulator of a pair starts at 2/gamma instead of zero (ScoreArgs::kp0 carries 2/gamma for these launches), so
// t costs nothing; the pair body is u = 1/t, val = u u, g = val u (3 instead of 5 instructions), and the feature gradient
// is multiplied by -4 ONCE per lane after the sweep (exact).  17 -> 15 VALU instructions per pair at D = 6 (config #4),
// 28 -> 26 at D = 12, C = 5.  kernel_eval<KF_RQ2> above stays the textbook form (dcx_kernel_matrix uses it).
template <int KF>
__device__ __forceinline__ float d2_seed(const ScoreArgs& a) {
    if constexpr (KF == KF_RQ2) return a.kp0;
    else return 0.0f;
}
template <int KF>
inline constexpr float kGradScale = (KF == KF_RQ2) ? -4.0f : 1.0f;
// value and gradient coefficient inside a sweep; `t` = the accumulated d2 INCLUDING d2_seed
template <int KF, bool CLAMPED = false>
__device__ __forceinline__ void sweep_eval(float t, const ScoreArgs& a, float& val, float& g) {
    if constexpr (KF == KF_RQ2) {
        const float u = __builtin_amdgcn_rcpf(t);
        val = u * u;
        g = val * u;
    } else {
        kernel_eval<KF, CLAMPED>(t, a, val, g);
    }
}

// LDS carve (floats).  Everything per-lane is in column layout [e][64].
struct LdsPlan {
    int fk, q, x, g, f, red, total;
};
__host__ __device__ inline LdsPlan lds_plan(int dof, int d_fk, int frame_floats, int red_slots, int acc_floats,
                                            bool alias_xg = false) {
    LdsPlan p;
    p.x = 0;
    // The fused kernel writes G only after the sweep, when X is dead (alias_xg); the split launch's finish kernel
    // needs both at once.
    p.g = alias_xg ? p.x : p.x + 64 * d_fk;
    // The reduction scratch OVERLAYS X and G: X is dead once every wave has copied its features into registers
    // (the kernel barriers after that copy), G is written only after wave 0 has read every partial.
    p.red = p.x;
    const int end_xg = p.g + 64 * d_fk;
    const int end_red = p.red + (red_slots < 1 ? 1 : red_slots) * acc_floats * 64;  // always one row: the one-wave hand-over's totals
    p.q = end_xg > end_red ? end_xg : end_red;
    p.f = p.q + ((64 * dof + 3) & ~3);
    p.fk = p.f + 64 * frame_floats;              // the FK program comes last: its size varies with the robot
    p.total = p.fk;                              // and only the host needs it (+ fk_prog_floats)
    return p;
}

// waves per SIMD the register allocator is asked to make room for (gfx950: 512 VGPRs per lane per SIMD, 8-register
// granules: <=64 -> 8 waves, 72 -> 7, 80 -> 6, 96 -> 5, 128 -> 4, 168 -> 3, 256 -> 2).  The sweep keeps about
// 3*D + 2*C + 16 values live; asking for more waves than that allows would spill inside the hot loop.
#ifndef DCX_MINW_SLACK
#define DCX_MINW_SLACK 0
#endif
// The XM sweep (round 3): the expanded form with its distance GEMM x . s^T on v_mfma_f32_16x16x32_bf16 in split operands
// (sweep_rows, XM).  One class, Polyharmonic(1), even D <= 16 (a term's 16 K slots hold the features).
// (compiled for the two widths it was measured at: profiles/r03_mfma_ab.txt - measured slower, kept as evidence)
constexpr bool xm_applies(int D, int CC, int KF) { return KF == 1 /* KF_POLY1 */ && CC == 1 && (D == 12 || D == 16); }
// Does the EXPANDED form of this shape take two rows per packed instruction (sweep_rows, X2; round 6)?  One class, the two
// specialised kernel functions, even D <= DCX_XF2_MAX_D (2 D gradient accumulators).  Such a model's centred rows are stored
// pair-interleaved as well (dcx_api.hip rows_x2), its expanded sweeps run on even-aligned slices.
// MEASURED, NOT SHIPPED (default 0; `make EXTRA="-DDCX_XF2=1 -DDCX_XF2_MINW=4"` builds it - profiles/r06_xf2.txt): the body needs ~70
// VGPRs.  Under the sweep kernel's 64-register budget (two 16-wave blocks per CU) it spills into the loop; given 128 registers a
// CU holds ONE block, whose prologue and epilogue nothing overlaps any more: headline 85.0 -> 104.0 us at B = 65536, 1200 -> 1217 us
// at B = 1 M, while the launches that run one block per CU anyway gain 3 - 5 % (B = 8192: 19.1 -> 18.6 us, config #5's persistent
// kernel 28.2 -> 27.4 us per iteration).  Not worth a second expanded form of every narrow kernel.
#ifndef DCX_XF2
#define DCX_XF2 0
#endif
#ifndef DCX_XF2_MAX_D
#define DCX_XF2_MAX_D 12
#endif
constexpr bool x2_applies(int D, int CC, int KF) {
    return DCX_XF2 && CC == 1 && (D % 2) == 0 && D >= 4 && D <= DCX_XF2_MAX_D && (KF == 1 /* KF_POLY1 */ || KF == 0 /* KF_RQ2 */) &&
           4 * (D + 2) <= DCX_P0_MAX_SINGLE;
}
#ifndef DCX_XF2_MINW
#define DCX_XF2_MINW 8   // waves per SIMD the register allocator must leave room for in such a kernel (8 = 64 VGPRs)
#endif
constexpr int sweep_min_waves(int D, int CC, int KF, bool MF = false, bool XM = false, bool QT = false, bool XF = false) {
    if (QT) return 4;  // one block per CU (the rows fill its LDS), at most 16 waves: two row buffers in VGPRs + the direct body
    if (XM) return 4;  // 48 VGPRs of loop-invariant B fragments + 16 distances in flight: 128 VGPRs
    if (XF && x2_applies(D, CC, KF)) return DCX_XF2_MINW;
    // KF_GEN calls powf/logf; the MFMA form adds 16 accumulator registers per contraction + the operand fragments
    const int need = 3 * D + 2 * CC + 16 + DCX_MINW_SLACK + (KF == 2 ? 40 : 0) + (MF ? (CC > 1 ? 48 : 24) : 0);
    return need <= 64 ? 8 : need <= 72 ? 7 : need <= 80 ? 6 : need <= 96 ? 5 : need <= 128 ? 4 : need <= 168 ? 3 : need <= 256 ? 2 : 1;
}

// The lane index, derived afresh from the execution mask (all lanes active) behind a compiler barrier.  The kernel
// re-derives it after the sweep: everything computed from the prologue's copy (LDS column addresses, row offsets)
// would otherwise stay live across the sweep, and at the sweep's 64-VGPR budget the allocator spilled exactly those
// to scratch (8 B per lane per launch = 8 MB of HBM writes at B = 65536, and reloads inside the lone-wave FK code).
__device__ __forceinline__ int fresh_lane() {
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}

// The kernel's arguments, read AFRESH from the kernarg segment behind a compiler barrier.  The epilogue uses this copy:
// anything of `a` that both the prologue and the epilogue touch would otherwise stay live in SGPRs across the sweep, whose
// scalar row pipeline needs every SGPR the wave has - the allocator then parks row registers in VGPR lanes INSIDE the
// hot loop (v_writelane / v_readlane: +20 instructions per 4 rows when round 3 first added arguments for the epilogue;
// tools/check_sgpr_parking.py).  The argument struct is the kernel's only parameter, so it sits at offset 0 of the segment.
template <class Args>
__device__ __forceinline__ const __attribute__((address_space(4))) Args& reload_kernargs() {
    auto p = (const __attribute__((address_space(4))) Args*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
}
__device__ __forceinline__ const __attribute__((address_space(4))) ScoreArgs& reload_args() { return reload_kernargs<ScoreArgs>(); }
// member by member: the reloaded arguments live in the constant address space
#define DCX_COPY_DH(dst, src)          \
    do {                               \
        (dst).prog = (src).prog;       \
        (dst).n_dwords = (src).n_dwords; \
        (dst).n_steps = (src).n_steps; \
        (dst).end0 = (src).end0;       \
        (dst).n_chains = (src).n_chains; \
        (dst).pt = (src).pt;           \
        (dst).bare = (src).bare;       \
        (dst).real = (src).real;       \
        (dst).n_pt = (src).n_pt;       \
        (dst).shared_q = (src).shared_q; \
    } while (0)

// ---- the two-buffer pipeline of wide rows, round 4 ---------------------------------------------------------------------
// C > 1 rows (18-20 floats) run  wait -> issue B -> consume A -> wait -> issue A -> consume B  between sched_barriers.  The
// machine scheduler respects those, but the IR reached it already rearranged (rocprofv3 counters of config #3 at B = 65536:
// 47 % of wave-cycles in s_waitcnt, VALU 62 % busy, 10 SALU per row: profiles/r04_direct_pre_cfg3_b65536_rocprof_summary.txt):
//   * the SLP vectoriser paired row A's kernel function and coefficient with row B's (v_pk_mul across the two rows).  That
//     ties A's gradient fold to B's distance: row B's s_load_dwordx8 pair ended up BELOW `consume A`, next to its first
//     use - every second row waited out a full scalar-cache / L2 round trip - and A's coordinates lived on in twelve
//     s_mov copies.  row_opaque below makes a row's kernel values and coefficient opaque where they are formed, so a body
//     is complete before the next begins, and the loads stay where the source puts them.
//   * nothing reads the class scores before the end of the sweep, so their updates were sunk into the loop latch, again
//     with the rows' weights kept alive in copies: explicit packed accumulators, pinned (sweep_rows add_scores).
// Tried and dropped: issuing the row loads as `asm volatile` s_load (program order binding, explicit wait, values handed
// over through an empty asm).  Order was right, but the compiler does not know a register is still in flight: a phi copy
// of the x8 tail in front of the wait (D = 21, C = 1) read it before the data had landed - wrong results, caught by
// tools/xf_rq_rule.py.  Loads the compiler tracks itself cannot have that problem.  (-mllvm -pre-RA-sched=source also
// restores the order, but once the bodies no longer pair up the default scheduler leaves it alone too.)
// keeps the SLP vectoriser from pairing one row's kernel function with the next row's
__device__ __forceinline__ void row_opaque(float& u, float& v) { asm volatile("" : "+v"(u), "+v"(v)); }

// ---- the sweep: supports [j0, j1) against this lane's configuration, rows broadcast through SGPRs ---------------
// Accumulates into sc[] (scores) and gx[] (feature gradient; untouched for MODE_SCORE).  A function of its own so that
// kernel variants can share it (e.g. the two-tile helper-wave experiment of DESIGN.md 3.1).
// ---- the expanded form (XF) ------------------------------------------------------------------------------------------
// The direct form spends D/2 v_pk_add per pair on the differences x - s, which only exist to be squared and to be
// scaled by the gradient coefficient.  Expanded,
//     d2        = (|x|^2 + |s_j|^2) + sum_k (-2 x_k) s_jk            one v_add + D/2 v_pk_fma, s_jk straight from SGPRs
//     gX[b, :]  = x_b * sum_j c_bj  -  sum_j c_bj s_j                D/2 v_pk_fma + one v_add per pair, s_jk from SGPRs
// needs no difference at all: 21 instead of 24 VALU instructions per pair at D = 12 (the six dropped are the 4.4-cycle
// packed adds: -17 % issue cycles), and D fewer live registers.  Both sums cancel when x is close to s_j, so:
//   * a pair with d2 < thr = DCX_XF_TAU * |x|^2 (r below a tenth of |x|) is a NEAR pair.  The hot loop stays branch-free:
//     every pair goes through the expanded form with d2 clamped at thr, so a near pair adds a bounded (wrong)
//     term.  After each two-row stage of the pipeline one ballot asks whether any lane saw d2 <= thr; if so — rare — a
//     correction block takes the expanded term of the near (lane, row) pairs out again (the same operations on the same
//     operands reproduce it exactly; what is left is one rounding of a bounded number) and adds the DIRECT form:
//     differences, squared distance and gradient term exactly as the direct sweep computes them (r = 0 stays exact).
//     Lanes that are not near add +-0 in that block, so a configuration's result does not depend on which other
//     configurations share its wave.
//   * the expanded gradient lives in ONE accumulator set H = sum c s - (what has been folded so far); every DCX_XF_FLUSH
//     rows the run's x * sum(c) is folded in place (H <- H - x A; A <- 0), which realises the cancellation while the
//     run's sums are still small, so what H carries between runs is (minus) the true partial gradient.  No second
//     accumulator set: the sweep holds -2x, H and a handful of temporaries — 12 VGPRs fewer than the direct form at
//     D = 12 (with a second set the 64-register budget spilled long-lived values into scratch, and their reloads inside
//     the single-wave FK / J^T code cost 2 us per block at small batches).
// tools/split_numerics.py sizes the rounding error of exactly this arithmetic against float64: gradient 2e-7 .. 4e-6
// relative (the direct form: 1e-6 .. 2e-6), score unchanged.  |s_j|^2 rides in the support row (RowLayout::SS_OFF).
// Rows wider than 38 floats are consumed in parts whose SGPRs are gone when the coefficient is known: they keep the
// direct form.
#ifndef DCX_XF_TAU
#define DCX_XF_TAU 0.01f
#endif
#ifndef DCX_XF_FLUSH
#define DCX_XF_FLUSH 64
#endif

// Does the expanded form exist for this shape?  Whole rows must sit in SGPRs when the coefficient is known, and the
// kernel function must tolerate the expanded distance: its absolute error is ~ 2^-23 (|x|^2 + |s|^2) however small d2
// is.  Polyharmonic(k = 1) — the reference's inference kernel, K = r — turns that into an absolute score error far
// below the sum's own rounding (measured: the expanded form is as close to float64 as the direct one, gradient closer).
// A sharp kernel does not: RQ(gamma = 10) weighs exactly the pairs with small d2, and on config #4's data (|x|^2 ~ 100)
// the expanded form was 7e-6 from float64 where the direct form is 5e-7 — inside the 1e-5 bar but not by a margin
// worth 3-7 %, so every kernel other than Polyharmonic(1) keeps the direct form.
constexpr bool xf_applies(int D, int CC, int KF) {
    const int used = D + CC + (CC > 1 ? 1 : 0);
    const int parts = (4 * used <= (CC > 1 ? DCX_P0_MAX_MULTI : DCX_P0_MAX_SINGLE)) ? 0 : (used + 37) / 38;
    return (KF == KF_POLY1 || KF == KF_RQ2) && used + 1 <= 38 && parts <= 1;
}
// Does a launch of this shape in the DIRECT form take two rows per packed instruction (sweep_rows, P2) - and therefore read the
// pair-interleaved copy of the rows on even-aligned slices?  (the host's mirror of sweep_rows' own condition)
constexpr bool p2_applies(int D, int CC, int KF) {
    return DCX_PAIR2 && DCX_PAIR2_LOADS && CC == 1 && (D % 2) == 0 && D <= DCX_PAIR2_MAX_D && 4 * (D + 1) <= DCX_P0_MAX_SINGLE &&
           (KF == KF_RQ2 || KF == KF_POLY1);
}
// Round 4: RQKernel(p = 2) takes the expanded form as well, on FK-CENTRED features only (the host's rule, dcx_api.hip
// xf_rq_ok: gamma * max |s - c|^2 <= 32, a transform present).  The kernel is smooth at d2 = 0, so there is no near-pair
// block: the expanded distance's absolute error 2^-23 (|x - c|^2 + |s - c|^2) moves K by at most gamma times that (~2e-6
// at arm scale, measured 1.7e-6 on config #3 against 3.8e-7 for the direct form: inside the 1e-5 bar with a margin),
// and the pair body drops the six packed differences AND the clamp / ballot: 28 -> 22 VALU instructions at D = 12, C = 5.
// The rows of the centred copy carry |s - c|^2 + 2/gamma in their last column (sweep_eval's t needs no add).  Raw-input
// models (config #4: |x| ~ 10) keep the direct form: there the same error would be 5e-4.

// NACC > 0 overrides the number of independent squared-distance accumulator pairs of the expanded form (callers that run
// at few waves per SIMD trade one packed add per row for a shorter dependent chain)
// NS (several classes, MODE_GRAD_UP): no score accumulation - a caller that already holds this batch's class scores (the persistent
// trajectory kernel's second sweep) saves the CC fma per pair; sc[] comes back untouched
template <int D, int KF, int CC, int MODE, bool XF = false, int NACC = 0, bool XM = false, bool NS = false>
__device__ __forceinline__ void sweep_rows(const ScoreArgs& a, const float (&x)[D], const float (&up)[CC], int j0, int j1,
                                           float (&sc)[CC], float (&gx)[D]) {
    static_assert(!NS || (CC > 1 && MODE == MODE_GRAD_UP), "NS: the gradient sweep of a multi-class model");
    using L = RowLayout<D, CC>;
    constexpr bool GRAD = (MODE != MODE_SCORE);
    v2f gx2[D / 2 + 1];
#pragma unroll
    for (int k = 0; k < D / 2 + 1; ++k) gx2[k] = v2f{0.0f, 0.0f};
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)a.rows;
    // only the floats a row really carries are loaded (the tail of the padded stride is never touched)
    constexpr int USED_DIRECT = D + CC + (CC > 1 ? 1 : 0);
    constexpr bool XFA = XF && xf_applies(D, CC, KF);
    constexpr int USED = USED_DIRECT + (XFA ? 1 : 0);
    constexpr int RSTRIDE = L::RS;
    // (how a row travels through the SGPRs: 0 = four whole rows in flight, 1 = two whole rows, >= 2 = parts of a row; see below)
    constexpr int PARTS = (4 * USED <= (CC > 1 ? DCX_P0_MAX_MULTI : DCX_P0_MAX_SINGLE)) ? 0 : (USED + 37) / 38;  // parts of <= 38 floats
    // Two-buffer pipeline, several classes: the class scores accumulate as explicit packed pairs.  Left as CC scalar fmaf
    // chains the SLP vectoriser packs them itself - across BOTH rows of the loop body, which moves row A's score updates
    // behind row B's body and keeps A's weights alive in copies (see row_opaque).  Same sums, class by class.
    constexpr bool SC2 = (CC > 1 && PARTS == 1);
    v2f sc2[CC / 2 + 1];
#pragma unroll
    for (int i = 0; i < CC / 2 + 1; ++i) sc2[i] = v2f{0.0f, 0.0f};
    auto add_scores = [&](auto weight_of, float val) __attribute__((always_inline)) {
        if constexpr (NS) {
            (void)val;
        } else if constexpr (SC2) {
            const v2f v2 = {val, val};
#pragma unroll
            for (int c = 0; c + 1 < CC; c += 2) sc2[c / 2] = __builtin_elementwise_fma(v2f{weight_of(c), weight_of(c + 1)}, v2, sc2[c / 2]);
            if constexpr (CC & 1) sc[CC - 1] = fmaf(weight_of(CC - 1), val, sc[CC - 1]);
            // pinned where they stand: nothing reads the score accumulators before the end of the sweep, so LLVM sinks these
            // updates into the loop latch (behind the flush branch) - with the row's weights kept alive in copies
#pragma unroll
            for (int c = 0; c + 1 < CC; c += 2) asm volatile("" : "+v"(sc2[c / 2]));
            if constexpr (CC & 1) asm volatile("" : "+v"(sc[CC - 1]));
        } else {
#pragma unroll
            for (int c = 0; c < CC; ++c) sc[c] = fmaf(weight_of(c), val, sc[c]);
        }
    };

    // expanded-form state: -2 x (packed), |x|^2, the near threshold (also the hot path's clamp), H and the run's sum(c)
    v2f xm[D / 2 + 1], ga[D / 2 + 1];
    float xm_tail = 0.0f, ga_tail = 0.0f, xx = 0.0f, thr = 1e-30f, asum = 0.0f;
    if constexpr (XFA) {
#pragma unroll
        for (int k = 0; k < D; ++k) xx = fmaf(x[k], x[k], xx);
        thr = fmaxf(DCX_XF_TAU * xx, 1e-30f);
#pragma unroll
        for (int k = 0; k + 1 < D; k += 2) {
            xm[k / 2] = v2f{-2.0f * x[k], -2.0f * x[k + 1]};
            ga[k / 2] = v2f{0.0f, 0.0f};
        }
        if constexpr (D & 1) xm_tail = -2.0f * x[D - 1];
    }
    // squared distance of one support row in the expanded form, clamped at thr (== thr marks a near pair)
    auto d2_x = [&](const auto& r) __attribute__((always_inline)) -> float {
        constexpr int NA = NACC > 0 ? NACC : DCX_D2_ACCS(D);
        v2f acc[NA];
        acc[0] = v2f{xx + r[L::SS_OFF], 0.0f};
#pragma unroll
        for (int i = 1; i < NA; ++i) acc[i] = v2f{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k + 1 < D; k += 2) {
            const v2f rv = {r[k], r[k + 1]};
            acc[(k / 2) % NA] = __builtin_elementwise_fma(xm[k / 2], rv, acc[(k / 2) % NA]);
        }
#pragma unroll
        for (int i = 1; i < NA; ++i) acc[0] += acc[i];
        float d2 = acc[0].x + acc[0].y;
        if constexpr (D & 1) d2 = fmaf(xm_tail, r[D - 1], d2);
        if constexpr (KF == KF_POLY1) return fmaxf(d2, thr);
        else return d2;   // RQ2: t = d2 + 2/gamma (the seed rides in the row's last column), no clamp, no near pairs
    };
    auto coef_of = [&](const auto& r, float g) __attribute__((always_inline)) -> float {
        if constexpr (MODE == MODE_GRAD_ROW) {
            return g * r[CC > 1 ? L::WSUM_OFF : L::W_OFF];
        } else {
            float wb = 0.0f;
#pragma unroll
            for (int c = 0; c < CC; ++c) wb = fmaf(up[c], r[L::W_OFF + c], wb);
            return g * wb;
        }
    };
    // SIGN = +1: the hot path's expanded term of one row;  SIGN = -1 with `keep`: the same term taken out again for the
    // lanes in `keep` (others add -0)
    auto apply_x = [&](const auto& r, float d2c, auto sign, bool keep) __attribute__((always_inline)) {
        constexpr int SIGN = decltype(sign)::value;
        float val, g;
        sweep_eval<KF, true>(d2c, a, val, g);
        if constexpr (PARTS == 1) row_opaque(val, g);   // (two-buffer pipeline: this row's body must not pair up with the next row's)
        // one class, row weight: w r = (w / r) d2 — the score rides on the gradient coefficient (one multiply fewer)
        constexpr bool SCORE_BY_COEF = (KF == KF_POLY1 && CC == 1 && MODE == MODE_GRAD_ROW);
        if constexpr (!SCORE_BY_COEF) {
            if constexpr (SIGN < 0) val = keep ? -val : 0.0f;
            add_scores([&](int c) __attribute__((always_inline)) { return r[L::W_OFF + c]; }, val);
        }
        if constexpr (GRAD) {
            float coef = coef_of(r, g);
            if constexpr (PARTS == 1) asm volatile("" : "+v"(coef));
            if constexpr (SIGN < 0) coef = keep ? -coef : 0.0f;
            if constexpr (SCORE_BY_COEF) sc[0] = fmaf(coef, d2c, sc[0]);
            const v2f c2 = {coef, coef};
#pragma unroll
            for (int k = 0; k + 1 < D; k += 2) {
                const v2f rv = {r[k], r[k + 1]};
                ga[k / 2] = __builtin_elementwise_fma(c2, rv, ga[k / 2]);
            }
            if constexpr (D & 1) ga_tail = fmaf(coef, r[D - 1], ga_tail);
            asum += coef;
        }
    };
    auto pair_x = [&](const auto& r) __attribute__((always_inline)) -> float {
        const float d2 = d2_x(r);
        apply_x(r, d2, std::integral_constant<int, 1>{}, true);
        return d2;
    };
    // the rare block: row `r` with raw expanded distance d2raw has near lanes -> out with their expanded term, in with
    // the direct one
    auto fix_near = [&](const auto& r, float d2c) __attribute__((always_inline)) {
        const bool nr = d2c <= thr;
        apply_x(r, d2c, std::integral_constant<int, -1>{}, nr);
        // the differences are formed twice (once for the distance, once for the gradient term) rather than kept: this
        // block sets the kernel's peak register pressure, and at 64 VGPRs every register held here is a long-lived
        // value of the lone-wave code spilled to scratch
        const v2f mh = {-0.5f, -0.5f};  // x = -0.5 * (-2 x), exact: only the scaled copy stays in registers
        v2f dacc = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k + 1 < D; k += 2) {
            const v2f rv = {r[k], r[k + 1]};
            const v2f dk = xm[k / 2] * mh - rv;
            dacc = __builtin_elementwise_fma(dk, dk, dacc);
        }
        float d2d = dacc.x + dacc.y;
        if constexpr (D & 1) {
            const float dl_tail = -0.5f * xm_tail - r[D - 1];
            d2d = fmaf(dl_tail, dl_tail, d2d);
        }
        float val, g;
        kernel_eval<KF>(d2d, a, val, g);
        val = nr ? val : 0.0f;
        if constexpr (!NS) {
#pragma unroll
            for (int c = 0; c < CC; ++c) sc[c] = fmaf(r[L::W_OFF + c], val, sc[c]);
        }
        if constexpr (GRAD) {
            const float cd = nr ? -coef_of(r, g) : 0.0f;  // H carries MINUS the gradient
            const v2f cd2 = {cd, cd};
#pragma unroll
            for (int k = 0; k + 1 < D; k += 2) {
                const v2f rv = {r[k], r[k + 1]};
                ga[k / 2] = __builtin_elementwise_fma(cd2, xm[k / 2] * mh - rv, ga[k / 2]);
            }
            if constexpr (D & 1) ga_tail = fmaf(cd, -0.5f * xm_tail - r[D - 1], ga_tail);
        }
    };
    // two rows of a pipeline stage in the expanded form + the near check.  Each row's "any lane near?" goes to an SGPR
    // mask straight away (the distances are not kept: the rare block recomputes them, bit for bit), one scalar branch
    // per stage.
    auto stage_x = [&](const auto& r0, const auto& r1) __attribute__((always_inline)) {
        if constexpr (KF != KF_POLY1) {
            (void)pair_x(r0);
            (void)pair_x(r1);
            return;
        }
        const auto m0 = __builtin_amdgcn_ballot_w64(pair_x(r0) <= thr);
        const auto m1 = __builtin_amdgcn_ballot_w64(pair_x(r1) <= thr);
        if (__builtin_expect((m0 | m1) != 0, 0)) {
            fix_near(r0, d2_x(r0));
            fix_near(r1, d2_x(r1));
        }
    };
    auto single_x = [&](const auto& r0) __attribute__((always_inline)) {
        if constexpr (KF != KF_POLY1) {
            (void)pair_x(r0);
            return;
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(pair_x(r0) <= thr) != 0, 0)) fix_near(r0, d2_x(r0));
    };
    // fold the run's x * sum(c) into H in place: H <- H - x A = H + (-2 x) (A / 2), bit for bit the same product
    auto flush_x = [&]() __attribute__((always_inline)) {
        if constexpr (XFA && GRAD) {
            const v2f ah = {0.5f * asum, 0.5f * asum};
#pragma unroll
            for (int k = 0; k + 1 < D; k += 2) ga[k / 2] = __builtin_elementwise_fma(xm[k / 2], ah, ga[k / 2]);
            if constexpr (D & 1) ga_tail = fmaf(xm_tail, 0.5f * asum, ga_tail);
            asum = 0.0f;
        }
    };
    // one support row against this lane's configuration; `r` is wave-uniform (SGPRs)
    auto pair = [&](const float (&r)[L::RS]) __attribute__((always_inline)) {
        float dl[D];
        float d2;
        v2f dp[D / 2 + 1];
        {
            // NA independent accumulator pairs: a wide row's D/2 dependent v_pk_fma would otherwise be one serial
            // chain, and wide shapes run at 2-4 waves per SIMD, too few to hide it
            constexpr int NA = DCX_D2_ACCS(D);
            v2f acc[NA];
            acc[0] = v2f{d2_seed<KF>(a), 0.0f};
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
            d2 = acc[0].x + acc[0].y;
            if constexpr (D & 1) {
                dl[D - 1] = x[D - 1] - r[D - 1];
                d2 = fmaf(dl[D - 1], dl[D - 1], d2);
            }
        }
        float val, g;
        sweep_eval<KF>(d2, a, val, g);
        if constexpr (!NS) {
#pragma unroll
            for (int c = 0; c < CC; ++c) sc[c] = fmaf(r[L::W_OFF + c], val, sc[c]);
        }
        if constexpr (GRAD) {
            float coef;
            if constexpr (MODE == MODE_GRAD_ROW) {
                coef = g * r[CC > 1 ? L::WSUM_OFF : L::W_OFF];
            } else {
                float wb = 0.0f;
#pragma unroll
                for (int c = 0; c < CC; ++c) wb = fmaf(up[c], r[L::W_OFF + c], wb);
                coef = g * wb;
            }
            const v2f c2 = {coef, coef};
#pragma unroll
            for (int k = 0; k + 1 < D; k += 2) gx2[k / 2] = __builtin_elementwise_fma(c2, dp[k / 2], gx2[k / 2]);
            if constexpr (D & 1) gx[D - 1] = fmaf(coef, dl[D - 1], gx[D - 1]);
        }
    };
    // ---- two rows per packed instruction (round 5) -------------------------------------------------------------------------
    // `pair` packs a row's features two by two: D/2 differences, D/2 squared-distance terms, one add of the two halves, then a
    // scalar chain (kernel function, score, coefficient) and D/2 gradient terms.  For NARROW rows the chain is a third of the
    // body (config #4, D = 6: 15 VALU instructions per pair, 6 of them the chain + the add of the halves).  Here the two rows
    // of a pipeline stage ride in the two halves of every packed register instead: feature k of both rows is one operand pair
    // (x_k in both halves - D register pairs made once per sweep -, (r0_k, r1_k) an SGPR pair built by two s_mov), the squared
    // distances of the two rows come out of D packed
    // fmas with no add of halves, the chain runs once for both rows in packed multiplies (the reciprocal / rsqrt stay one per
    // row: no packed form exists), and the gradient accumulators hold the even and the odd rows' sums side by side (2 D
    // registers instead of D: why this is for D <= 8 only).  24 instead of 30 VALU instructions per two rows at D = 6.
    constexpr bool P2 = !XF && p2_applies(D, CC, KF);
    static_assert(!P2 || PARTS == 0, "pair2: whole rows in the four-row budget");
    v2f gp[P2 ? D : 1];     // gradient, rows of even / odd position in the stage
    v2f scp = {0.0f, 0.0f};
    if constexpr (P2) {
#pragma unroll
        for (int k = 0; k < D; ++k) gp[k] = v2f{0.0f, 0.0f};
    }
    // b: the stage's 2 RS floats, pair-interleaved (element e of the even row at b[2 e], of the odd row at b[2 e + 1])
    auto pair2 = [&](const float (&b)[2 * L::RS]) __attribute__((always_inline)) {
        v2f dk[D];
        v2f acc = {d2_seed<KF>(a), d2_seed<KF>(a)};
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const v2f xv = {x[k], x[k]};
            const v2f rv = {b[2 * k], b[2 * k + 1]};
            dk[k] = xv - rv;
            acc = __builtin_elementwise_fma(dk[k], dk[k], acc);
        }
        v2f val, g;
        if constexpr (KF == KF_RQ2) {   // sweep_eval<KF_RQ2> on both halves
            const v2f u = {__builtin_amdgcn_rcpf(acc.x), __builtin_amdgcn_rcpf(acc.y)};
            val = u * u;
            g = val * u;
        } else {                        // kernel_eval<KF_POLY1>
            const v2f d2c = {fmaxf(acc.x, 1e-30f), fmaxf(acc.y, 1e-30f)};
            g = v2f{__builtin_amdgcn_rsqf(d2c.x), __builtin_amdgcn_rsqf(d2c.y)};
            val = d2c * g;
        }
        const v2f w2 = {b[2 * L::W_OFF], b[2 * L::W_OFF + 1]};
        scp = __builtin_elementwise_fma(w2, val, scp);
        if constexpr (GRAD) {
            v2f coef = g * w2;
            if constexpr (MODE != MODE_GRAD_ROW) coef = coef * v2f{up[0], up[0]};
#pragma unroll
            for (int k = 0; k < D; ++k) gp[k] = __builtin_elementwise_fma(coef, dk[k], gp[k]);
        }
    };
    // (whole groups of four floats: the padding of the row stride is readable, and 7 floats fetched as 8 are ONE s_load_dwordx8
    // instead of x4 + x2 + x1, 14 as 16 one x16 instead of x8 + x4 + x2)
    constexpr int USED4 = (USED + 3) / 4 * 4;
    static_assert(USED4 <= L::RS, "the row stride covers whole groups of four");
    constexpr int NLOAD = DCX_LOAD_GROUPS ? USED4 : USED;
#ifdef DCX_EXP_NO_LOADS   // timing experiment only (wrong results): after the first few, a row "load" fetches nothing - the
    int exp_real_loads = 4;  // registers keep their values behind an opaque barrier - what do the scalar loads cost the stream?
#endif
    auto load_row = [&](float (&dst)[L::RS], int j) __attribute__((always_inline)) {
        cfloat_ptr r = rows + (size_t)j * RSTRIDE;
#ifdef DCX_EXP_NO_LOADS
        if (exp_real_loads <= 0) {
#pragma unroll
            for (int e = 0; e < NLOAD; ++e) asm volatile("" : "+s"(dst[e]));
            return;
        }
        --exp_real_loads;
#endif
#pragma unroll
        for (int e = 0; e < NLOAD; ++e) dst[e] = r[e];
    };

    // How many SGPRs a pipeline may keep in flight: ~100 exist, the kernel needs a dozen for itself.  Four whole rows
    // (the deepest pipeline) fit up to 22 floats per row; wider rows run a two-buffer pipeline over whole rows
    // (<= 38 floats), over half rows, or over thirds of a row (D > 75).  Before this split the compiler kept the four-row pipeline
    // alive for every width by parking SGPRs in VGPR lanes: D=24 +37 %, D=42 +75 %, D=60 +97 % VALU instructions
    // (v_writelane / v_readlane) inside the sweep.
    static_assert(!XFA || PARTS <= 1, "expanded form: whole rows only");
    constexpr bool X2 = XFA && !XM && x2_applies(D, CC, KF);
    if constexpr (X2) {
    // ---- X2 (round 6): the expanded form, two rows per packed instruction ----------------------------------------------------------
    // The expanded body above packs a row's features two by two: D/2 v_pk_fma for the distance and D/2 for the gradient, an add of
    // the two halves, and a scalar chain per row (seed, clamp, rsq, coefficient, score, sum of coefficients, near test): 20 VALU
    // instructions per row at D = 12.  Here the two rows of a stage ride in the two halves of every packed register instead (the
    // centred rows are stored pair-interleaved: element e of the even row at b[2 e], of the odd row at b[2 e + 1], so every operand
    // pair is an aligned SGPR pair of the stage's one load): D + D v_pk_fma per TWO rows, no add of halves, the chain's adds and
    // multiplies packed (clamp, rsq and near test stay one per row: no packed form exists) - 34 per two rows at D = 12 - and the
    // expanded gradient H in 2 D accumulators (even rows' sums in .x, odd rows' in .y; one add per feature at the end).  The bare
    // body (tools/sweep_body_ubench.hip): 39.3 ns per wave-row per SIMD against 44.2 at 4 waves per SIMD, 40.0 against 41.1 at 8.
    // A row's distance is ONE fma chain over its features here (xx + ss first), and the rare near-pair block below reproduces
    // exactly that chain in scalar fma before it takes the expanded term out and puts the direct one in (the same semantics as
    // fix_near above).  j0 is even (the host's slicing); a slice that ends on the model's odd last row runs into the zero-weight
    // padding row (d2 = |x - c|^2 there: finite, coefficient 0).
    if (j0 < j1) {
        constexpr int R2 = 2 * L::RS;
        v2f h2[GRAD ? D : 1], sc2x = {0.0f, 0.0f}, as2 = {0.0f, 0.0f};
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) h2[k] = v2f{0.0f, 0.0f};
        }
        float xmk[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xmk[k] = -2.0f * x[k];
        float upc = 1.0f;
        if constexpr (MODE == MODE_GRAD_UP) upc = up[0];
        auto load2 = [&](float (&dst)[R2], int j) __attribute__((always_inline)) {
            cfloat_ptr r = rows + (size_t)j * RSTRIDE;
#pragma unroll
            for (int e = 0; e < R2; ++e) dst[e] = r[e];
        };
        // the rare block, one row (half HF of the stage): out with the expanded term of the near lanes, in with the direct one
        auto fix2 = [&](const float (&b)[R2], auto hf) __attribute__((always_inline)) {
            constexpr int HF = decltype(hf)::value;
            float d2 = xx + b[2 * L::SS_OFF + HF];
#pragma unroll
            for (int k = 0; k < D; ++k) d2 = fmaf(xmk[k], b[2 * k + HF], d2);
            const float d2c = fmaxf(d2, thr);
            const bool nr = d2c <= thr;
            float val, g;
            sweep_eval<KF, true>(d2c, a, val, g);
            const float w = b[2 * L::W_OFF + HF];
            float coef = g * w;
            if constexpr (MODE == MODE_GRAD_UP) coef *= upc;
            const float cneg = nr ? -coef : 0.0f;
            // this row's half of a packed accumulator: v <- fma(p, q, v)
            auto hfma = [&](v2f& v, float p_, float q_) __attribute__((always_inline)) {
                if constexpr (HF) v.y = fmaf(p_, q_, v.y);
                else v.x = fmaf(p_, q_, v.x);
            };
            if constexpr (KF == KF_POLY1 && MODE == MODE_GRAD_ROW) hfma(sc2x, cneg, d2c);
            else hfma(sc2x, nr ? -w : 0.0f, val);
            float dk[D];
            float da = 0.0f, db = 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                dk[k] = -0.5f * xmk[k] - b[2 * k + HF];     // x = -0.5 * (-2 x), exact
                if (k & 1) db = fmaf(dk[k], dk[k], db);
                else da = fmaf(dk[k], dk[k], da);
            }
            float vald, gd;
            kernel_eval<KF>(da + db, a, vald, gd);          // (the direct sweep's distance: even / odd features, then the two halves)
            hfma(sc2x, w, nr ? vald : 0.0f);
            if constexpr (GRAD) {
                float cdir = gd * w;
                if constexpr (MODE == MODE_GRAD_UP) cdir *= upc;
                const float cd = nr ? -cdir : 0.0f;          // H carries MINUS the gradient
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    hfma(h2[k], cneg, b[2 * k + HF]);
                    hfma(h2[k], cd, dk[k]);
                }
                if constexpr (HF) as2.y += cneg;
                else as2.x += cneg;
            }
        };
        auto body2 = [&](const float (&b)[R2]) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 2 * USED; e < R2; ++e) asm volatile("" ::"s"(b[e]));
            v2f acc = v2f{xx, xx} + v2f{b[2 * L::SS_OFF], b[2 * L::SS_OFF + 1]};
#pragma unroll
            for (int k = 0; k < D; ++k) acc = __builtin_elementwise_fma(v2f{xmk[k], xmk[k]}, v2f{b[2 * k], b[2 * k + 1]}, acc);
            v2f d2c = acc, val, g;
            if constexpr (KF == KF_POLY1) d2c = v2f{fmaxf(acc.x, thr), fmaxf(acc.y, thr)};
            {
                float v0, g0, v1, g1;
                sweep_eval<KF, true>(d2c.x, a, v0, g0);
                sweep_eval<KF, true>(d2c.y, a, v1, g1);
                val = v2f{v0, v1};
                g = v2f{g0, g1};
            }
            const v2f w2 = {b[2 * L::W_OFF], b[2 * L::W_OFF + 1]};
            v2f coef = g * w2;
            if constexpr (MODE == MODE_GRAD_UP) coef = coef * v2f{upc, upc};
            // one class, row weight, Polyharmonic(1): w r = (w / r) d2 - the score rides on the gradient coefficient
            if constexpr (KF == KF_POLY1 && MODE == MODE_GRAD_ROW) sc2x = __builtin_elementwise_fma(coef, d2c, sc2x);
            else sc2x = __builtin_elementwise_fma(w2, val, sc2x);
            if constexpr (GRAD) {
#pragma unroll
                for (int k = 0; k < D; ++k) h2[k] = __builtin_elementwise_fma(coef, v2f{b[2 * k], b[2 * k + 1]}, h2[k]);
                as2 += coef;
            }
            if constexpr (KF == KF_POLY1) {
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(fminf(d2c.x, d2c.y) <= thr) != 0, 0)) {
                    fix2(b, std::integral_constant<int, 0>{});
                    fix2(b, std::integral_constant<int, 1>{});
                }
            }
        };
        // fold the run's x * sum(c) into H in place: H <- H + (-2 x) (A / 2), per half
        auto flush2 = [&]() __attribute__((always_inline)) {
            if constexpr (GRAD) {
                const v2f ah = {0.5f * as2.x, 0.5f * as2.y};
#pragma unroll
                for (int k = 0; k < D; ++k) h2[k] = __builtin_elementwise_fma(v2f{xmk[k], xmk[k]}, ah, h2[k]);
                as2 = v2f{0.0f, 0.0f};
            }
        };
        float ab[R2], cd[R2];
        const int j1e = (j1 + 1) & ~1;
        const int jl = j1e - 2;
        load2(ab, j0);
        int j = j0;
        for (; j + 3 < j1e; j += 4) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load2(cd, j + 2);
            __builtin_amdgcn_sched_barrier(0);
            body2(ab);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load2(ab, (j + 4 < j1e) ? j + 4 : jl);
            __builtin_amdgcn_sched_barrier(0);
            body2(cd);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (GRAD) {
                if ((j & (DCX_XF_FLUSH - 4)) == 0) flush2();  // once per DCX_XF_FLUSH rows, whatever j0's alignment
            }
        }
        if (j < j1e) body2(ab);   // one stage left; ab holds it
        flush2();
        sc[0] += sc2x.x + sc2x.y;
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) gx[k] -= h2[k].x + h2[k].y;
        }
    }
    } else if constexpr (XM) {
    // ---- XM: the expanded form with x . s^T on the matrix cores (round 3) ---------------------------------------------------
    // d2 = (|x|^2 + |s_j|^2) + sum_k (-2 x_k) s_jk: the one contraction of the sweep whose per-lane operand is loop
    // invariant.  -2x is split ONCE per lane into three bf16 planes (hi, mid, lo by truncation: 24 bits) and turned into
    // the B fragments of the four 16-configuration tiles with the 4 x 4 lane transpose; the supports were split on the
    // host and laid out as A operands (dcx_model_create, `aplanes`).  The six plane products that matter (hi.hi, hi.mid,
    // mid.hi, hi.lo, lo.hi, mid.mid) sit side by side along K (6 terms x 16 slots = three v_mfma_f32_16x16x32_bf16 per
    // tile), accumulated in fp32 inside the instruction: the sum carries the 2^-24 (|x| |s|) error of the fp32 expanded
    // form.  Per 16 supports and wave: 12 MFMAs + 16 lane swaps give every lane its 16 dot products; the rest of the
    // pair body (clamp, 1 / r, score, fold, near pairs) is the XF one, two rows per scalar-load stage.  Measured in
    // isolation (tools/contraction_ubench.hip, "pair body"): 82-86 cycles per wave-row against 96-98.
    static_assert(XFA && CC == 1 && D <= 16 && (D % 2) == 0, "XM: one class, even D <= 16, expanded form");
    if (j0 < j1) {
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        const int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        auto frags = [&](unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int (&f)[4]) __attribute__((always_inline)) {
            auto s02 = __builtin_amdgcn_permlane32_swap(c0, c2, false, false);
            auto s13 = __builtin_amdgcn_permlane32_swap(c1, c3, false, false);
            auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
            auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
            f[0] = t01[0]; f[1] = t01[1]; f[2] = t23[0]; f[3] = t23[1];
        };
        v4u bfrag[4][3];
        {
            unsigned int pk[3][8];  // [plane][feature pair], features past D are zero
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float r0 = (2 * p < D) ? -2.0f * x[2 * p < D ? 2 * p : 0] : 0.0f;
                float r1 = (2 * p + 1 < D) ? -2.0f * x[2 * p + 1 < D ? 2 * p + 1 : 0] : 0.0f;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned int u0 = __float_as_uint(r0) & 0xFFFF0000u, u1 = __float_as_uint(r1) & 0xFFFF0000u;
                    pk[pl][p] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
                    r0 -= __uint_as_float(u0);
                    r1 -= __uint_as_float(u1);
                }
            }
            constexpr int xplane_of_term[6] = {0, 0, 1, 0, 2, 1};  // terms: hi.hi hi.mid mid.hi hi.lo lo.hi mid.mid (x plane)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // lane (n, k'): k' = 0, 1 -> term 2c, features 8 k' + 2e, + 1;  k' = 2, 3 -> term 2c + 1
                    unsigned int f[4];
                    frags(pk[xplane_of_term[2 * c]][e], pk[xplane_of_term[2 * c]][4 + e], pk[xplane_of_term[2 * c + 1]][e],
                          pk[xplane_of_term[2 * c + 1]][4 + e], f);
#pragma unroll
                    for (int t = 0; t < 4; ++t) bfrag[t][c][e] = f[t];
                }
        }
        // one row with its distance term from the matrix cores; returns the clamped distance
        auto pair_m = [&](const auto& r, float dot) __attribute__((always_inline)) -> float {
            const float d2 = fmaxf((xx + r[L::SS_OFF]) + dot, thr);
            apply_x(r, d2, std::integral_constant<int, 1>{}, true);
            return d2;
        };
        auto stage_m = [&](const auto& r0, const auto& r1, float dt0, float dt1) __attribute__((always_inline)) {
            const float d0 = pair_m(r0, dt0), d1 = pair_m(r1, dt1);
            const auto m0 = __builtin_amdgcn_ballot_w64(d0 <= thr);
            const auto m1 = __builtin_amdgcn_ballot_w64(d1 <= thr);
            if (__builtin_expect((m0 | m1) != 0, 0)) {
                fix_near(r0, d0);
                fix_near(r1, d1);
            }
        };
        float rowA[L::RS], rowB[L::RS], rowC[L::RS], rowD[L::RS];
        const int jl = j1 - 1;
        int j = j0;
        // rows in front of the first 16-row block of the A planes (slices start on a block boundary except where a caller's
        // slicing does not: then at most 15 rows): the XF body
        for (; j < j1 && (j & 15) != 0; ++j) {
            load_row(rowA, j);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            single_x(rowA);
        }
        if (j + 16 <= j1) {
            // A operand of a 16-row block: lane (m, k') reads 16 bytes of [block][chunk][support m][k']
            const v4u* ap = reinterpret_cast<const v4u*>(a.aplanes) + (lane_ & 15) * 4 + (lane_ >> 4);
            v4u a0 = ap[(size_t)(j >> 4) * 192], a1 = ap[(size_t)(j >> 4) * 192 + 64], a2 = ap[(size_t)(j >> 4) * 192 + 128];
            load_row(rowA, j);
            load_row(rowB, j + 1);
            for (; j + 16 <= j1; j += 16) {
                const size_t nb = (size_t)((j >> 4) + 1) * 192;   // (the planes are padded by two blocks)
                const v4u n0 = ap[nb], n1 = ap[nb + 64], n2 = ap[nb + 128];
                float dot[16];
                {
                    v4f_ d[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        v4f_ c = {0.f, 0.f, 0.f, 0.f};  // small terms first
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a2), __builtin_bit_cast(v8bf, bfrag[t][2]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a1), __builtin_bit_cast(v8bf, bfrag[t][1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a0), __builtin_bit_cast(v8bf, bfrag[t][0]), c, 0, 0, 0);
                        d[t] = c;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        unsigned int f[4];
                        frags(__float_as_uint(d[0][i]), __float_as_uint(d[1][i]), __float_as_uint(d[2][i]), __float_as_uint(d[3][i]), f);
#pragma unroll
                        for (int g = 0; g < 4; ++g) dot[4 * g + i] = __uint_as_float(f[g]);
                    }
                }
                a0 = n0; a1 = n1; a2 = n2;
#pragma unroll
                for (int q = 0; q < 16; q += 4) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);
                    load_row(rowC, j + q + 2);
                    load_row(rowD, j + q + 3);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_m(rowA, rowB, dot[q], dot[q + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);
                    load_row(rowA, (j + q + 4 < j1) ? j + q + 4 : jl);
                    load_row(rowB, (j + q + 5 < j1) ? j + q + 5 : jl);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_m(rowC, rowD, dot[q + 2], dot[q + 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (GRAD) {
                    if ((j & 48) == 0) flush_x();  // once per 64 rows
                }
            }
        }
        // what is left of the slice (< 16 rows): the XF body
        for (; j < j1; ++j) {
            load_row(rowA, j);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            single_x(rowA);
        }
    }
    } else if constexpr (P2 && DCX_PAIR2_LOADS) {
    // pair2's own pipeline: the two rows of a stage are ONE piece of memory (pair-interleaved, see above), so they arrive by one
    // scalar load of 2 RS floats (one address computation per stage instead of one per row: with 12 VALU instructions per pair the
    // scalar unit, which the four SIMDs of a CU share, had become nearly as busy as the vector one).  j0 is even (the host's
    // slicing); a slice that ends on an odd row - the model's last - takes the zero-weight padding row with it.  The rows have a
    // readable tail (rows_tail_floats), so the look-ahead load at the end of a slice may run one stage past it.
    //   wait -> issue {C,D} -> body(A,B) -> wait -> issue {A,B} -> body(C,D)
    if (j0 < j1) {
        constexpr int R2 = 2 * L::RS;
        float ab[R2], cd[R2];
        auto load2 = [&](float (&dst)[R2], int j) __attribute__((always_inline)) {
            cfloat_ptr r = rows + (size_t)j * RSTRIDE;
#pragma unroll
            for (int e = 0; e < R2; ++e) dst[e] = r[e];
        };
        auto body2 = [&](const float (&b)[R2]) __attribute__((always_inline)) {
            // (the stride's padding counts as used where the rows are consumed: the compiler narrows a load whose tail is dead
            // into x4 + x2 + x1 pieces otherwise)
#pragma unroll
            for (int e = 2 * USED; e < R2; ++e) asm volatile("" ::"s"(b[e]));
            pair2(b);
        };
        const int j1e = (j1 + 1) & ~1;
        const int jl = j1e - 2;
        load2(ab, j0);
        int j = j0;
        for (; j + 3 < j1e; j += 4) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load2(cd, j + 2);
            __builtin_amdgcn_sched_barrier(0);
            body2(ab);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load2(ab, (j + 4 < j1e) ? j + 4 : jl);
            __builtin_amdgcn_sched_barrier(0);
            body2(cd);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j < j1e) body2(ab);   // one stage left; ab holds it
    }
    } else if constexpr (PARTS == 0) {
    // Explicit software pipeline, two rows per stage (4 row buffers): the wait before a stage covers loads
    // issued TWO row bodies earlier, which is what hides an L2-latency scalar miss when only a few waves
    // share a SIMD (small batches).  wait -> issue {C,D} -> body(A), body(B) -> wait -> issue {A,B} -> body(C), body(D)
    if (j0 < j1) {
        float rowA[L::RS], rowB[L::RS], rowC[L::RS], rowD[L::RS];
        const int jl = j1 - 1;  // clamp target for the look-ahead loads (harmless re-reads at the end)
        load_row(rowA, j0);
        load_row(rowB, (j0 + 1 < j1) ? j0 + 1 : jl);
        int j = j0;
        for (; j + 3 < j1; j += 4) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load_row(rowC, j + 2);
            load_row(rowD, j + 3);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (XFA) {
                stage_x(rowA, rowB);
            } else {
                pair(rowA);
                pair(rowB);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load_row(rowA, (j + 4 < j1) ? j + 4 : jl);
            load_row(rowB, (j + 5 < j1) ? j + 5 : jl);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (XFA) {
                stage_x(rowC, rowD);
            } else {
                pair(rowC);
                pair(rowD);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (XFA && GRAD) {
                if ((j & (DCX_XF_FLUSH - 4)) == 0) flush_x();  // once per DCX_XF_FLUSH rows, whatever j0's alignment
            }
        }
        // up to three rows left: rowA / rowB already hold rows j and j+1
        auto one = [&](const float (&r)[L::RS]) __attribute__((always_inline)) {
            if constexpr (XFA) single_x(r);
            else pair(r);
        };
        if (j < j1) one(rowA);
        if (j + 1 < j1) one(rowB);
        if (j + 2 < j1) {
            load_row(rowC, j + 2);
            one(rowC);
        }
    }
    } else {
        constexpr int PS = ((USED + PARTS - 1) / PARTS + 3) / 4 * 4;  // floats per part (pairs never straddle parts)
        constexpr int NA = DCX_D2_ACCS(D);
        float bufA[PS], bufB[PS];
        // state of the row being consumed
        v2f acc[NA];
        v2f dp[D / 2 + 1];
        float tail_d = 0.0f;   // odd D: the unpaired last feature's difference
        float wv[CC];
        float wsum = 0.0f;
        auto load_part = [&](float (&dst)[PS], int j, auto pc) __attribute__((always_inline)) {
            constexpr int P0 = decltype(pc)::value * PS;
            constexpr int LEN = (PARTS == 1 && DCX_LOAD_GROUPS) ? PS : (USED - P0 < PS) ? (USED - P0) : PS;
            cfloat_ptr r = rows + (size_t)j * RSTRIDE + P0;
#ifdef DCX_EXP_NO_LOADS
            if (exp_real_loads <= 2) {
#pragma unroll
                for (int e = 0; e < LEN; ++e) asm volatile("" : "+s"(dst[e]));
                return;
            }
            --exp_real_loads;
#endif
#if defined(DCX_EXP_LOAD_DWORDS)   // timing experiment only (wrong results): fetch this many dwords per row, whatever its width
            static_assert(PARTS != 1 || DCX_EXP_LOAD_DWORDS <= 16 || RSTRIDE >= 32, "build with DCX_ROW_ALIGN_MULTI=32");
            if constexpr (PARTS == 1) {
                float extra = 0.0f;
#pragma unroll
                for (int e = 0; e < (DCX_EXP_LOAD_DWORDS < LEN ? DCX_EXP_LOAD_DWORDS : LEN); ++e) dst[e] = r[e];
#pragma unroll
                for (int e = DCX_EXP_LOAD_DWORDS; e < LEN; ++e) dst[e] = dst[e - DCX_EXP_LOAD_DWORDS];
#pragma unroll
                for (int e = LEN; e < DCX_EXP_LOAD_DWORDS; ++e) extra += r[e];   // dwords beyond the row: fetched and folded into one operand
                if constexpr (DCX_EXP_LOAD_DWORDS > LEN) dst[LEN - 1] += 1e-30f * extra;
                return;
            }
#endif
#pragma unroll
            for (int e = 0; e < LEN; ++e) dst[e] = r[e];
        };
        auto consume = [&](const float (&b)[PS], auto pc) __attribute__((always_inline)) {
            if constexpr (XFA) {  // whole rows (PARTS == 1)
                single_x(b);
                return;
            }
            constexpr int P0 = decltype(pc)::value * PS;
            constexpr int LEN = (USED - P0 < PS) ? (USED - P0) : PS;
            if constexpr (decltype(pc)::value == 0) {
                acc[0] = v2f{d2_seed<KF>(a), 0.0f};
#pragma unroll
                for (int i = 1; i < NA; ++i) acc[i] = v2f{0.0f, 0.0f};
            }
#pragma unroll
            for (int e = 0; e < LEN; ++e) {
                const int g = P0 + e;  // compile-time after unrolling
                if (g + 1 < D && (g & 1) == 0) {
                    const v2f xv = {x[g], x[g + 1]};
                    const v2f rv = {b[e], b[e + 1]};
                    dp[g / 2] = xv - rv;
                    acc[(g / 2) % NA] = __builtin_elementwise_fma(dp[g / 2], dp[g / 2], acc[(g / 2) % NA]);
                } else if (g == D - 1 && (D & 1)) {
                    tail_d = x[g] - b[e];
                } else if (g >= D && g < D + CC) {
                    wv[g - D] = b[e];
                } else if (CC > 1 && g == D + CC) {
                    wsum = b[e];
                }
            }
            if constexpr (decltype(pc)::value == PARTS - 1) {  // the row is complete
#pragma unroll
                for (int i = 1; i < NA; ++i) acc[0] += acc[i];
                float d2 = acc[0].x + acc[0].y;
                if constexpr (D & 1) d2 = fmaf(tail_d, tail_d, d2);
                float val, g;
                sweep_eval<KF>(d2, a, val, g);
                if constexpr (PARTS == 1) row_opaque(val, g);
                add_scores([&](int c) __attribute__((always_inline)) { return wv[c]; }, val);
                if constexpr (GRAD) {
                    float coef;
                    if constexpr (MODE == MODE_GRAD_ROW) {
                        coef = g * (CC > 1 ? wsum : wv[0]);
                    } else {
                        float wb = 0.0f;
#pragma unroll
                        for (int c = 0; c < CC; ++c) wb = fmaf(up[c], wv[c], wb);
                        coef = g * wb;
                    }
                    if constexpr (PARTS == 1) asm volatile("" : "+v"(coef));
                    const v2f c2 = {coef, coef};
#pragma unroll
                    for (int k = 0; k + 1 < D; k += 2) gx2[k / 2] = __builtin_elementwise_fma(c2, dp[k / 2], gx2[k / 2]);
                    if constexpr (D & 1) gx[D - 1] = fmaf(coef, tail_d, gx[D - 1]);
                }
            }
        };
        using P0c = std::integral_constant<int, 0>;
        using P1c = std::integral_constant<int, PARTS - 1>;
        if (j0 < j1) {
            const int jl = j1 - 1;
            if constexpr (PARTS == 1) {
                // two whole-row buffers: wait -> issue B -> consume A -> wait -> issue A -> consume B
                static_assert(PS <= RSTRIDE, "a whole row in groups of four floats stays inside its stride");
                load_part(bufA, j0, P0c{});
                int j = j0;
                for (; j + 1 < j1; j += 2) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);
                    load_part(bufB, j + 1, P0c{});
                    __builtin_amdgcn_sched_barrier(0);
                    consume(bufA, P0c{});
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);
                    load_part(bufA, (j + 2 < j1) ? j + 2 : jl, P0c{});
                    __builtin_amdgcn_sched_barrier(0);
                    consume(bufB, P0c{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (XFA && GRAD) {
                        if ((j & (DCX_XF_FLUSH - 2)) == 0) flush_x();
                    }
                }
                if (j < j1) consume(bufA, P0c{});  // bufA holds row j
            } else if constexpr (PARTS == 2) {
                // two half-row buffers: wait -> issue second half -> consume first half -> wait -> issue the next row's
                // first half -> consume second half (row complete)
                load_part(bufA, j0, P0c{});
                for (int j = j0; j < j1; ++j) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);
                    load_part(bufB, j, P1c{});
                    __builtin_amdgcn_sched_barrier(0);
                    consume(bufA, P0c{});
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);
                    load_part(bufA, (j + 1 < j1) ? j + 1 : jl, P0c{});
                    __builtin_amdgcn_sched_barrier(0);
                    consume(bufB, P1c{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // three or more parts per row (D > 75): the same two buffers, alternating over the part stream; with an
                // odd part count a row boundary flips the buffer parity, so the loop body covers two rows
                auto step = [&](auto self, auto uc, int j) __attribute__((always_inline)) -> void {
                    constexpr int U = decltype(uc)::value;          // position in the two-row part stream
                    constexpr int NU = U + 1;                        // the part requested now
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const int jr = j + NU / PARTS;
                        const int row = jr < j1 ? jr : jl;
                        if constexpr (NU % 2 == 0) load_part(bufA, row, std::integral_constant<int, NU % PARTS>{});
                        else load_part(bufB, row, std::integral_constant<int, NU % PARTS>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (U % 2 == 0) consume(bufA, std::integral_constant<int, U % PARTS>{});
                    else consume(bufB, std::integral_constant<int, U % PARTS>{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (U + 1 < 2 * PARTS) self(self, std::integral_constant<int, U + 1>{}, j);
                };
                load_part(bufA, j0, P0c{});
                int j = j0;
                for (; j + 1 < j1; j += 2) step(step, P0c{}, j);
                if (j < j1) {  // one row left; bufA holds its first part (2 * PARTS steps keep the parity)
                    auto tail = [&](auto self, auto pc) __attribute__((always_inline)) -> void {
                        constexpr int P = decltype(pc)::value;
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (P + 1 < PARTS) {
                            if constexpr ((P + 1) % 2 == 0) load_part(bufA, j, std::integral_constant<int, P + 1>{});
                            else load_part(bufB, j, std::integral_constant<int, P + 1>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (P % 2 == 0) consume(bufA, pc);
                        else consume(bufB, pc);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (P + 1 < PARTS) self(self, std::integral_constant<int, P + 1>{});
                    };
                    tail(tail, P0c{});
                }
            }
        }
    }

    if constexpr (X2) {
        // (flushed and merged inside its own block above)
    } else if constexpr (XFA && GRAD) {
        flush_x();
#pragma unroll
        for (int k = 0; k + 1 < D; k += 2) {
            gx[k] -= ga[k / 2].x;
            gx[k + 1] -= ga[k / 2].y;
        }
        if constexpr (D & 1) gx[D - 1] -= ga_tail;
    } else {
#pragma unroll
        for (int k = 0; k + 1 < D; k += 2) {
            gx[k] += gx2[k / 2].x;
            gx[k + 1] += gx2[k / 2].y;
        }
        if constexpr (P2) {   // the stages' even and odd rows, then the rows the tail took one by one (above)
            sc[0] += scp.x + scp.y;
            if constexpr (GRAD) {
#pragma unroll
                for (int k = 0; k < D; ++k) gx[k] += gp[k].x + gp[k].y;
            }
        }
    }
    if constexpr (SC2) {
#pragma unroll
        for (int c = 0; c + 1 < CC; c += 2) {
            sc[c] += sc2[c / 2].x;
            sc[c + 1] += sc2[c / 2].y;
        }
    }
    if constexpr (GRAD && kGradScale<KF> != 1.0f) {   // the folded constant of the gradient (sweep_eval), once per lane
#pragma unroll
        for (int k = 0; k < D; ++k) gx[k] *= kGradScale<KF>;
    }
}

// ---- the 16-configuration tile (QT, round 4): rows from LDS, lane = (configuration, row slice) -----------------------------
// a batch with fewer 64-configuration tiles than CUs splits the supports over several blocks per tile and
// pays a cross-block hand-over (4.5 k of a config-#2 block's 24 k cycles even with the owner polling).  Here a block takes 16
// configurations and ALL the rows: 4096 configurations = 256 blocks, one per CU, no arrival counter, no second FK, the
// partial sums meet inside the block (the usual fold over the waves, then two lane exchanges over the four slices of a wave).
// The row operands cannot be wave-uniform any more (a wave sweeps four rows at once to have 64 lanes of work), so they come
// from an LDS copy of the rows: three ds_read_b128 + one ds_read_b32 per pair and lane, each row broadcast to its 16 lanes.
// The pair body is the DIRECT form (differences; the same operations in the same order as sweep_rows' `pair`), on the
// model's own rows: no centred data, no near-pair block.
// What the stamps of a config-#2 block say (profiles/r04_qt.txt): the hand-over's 4.2 k cycles are gone, the copy of the rows
// adds 1.4 k (45 % of it rides with the q rows, the rest is done by the idle waves during the FK chain), the sweep is 9.2 k
// against 8.6 k - it is bound by the four waves a SIMD has to cover the LDS latency and the body's dependent chains, not by
// the LDS pipe (6.7 k) or the VALU count: the EXPANDED body (17 instead of 24 VALU instructions, one more ds_read_b32 and a
// ballot per pair) came out SLOWER, 10.6 k, and two configurations per lane (half the LDS reads per pair) 10.4 k.  Net:
// config #2 11.2 -> 10.5 us.
// One class, row weights (MODE_GRAD_ROW) or the score alone (MODE_SCORE), the two specialised kernel functions, D = 12 / 24; the host takes it for batches of
// at most 16 configurations per CU when the rows fit the LDS (dcx_api.hip run_score).
constexpr int kQtSlices = 4;   // row slices per wave
constexpr bool qt_applies(int D, int CC, int KF, int MODE) {
    return (D == 12 || D == 24) && CC == 1 && (MODE == 1 /* MODE_GRAD_ROW */ || MODE == 0 /* MODE_SCORE */) && KF != 2 /* KF_GEN */;
}
template <int D, int KF, bool GRAD>
__device__ __forceinline__ void sweep_rows_lds(const ScoreArgs& a, const float (&x)[D], const float* slice, int per, float& sc0,
                                               float (&gx)[D]) {
    static_assert(D % 4 == 0, "rows are read as whole float4s");
    using L = RowLayout<D, 1>;
    v2f g2[D / 2];
#pragma unroll
    for (int k = 0; k < D / 2; ++k) g2[k] = v2f{0.0f, 0.0f};
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    auto load_row = [&](float (&r)[D + 1], int jj) __attribute__((always_inline)) {
        const float* p = slice + (size_t)jj * L::RS;
#pragma unroll
        for (int q = 0; q < D / 4; ++q) {
            const v4f_t v = *reinterpret_cast<const v4f_t*>(p + 4 * q);
            r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
        }
        r[D] = p[L::W_OFF];
    };
    auto pair = [&](const float (&r)[D + 1]) __attribute__((always_inline)) {
        v2f dp[D / 2];
        constexpr int NA = DCX_D2_ACCS(D);
        v2f acc[NA];
        acc[0] = v2f{d2_seed<KF>(a), 0.0f};
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
        const float d2 = acc[0].x + acc[0].y;
        float val, g;
        sweep_eval<KF>(d2, a, val, g);
        sc0 = fmaf(r[D], val, sc0);
        if constexpr (GRAD) {
            const float coef = g * r[D];
            const v2f c2 = {coef, coef};
#pragma unroll
            for (int k = 0; k + 1 < D; k += 2) g2[k / 2] = __builtin_elementwise_fma(c2, dp[k / 2], g2[k / 2]);
        }
    };
    // two row buffers: the next row's reads are in flight while this one is consumed (every slice holds `per` rows: the
    // staging pads the short ones with zero-weight rows)
    float ra[D + 1], rb[D + 1];
    load_row(ra, 0);
    int jj = 0;
    for (; jj + 2 <= per; jj += 2) {
        load_row(rb, jj + 1);
        pair(ra);
        load_row(ra, jj + 2 < per ? jj + 2 : jj + 1);
        pair(rb);
    }
    if (jj < per) pair(ra);
    if constexpr (GRAD) {
#pragma unroll
        for (int k = 0; k + 1 < D; k += 2) {
            gx[k] += g2[k / 2].x * kGradScale<KF>;
            gx[k + 1] += g2[k / 2].y * kGradScale<KF>;
        }
    }
}

// ---- the sweep with the (configurations x supports) . (supports x features) contraction on the matrix cores ------
// The gradient fold  gX[b, :] = sum_j coef_bj (x_b - s_j)  is  x_b * (sum_j coef_bj) - (coef[B, S] . s[S, D])[b, :].
// The second term is a dense GEMM; here it runs on v_mfma_f32_16x16x4_f32 (exact fp32, an fmaf chain in k order — the
// same arithmetic as the VALU form) while the VALU keeps the per-pair work that is not GEMM-shaped (differences,
// squared distance, kernel function).  The VALU and matrix pipes issue side by side, so the D fma per pair of the
// gradient accumulation leave the critical pipe altogether.
//   * A operand (16 configurations x 4 supports): the four coefficients a lane computed for supports j .. j+3 sit in
//     four VGPRs; a 4x4 transpose of the 16-lane groups (2 v_permlane32_swap + 2 v_permlane16_swap) turns them into
//     the A fragments of the wave's four 16-configuration tiles.
//   * B operand (4 supports x 16 columns): one dword per lane straight from the support rows (lane l reads column
//     l % 16 of row j + l / 16), prefetched one step ahead.  Columns >= D are never read back.
//   * the expanded form cancels (x * sum(coef) against coef . s), so it is only ever applied to SHORT runs of
//     supports: every MF_FLUSH steps the four accumulators go through this wave's LDS scratch back to the
//     lane-per-configuration layout, are combined with x * sum(coef) of the same run, and restart from zero.  A run's
//     partial sums stay within 4 * MF_FLUSH supports' worth of magnitude, so the rounding of the subtraction is
//     ~1e-6 of the result even when every weight has the same sign (distance-regression models).
#ifndef DCX_MF_FLUSH
#define DCX_MF_FLUSH 16
#endif
typedef float v4f __attribute__((ext_vector_type(4)));

template <int D, int KF, int CC, int MODE>
__device__ __forceinline__ void sweep_rows_mfma(const ScoreArgs& a, const float (&x)[D], const float (&up)[CC], int j0, int j1,
                                                float (&sc)[CC], float (&gx)[D], float* wscr, int lane) {
    using L = RowLayout<D, CC>;
    static_assert(D <= 16 && (D % 2) == 0 && MODE != MODE_SCORE, "MFMA sweep: even D <= 16, gradient modes");
    constexpr int PITCH = D + 1;  // odd pitch: conflict-free ds_read across configurations
    cfloat_ptr rows = (cfloat_ptr)(uintptr_t)a.rows;
    const float* rows_g = a.rows;
    const int grp = lane >> 4, col = lane & 15;
    v4f acc[4];
    // C > 1: the weight contraction K[configurations x supports] . W[supports x classes] runs on the matrix cores too
    // (KW): the kernel values of a step are transposed like the coefficients, the B operand is the W block of the same
    // four rows (classes past C read as zero), one accumulator per 16-configuration tile
    constexpr bool KW = CC > 1;
    v4f accS[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = accS[t] = v4f{0.f, 0.f, 0.f, 0.f};
    float asum = 0.0f;
    float near2 = 1e-30f;  // pairs closer than 0.1 |x| take the direct form (see pair): the expanded sweep's threshold.  At the
                           // first version's 1e-3 |x| the fold lost 2e-5 on queries planted 0.001-0.1 |x| from a support
                           // (tests/test_gpu_parity.py::test_expanded_form_around_the_near_threshold)
#pragma unroll
    for (int k = 0; k < D; ++k) near2 = fmaf(DCX_XF_TAU * x[k], x[k], near2);

    // one support row: score accumulation on the VALU, returns the gradient coefficient
    auto pair = [&](const float (&r)[L::RS], float& val) __attribute__((always_inline)) -> float {
        v2f d2a = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k + 1 < D; k += 2) {
            const v2f xv = {x[k], x[k + 1]};
            const v2f rv = {r[k], r[k + 1]};
            const v2f dv = xv - rv;
            d2a = __builtin_elementwise_fma(dv, dv, d2a);
        }
        const float d2 = d2a.x + d2a.y;
        float g;
        sweep_eval<KF>(d2 + d2_seed<KF>(a), a, val, g);
        if constexpr (!KW) sc[0] = fmaf(r[L::W_OFF], val, sc[0]);
        float coef;
        if constexpr (MODE == MODE_GRAD_ROW) {
            coef = g * r[CC > 1 ? L::WSUM_OFF : L::W_OFF];
        } else {
            float wb = 0.0f;
#pragma unroll
            for (int c = 0; c < CC; ++c) wb = fmaf(up[c], r[L::W_OFF + c], wb);
            coef = g * wb;
        }
        // A query (almost) on top of a support: coef ~ 1/r is huge and the expanded form would subtract two huge
        // numbers.  Such a pair takes the direct form  gx += coef * (x - s)  here and leaves the GEMM with a zero
        // coefficient (r == 0 contributes exactly zero, like the VALU sweep).  Rare: one wave-uniform branch per row.
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(d2 < near2) != 0, 0)) {
            if (d2 < near2) {
#pragma unroll
                for (int k = 0; k < D; ++k) gx[k] = fmaf(coef, x[k] - r[k], gx[k]);
                coef = 0.0f;
            }
        }
        asum += coef;
        return coef;
    };
    constexpr int USED = D + CC + (CC > 1 ? 1 : 0);
    auto load_row = [&](float (&dst)[L::RS], int j) __attribute__((always_inline)) {
        cfloat_ptr r = rows + (size_t)j * L::RS;
#pragma unroll
        for (int e = 0; e < USED; ++e) dst[e] = r[e];
    };
    // accumulators -> lane-per-configuration layout through this wave's LDS scratch; gx += x * asum - (coef . s)
    auto flush = [&]() __attribute__((always_inline)) {
        if (col < D) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wscr[(16 * t + 4 * grp + i) * PITCH + col] = acc[t][i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < D; ++k) gx[k] = fmaf(x[k], asum, gx[k] - wscr[lane * PITCH + k]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if constexpr (KW) {   // the score tiles: lane (n, g) holds class n of configurations 16 t + 4 g + i
            if (col < CC) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) wscr[(16 * t + 4 * grp + i) * PITCH + col] = accS[t][i];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < CC; ++c) sc[c] += wscr[lane * PITCH + c];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = accS[t] = v4f{0.f, 0.f, 0.f, 0.f};
        asum = 0.0f;
    };
    // four per-lane values of a step (one per support row) -> the A fragments of the four 16-configuration tiles
    auto frags = [&](float c0, float c1, float c2, float c3, float (&f)[4]) __attribute__((always_inline)) {
        auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c0), __float_as_uint(c2), false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c1), __float_as_uint(c3), false, false);
        auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        f[0] = __uint_as_float(t01[0]); f[1] = __uint_as_float(t01[1]);
        f[2] = __uint_as_float(t23[0]); f[3] = __uint_as_float(t23[1]);
    };
    auto contract_kw = [&](float v0, float v1, float v2, float v3, float bw) __attribute__((always_inline)) {
        float f[4];
        frags(v0, v1, v2, v3, f);
#pragma unroll
        for (int t = 0; t < 4; ++t) accS[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[t], bw, accS[t], 0, 0, 0);
    };
    // the four coefficients of a step -> A fragments of the four 16-configuration tiles, then the MFMAs
    auto contract = [&](float c0, float c1, float c2, float c3, float bv) __attribute__((always_inline)) {
        auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c0), __float_as_uint(c2), false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c1), __float_as_uint(c3), false, false);
        auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(t01[0]), bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(t01[1]), bv, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(t23[0]), bv, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(t23[1]), bv, acc[3], 0, 0, 0);
    };

    if (j0 < j1) {
        float rowA[L::RS], rowB[L::RS], rowC[L::RS], rowD[L::RS];
        const int jl = j1 - 1;  // clamp target for the look-ahead loads (harmless re-reads at the end)
        // B operand: lane l reads column l % 16 of row j + l / 16.  A uniform base that advances by four rows per step
        // plus a constant per-lane offset (saddr + voffset addressing, no per-step vector address arithmetic).  Rows
        // past this wave's slice are real rows of the next slice or the zeroed tail padding of the row array
        // (dcx_model_create pads it): they only ever meet a zero coefficient.
        const float* bp = rows_g + (size_t)j0 * L::RS;
        const int boff = grp * L::RS + col;
        const int woff = grp * L::RS + L::W_OFF + (col < CC ? col : 0);  // B operand of the weight contraction
        const float wmask = (col < CC) ? 1.0f : 0.0f;
        load_row(rowA, j0);
        load_row(rowB, (j0 + 1 < j1) ? j0 + 1 : jl);
        float bcur = bp[boff];
        float wcur = KW ? bp[woff] * wmask : 0.0f;
        float v0, v1, v2, v3;
        int j = j0, since = 0;
        for (; j + 3 < j1; j += 4) {
            bp += 4 * L::RS;
            const float bnext = bp[boff];
            float wnext = 0.0f;
            if constexpr (KW) wnext = bp[woff] * wmask;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load_row(rowC, j + 2);
            load_row(rowD, j + 3);
            __builtin_amdgcn_sched_barrier(0);
            const float c0 = pair(rowA, v0);
            const float c1 = pair(rowB, v1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load_row(rowA, (j + 4 < j1) ? j + 4 : jl);
            load_row(rowB, (j + 5 < j1) ? j + 5 : jl);
            __builtin_amdgcn_sched_barrier(0);
            const float c2 = pair(rowC, v2);
            const float c3 = pair(rowD, v3);
            contract(c0, c1, c2, c3, bcur);
            if constexpr (KW) contract_kw(v0, v1, v2, v3, wcur);
            __builtin_amdgcn_sched_barrier(0);
            bcur = bnext;
            wcur = wnext;
            if (++since == DCX_MF_FLUSH) {
                flush();
                since = 0;
            }
        }
        // up to three rows left (rowA / rowB hold rows j and j+1): absent rows contribute a zero coefficient
        if (j < j1) {
            v1 = v2 = 0.0f;
            float c0 = pair(rowA, v0), c1 = 0.0f, c2 = 0.0f;
            if (j + 1 < j1) c1 = pair(rowB, v1);
            if (j + 2 < j1) {
                load_row(rowC, j + 2);
                c2 = pair(rowC, v2);
            }
            contract(c0, c1, c2, 0.0f, bcur);
            if constexpr (KW) contract_kw(v0, v1, v2, 0.0f, wcur);
        }
        flush();
    }
    if constexpr (kGradScale<KF> != 1.0f) {
#pragma unroll
        for (int k = 0; k < D; ++k) gx[k] *= kGradScale<KF>;
    }
}


// The parallel cross-wave fold: the nw waves of a block have left their ACC partial sums per lane in sRed[w][e][64]; every
// wave folds a few of the accumulators over the nw rows - row 0 first, then 1, 2, ..., the order a single wave would use,
// so the sums do not depend on nw's parallelism - into row 0.  The block sizes the launch rules pick are compiled in: all nw
// reads of an accumulator are in flight before the first add (a run-time trip count left one dependent LDS round trip per
// row).  Caller synchronises before and after.
template <int ACC, int E0 = 0, int E1 = ACC>   // accumulators [E0, E1) of rows with stride ACC (the persistent trajectory kernel folds class scores and gradient apart)
__device__ __forceinline__ void fold_partial_rows(float* sRed, int wave, int lane, int nw) {
    auto fold_rows = [&](auto nwc) __attribute__((always_inline)) {
        constexpr int NWC = decltype(nwc)::value;
        for (int e = E0 + wave; e < E1; e += NWC) {
            float r[NWC];
#pragma unroll
            for (int w = 0; w < NWC; ++w) r[w] = sRed[((size_t)w * ACC + e) * 64 + lane];
            float v = r[0];
#pragma unroll
            for (int w = 1; w < NWC; ++w) v += r[w];
            sRed[e * 64 + lane] = v;
        }
    };
    if (nw == 16) fold_rows(std::integral_constant<int, 16>{});
    else if (nw == 8) fold_rows(std::integral_constant<int, 8>{});
    else if (nw == 4) fold_rows(std::integral_constant<int, 4>{});
    else if (nw == 2) fold_rows(std::integral_constant<int, 2>{});
    else {
        for (int e = E0 + wave; e < E1; e += nw) {
            float v = sRed[e * 64 + lane];
            for (int w = 1; w < nw; ++w) v += sRed[((size_t)w * ACC + e) * 64 + lane];
            sRed[e * 64 + lane] = v;
        }
    }
}


template <int D, int KF, int CC, int MODE, int MAXT, bool MF = false, bool XF = false, bool XM = false, bool QT = false>
__global__ __launch_bounds__(MAXT, sweep_min_waves(D, CC, KF, MF, XM, QT, XF)) void score_kernel(const ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool GRAD = (MODE != MODE_SCORE);
    constexpr int ACC = (GRAD ? D : 0) + CC;
    constexpr int TILE = QT ? 16 : 64;   // configurations per block (QT: lane l works for configuration l & 15)
    static_assert(!QT || (!MF && !XF && !XM && qt_applies(D, CC, KF, MODE)), "the quarter tile: direct form, one class, row weights");

    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * TILE;
    const int nb = (int)((a.B - b0) < TILE ? (a.B - b0) : TILE);
    const int dof = a.dof;
    const LdsPlan lp = lds_plan(dof, a.d_fk, a.frame_floats, nw > 1 ? a.red_slots : 0, ACC, true);
    float* sQ = smem + lp.q;
    float* sX = smem + lp.x;
    float* sF = smem + lp.f;
    float* sRed = smem + lp.red;

    // QT: all the rows go into LDS, slice by slice (4 nw slices of qt_per rows; short slices end in zero rows: weight 0) - the
    // first qt_front rows with the q rows below (their loads ride on the same round trip), the rest by the waves that have no
    // part in the FK chain, meanwhile.  (Staged up front in one piece the copy cost 2.4 k cycles of a config-#2 block.)
    auto stage_rows = [&](int first_wave, int row0, int row1) __attribute__((always_inline)) {
        if constexpr (QT) {
            constexpr int RS4 = RowLayout<D, CC>::RS / 4;
            typedef float v4f_t __attribute__((ext_vector_type(4)));
            const v4f_t* src = reinterpret_cast<const v4f_t*>(a.rows);
            constexpr int RSQ = RowLayout<D, CC>::RS;
            const int p0 = a.qt_per_g[0], p1 = a.qt_per_g[1], p2 = a.qt_per_g[2], p3 = a.qt_per_g[3];
            const int r1 = 16 * p0, r2 = r1 + 16 * p1, r3 = r2 + 16 * p2;                  // first row of wave groups 1, 2, 3
            const int b1 = 16 * (p0 * RSQ + 4), b2 = b1 + 16 * (p1 * RSQ + 4), b3 = b2 + 16 * (p2 * RSQ + 4);   // ... and their LDS offsets
            float* dst = smem + a.qt_off;
            for (int e = row0 * RS4 + (int)threadIdx.x - 64 * first_wave; e < row1 * RS4; e += (int)blockDim.x - 64 * first_wave) {
                const int row = e / RS4, q4 = e - row * RS4;   // consecutive rows of the model: group, slice of the group, row of the slice
                const int g = (row >= r3) ? 3 : (row >= r2) ? 2 : (row >= r1) ? 1 : 0;
                const int per = (g == 3) ? p3 : (g == 2) ? p2 : (g == 1) ? p1 : p0;
                const int rr = row - ((g == 3) ? r3 : (g == 2) ? r2 : (g == 1) ? r1 : 0);
                const int base = (g == 3) ? b3 : (g == 2) ? b2 : (g == 1) ? b1 : 0;
                const int sl = per > 0 ? rr / per : 0, jj = rr - sl * per;
                v4f_t v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (row < a.S) v = src[(size_t)row * RS4 + q4];
                if (per > 0 && sl < 16) *reinterpret_cast<v4f_t*>(dst + base + sl * (per * RSQ + 4) + jj * RSQ + 4 * q4) = v;
            }
        }
    };
    DCX_TS(0);
    DCX_TSB(0);
    // ---- prologue: stage the FK description and the q rows (coalesced), FK per lane on wave 0 ----
    const FkWalk fw = fk_stage_sel(a.fkk, a.fk, a.fk_dwords, a.dh, smem + lp.fk, threadIdx.x, blockDim.x);
#ifdef DCX_TIMING
    if (threadIdx.x == 0) dcx_fk_ts = (a.ts && blockIdx.x == a.ts_block && blockIdx.y == 0) ? a.ts : nullptr;
#endif
    {
        const float* qsrc = a.q + b0 * dof;
        const int n = nb * dof;
        if constexpr (QT) {
            // lane l of every wave walks the arm of configuration l & 15 (the four quarters redundantly: same cost, and each
            // ends with the features its slice of the sweep needs)
            for (int i = threadIdx.x; i < 64 * dof; i += blockDim.x) {
                const int c = (i / dof) & 15;
                sQ[i] = qsrc[(c < nb ? c : nb - 1) * dof + (i % dof)];
            }
            stage_rows(0, 0, a.qt_front);
        } else {
            for (int i = threadIdx.x; i < 64 * dof; i += blockDim.x) sQ[i] = qsrc[i < n ? i : (i % dof) + (nb - 1) * dof];
        }
    }
    __syncthreads();
    DCX_TS(1);
#if defined(DCX_ABLATE) && (DCX_ABLATE & 2)  // timing ablation only (wrong results): no FK
    if (wave == 0) for (int k = 0; k < a.d_fk; ++k) sX[k * 64 + lane] = sQ[lane * dof + (k % dof)];
#else
    fk_trig_sel(fw, a.dh, sQ + lane * dof, sF + lane, wave, nw);   // all waves: sin/cos of the joint angles
    __syncthreads();
    DCX_TS(6);
    if (a.fkk == 2 && a.jt_rows) {
        // the step table, chains of <= kDhUnroll steps: every chain split by rows over two waves, chains side by side
        dh2_chain_rows_sel(fw.dh, a.dh, sX + lane, sF + lane, wave);
        if (QT && wave >= 2 * a.dh.n_chains) stage_rows(2 * a.dh.n_chains, a.qt_front, kQtSlices * nw * a.qt_per);
    } else if (wave == 0) {
        fk_chain_sel(fw, a.dh, sQ + lane * dof, sX + lane, sF + lane);
    } else {
        stage_rows(1, a.qt_front, kQtSlices * nw * a.qt_per);
    }
#endif
    __syncthreads();

    DCX_TS(2);
    float x[D];
    if (a.d_fk == D) {  // the usual case (no padding to a compiled width): no per-feature branch on every wave
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = sX[k * 64 + lane];
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = (k < a.d_fk) ? sX[k * 64 + lane] : 0.0f;
    }
    if constexpr (XF) {
        // The expanded distance |x|^2 + |s|^2 - 2 x.s carries an absolute error ~2^-23 (|x|^2 + |s|^2) and its near-pair
        // rule is relative to |x|: both are about the distance of the data from the ORIGIN, which means nothing to a kernel
        // that depends on x - s only (kernel.py:73-79).  Rows and features are therefore shifted by the support centroid
        // (the rows once, at dcx_model_create; here one subtraction per feature): a robot based at (100, 50, 0) sweeps
        // exactly like one at the origin - without the shift every pair there was a "near" pair.
        // The shift rounds (~6e-8 |x - c| per coordinate): below the noise fp32 forward kinematics leaves in x and s anyway,
        // so it is applied to features a transform PRODUCED; raw inputs (DCX_FK_NONE) are exact numbers whose differences
        // the near-pair block can still resolve to the last bit - their "centroid" is zero (dcx_model_create).
        cfloat_ptr cen = (cfloat_ptr)(uintptr_t)a.centre;
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] -= cen[k];
    }
    if (nw > 1) __syncthreads();  // X is dead from here on: the partial sums reuse its LDS (lds_plan)

    float up[CC];
    if constexpr (MODE == MODE_GRAD_UP) {
        const int64_t bl = b0 + (lane < nb ? lane : nb - 1);
        const int hot = (a.nz > 1) ? (int)blockIdx.z : a.one_hot;
#pragma unroll
        for (int c = 0; c < CC; ++c) up[c] = (hot >= 0) ? (c == hot ? 1.0f : 0.0f) : (c < a.c_out ? a.upstream[bl * a.c_out + c] : 0.0f);
        if (a.hinge == 2) {   // the multi-class hinge: what was read are the scores of an earlier launch (wave-uniform branch)
#pragma unroll
            for (int c = 0; c < CC; ++c) up[c] = (c < a.c_out && up[c] - a.hinge_margin_c[c < 8 ? c : 7] > 0.0f) ? a.hinge_weight : 0.0f;
        }
    }

    // ---- the sweep: this wave's slice of the supports ----------------------------------------
    float sc[CC];
    float gx[D];
#pragma unroll
    for (int c = 0; c < CC; ++c) sc[c] = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) gx[k] = 0.0f;
    const int ybase = blockIdx.y * a.s_super;                               // this block's super-chunk
    const int yend = (ybase + a.s_super < a.S) ? (ybase + a.s_super) : a.S;
    int j0, j1;
    wave_slice(wave, nw, a.s_chunk, a.s_skew, ybase, yend, j0, j1);
#ifdef DCX_EXP_SAME_SLICE   // timing experiment only (wrong results): every wave of a block sweeps the SAME rows (scalar-cache hits)
    j1 -= j0 - ybase;
    j0 = ybase;
#endif

    DCX_TSB(1);
    if constexpr (QT) {
        // this lane's slice of the rows: four per wave
        constexpr int RSQ = RowLayout<D, CC>::RS;
        const int g = wave >> 2;
        const int per = a.qt_per_g[g];
        int base = 0;
        for (int h = 0; h < g; ++h) base += 16 * (a.qt_per_g[h] * RSQ + 4);
        const float* slice = smem + a.qt_off + base + ((wave & 3) * kQtSlices + (lane >> 4)) * (per * RSQ + 4);
        sweep_rows_lds<D, KF, GRAD>(a, x, slice, per, sc[0], gx);
    } else if constexpr (MF) {
        // this wave's slice of the reduction scratch doubles as its transpose buffer (X is dead, the fold comes later)
        sweep_rows_mfma<D, KF, CC, MODE>(a, x, up, j0, j1, sc, gx, sRed + (size_t)wave * ACC * 64, lane);
    } else {
        sweep_rows<D, KF, CC, MODE, XF, 0, XM>(a, x, up, j0, j1, sc, gx);
    }
    DCX_TS(3);
    {   // ---- epilogue: everything below reads the kernel arguments afresh (reload_args) and re-derives what it needs ----
    const auto& b = reload_args();
    int lane = fresh_lane();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * TILE;
    const size_t tile = (size_t)blockIdx.x * gridDim.z + blockIdx.z;
    const int nb = (int)((b.B - b0) < TILE ? (b.B - b0) : TILE);
    const int dof = b.dof;
    const LdsPlan lp = lds_plan(dof, b.d_fk, b.frame_floats, nw > 1 ? b.red_slots : 0, ACC, true);
    float* sQ = smem + lp.q;
    float* sG = smem + lp.g;
    float* sF = smem + lp.f;
    float* sRed = smem + lp.red;
    DhArgs dhb;
    DCX_COPY_DH(dhb, b.dh);
    FkWalk fw;
    fw.fkk = b.fkk;
    fw.g = b.fk;
    fw.fk = (fk_cptr)(uintptr_t)(uint32_t)(uintptr_t)(smem + lp.fk);
    fw.dh = (dh_cptr)(uintptr_t)(uint32_t)(uintptr_t)(smem + lp.fk);
    // ---- the block's partial sums meet; the tile is finished ------------------------------------------------------------
    // Round-3 epilogue ("every wave works", taken whenever the block folds in parallel): the waves fold the partial rows,
    // and in a split launch the wave that folded an accumulator also PUBLISHES it (one write-through store per wave instead
    // of ACC on wave 0), and after the arrival count the owning block's waves each RE-READ their accumulator over the ys
    // rows (ys loads in flight per wave, one round trip) - a lone wave pays ~600 cycles to drain a store, ~700 for the
    // counter and ~200 per dependent load whatever the volume (tools/lone_wave_ubench.hip), so the hand-over is priced in
    // serial round trips: publish -> drain -> count -> re-read, each ONCE per block (round 2: 7.4 k cycles, now ~2.5 k).
    // The sums are formed in the same order as before (row 0, 1, ... in the block; y = 0, 1, ... across blocks).
    const bool split = b.partial != nullptr;
    const bool par_tail = nw > 1 && b.red_slots != 1 && (!split || b.tile_done != nullptr);
    bool r1_done = false;
    if constexpr (QT && GRAD) {
        // QT: phase R1 of J^T needs the frames only - the waves that carry it are the first to leave the sweep (the LDS pipe
        // serves the oldest wave first: they finish ~5 k cycles before the last), so it runs HERE, into its own scratch columns,
        // instead of behind the fold's barrier (1.4 k cycles of a config-#2 block)
        if (b.jt_waves && b.qt_scr > 0) {
            r1_done = true;
            dh2_vjp_r1_sel(fw.dh, dhb, sF + lane, smem + b.qt_scr + lane, wave);
        }
    }
    if (par_tail) {
        float* mine = sRed + (size_t)wave * ACC * 64 + lane;
#pragma unroll
        for (int c = 0; c < CC; ++c) mine[c * 64] = sc[c];
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) mine[(CC + k) * 64] = gx[k];
        }
        __syncthreads();
        DCX_TS(16);
        // Round 4, the owner-polls hand-over (b.pwords != null; the host selects it when every block of the launch is resident
        // at once and at most half the CUs hold owners): block y = 0 OWNS its tile.  The other blocks publish each partial value
        // as ONE 8-byte (value, tag = 1) word - single-copy atomic, so a reader that sees the tag sees the value - and leave: no
        // drain, no arrival counter, no barrier.  The owner folds its own row, runs phase R1 of J^T, then polls the words of
        // y = 1, 2, ... per accumulator (the wave that owns it), adds them in that order - the order of the counter protocol,
        // bit for bit - and puts the zeros back for the next launch (or graph replay).  The counter protocol below paid
        // publish -> drain -> atomic round trip -> re-read in EVERY block: 5.6 k of a config-#2 block's 25 k cycles
        // (profiles/r04_qt.txt, first table).  Blocks that do not own never wait, so the owners' polling cannot deadlock while
        // fewer than all CUs hold owners (the host allows ONE stream per device to use this protocol: opoll_stream_ok).
        // Round 5: the tag is the launch's own number (b.ptag), not a constant: a word that a late publisher of an
        // abandoned launch leaves behind can never be taken for a word of a later launch; and an owner that gives up says so
        // in a host-visible flag (b.giveup) besides turning its tile into NaN.
        const bool opoll = split && b.pwords != nullptr;
        const bool publisher = opoll && blockIdx.y != 0;
        unsigned long long* wout = opoll ? b.pwords + ((tile * b.ys + blockIdx.y) * ACC) * 64 + lane : nullptr;
        float* out = (split && !opoll) ? b.partial + (tile * b.ys + blockIdx.y) * ACC * 64 + lane : nullptr;
        auto fold_mine = [&](auto nwc) __attribute__((always_inline)) {
            constexpr int NWC = decltype(nwc)::value;
            const int nwr = NWC > 0 ? NWC : nw;
            for (int e = wave; e < ACC; e += nwr) {
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
                if constexpr (QT) {   // the four quarters of a configuration sit 16 lanes apart
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                }
                if (publisher) __hip_atomic_store(wout + e * 64, ((unsigned long long)b.ptag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (split && !opoll) __hip_atomic_store(out + e * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else sRed[e * 64 + lane] = v;  // row 0's slot of accumulator e: only this wave reads or writes it
            }
        };
        if (nw == 16) fold_mine(std::integral_constant<int, 16>{});
        else if (nw == 8) fold_mine(std::integral_constant<int, 8>{});
        else if (nw == 4) fold_mine(std::integral_constant<int, 4>{});
        else if (nw == 2) fold_mine(std::integral_constant<int, 2>{});
        else fold_mine(std::integral_constant<int, 0>{});
        DCX_TS(17);
        if (opoll) {
            if (publisher) return;
            __syncthreads();   // every wave has folded: rows 1 .. of the scratch are free for J^T's phase R1
            const bool jt_here = GRAD && b.jt_waves;
            r1_done = jt_here && nw >= 2 + 2 * dhb.n_chains;
            if (r1_done && wave >= 2) dh2_vjp_r1_sel(fw.dh, dhb, sF + lane, sRed + (size_t)ACC * 64 + lane, wave - 2);
            const unsigned long long* wtile = b.pwords + (tile * b.ys) * ACC * 64 + lane;
            for (int e = wave; e < ACC; e += nw) {
                float tot = 0.0f;
                int spins = 0;
                for (;;) {
                    tot = sRed[e * 64 + lane];   // this block's own row (y = 0), then y = 1, 2, ...
                    bool all = true;
                    for (int y = 1; y < b.ys; y += 8) {
                        unsigned long long wv[8];
#pragma unroll
                        for (int v = 0; v < 8; ++v)
                            if (y + v < b.ys) wv[v] = __hip_atomic_load(wtile + ((size_t)(y + v) * ACC + e) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int v = 0; v < 8; ++v)
                            if (y + v < b.ys) {
                                all = all && ((unsigned int)(wv[v] >> 32) == b.ptag);
                                tot += __uint_as_float((unsigned int)wv[v]);
                            }
                    }
                    if (__builtin_amdgcn_ballot_w64(!all) == 0) break;
                    if (++spins > (1 << 21)) {   // seconds: a peer never ran.  Loud, not silent: the tile's results are NaN
                        tot = __builtin_nanf("");   // and the host hears of it before the next launch of this model
                        if (lane == 0) __hip_atomic_store(b.giveup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                for (int y = 1; y < b.ys; ++y)   // zeros back: the next launch on this stream (or graph replay) starts clean
                    __hip_atomic_store(const_cast<unsigned long long*>(wtile) + ((size_t)y * ACC + e) * 64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sRed[e * 64 + lane] = tot;
            }
            DCX_FK_TS(11, 2);
        } else if (split) {
            // Every value of the row left as an agent-scope atomic store (global_store sc1: written through to where the
            // other XCDs see it), and this wave stored nothing else.  Waiting for those stores to be acknowledged orders
            // them before the counter increment; a release fence's buffer_wbl2 would write back OTHER dirty lines and has
            // nothing of ours to do (LLVM AMDGPU memory model, gfx942 / gfx950; MI355X guide "R1": every storing wave
            // drains, then a barrier, then ONE lane signals).  -DDCX_HANDOVER_FENCE restores the fence.
#ifdef DCX_HANDOVER_FENCE
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the drained write-through hand-over relies on gfx942 / gfx950 lowering (agent-scope atomic store = global_store sc1, counted by vmcnt): build other targets with -DDCX_HANDOVER_FENCE"
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __syncthreads();
            DCX_FK_TS(8, 2);
            // "last block done": whoever sees ys - 1 earlier arrivals owns the tile and adds ALL ys rows in the fixed order
            // y = 0, 1, ... (so the result does not depend on which block came last).  No block ever waits for another.
            // rows 1 .. nw-1 of the scratch are dead after the fold: [12 n_pt columns of J^T scratch][the flag word] when J^T
            // runs on several waves (the host checked that they fit), else the flag word alone
            const bool jt_here = GRAD && b.jt_waves;
            unsigned int* flag = reinterpret_cast<unsigned int*>(sRed + ((size_t)ACC + (jt_here ? 12 * dhb.n_pt : 0)) * 64);
            // phase R1 of J^T needs the frames only: it runs on waves 2 .. beside the counter's round trip (wasted, and
            // harmless, in the blocks that turn out not to own the tile)
            r1_done = jt_here && nw >= 2 + 2 * dhb.n_chains;
            if (wave == 0) {
                if (lane == 0) *flag = __hip_atomic_fetch_add(b.tile_done + tile * kCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (r1_done) {
                dh2_vjp_r1_sel(fw.dh, dhb, sF + lane, sRed + (size_t)ACC * 64 + lane, wave - 2);
            }
            __syncthreads();
            const unsigned int arrived = __builtin_amdgcn_readfirstlane(*flag);
            DCX_FK_TS(9, 2);
            if (arrived != (unsigned int)b.ys - 1u) return;
            if (wave == 0 && lane == 0) b.tile_done[tile * kCounterStride] = 0u;  // ready for the next launch on this stream
            // The rows were published with agent-scope (sc1, write-through) stores, so agent-scope (sc1) loads read them
            // where they were written: no acquire fence.  The control dependency on `arrived` keeps the loads behind it.
            asm volatile("" ::: "memory");
            const float* part = b.partial + tile * b.ys * ACC * 64 + lane;
            for (int e = wave; e < ACC; e += nw) {
                float tot = 0.0f;
                for (int y = 0; y < b.ys; y += 8) {
                    float r[8];
#pragma unroll
                    for (int v = 0; v < 8; ++v)
                        if (y + v < b.ys) r[v] = __hip_atomic_load(part + ((size_t)(y + v) * ACC + e) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int v = 0; v < 8; ++v)
                        if (y + v < b.ys) tot += r[v];
                }
                sRed[e * 64 + lane] = tot;
            }
            DCX_FK_TS(11, 2);
        }
        __syncthreads();  // the tile's totals sit in row 0 of the scratch: [c][64] scores, then [k][64] feature gradient
        DCX_TS(4);
        if (wave == 0 && b.score != nullptr && lane < nb && blockIdx.z == 0) {
#pragma unroll
            for (int c = 0; c < CC; ++c)
                if (c < b.c_out) b.score[(b0 + lane) * b.c_out + c] = sRed[c * 64 + lane];
        }
        if constexpr (!GRAD) {
            return;
        } else {
            if (b.jt_waves) {
                // J^T on several waves (fk_device.h dh2_vjp_waves), reading the totals row in place: nothing is staged
                float scale = 1.0f;
                if constexpr (CC == 1 && MODE == MODE_GRAD_ROW) {
                    if (b.upstream != nullptr) scale = b.upstream[b0 + (lane < nb ? lane : nb - 1)];
                    if (b.hinge) scale = (sRed[lane] - b.hinge_margin > 0.0f) ? b.hinge_weight : 0.0f;
                }
                float* gq = smem + lp.q;
                float* scr = (QT && r1_done) ? smem + b.qt_scr + lane : sRed + (size_t)ACC * 64 + lane;
#if defined(DCX_ABLATE) && (DCX_ABLATE & 4)  // timing ablation only (wrong results): no J^T on the several-wave path either
                if (wave != 0) return;
                for (int i = 0; i < dof; ++i) gq[lane * dof + i] = sRed[(CC + (i % b.d_fk)) * 64 + lane] * scale;
                if (true) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    float* gdst0 = b.grad + b0 * b.grad_stride;
                    for (int i = lane; i < nb * dof; i += 64) gdst0[(int64_t)(i / dof) * b.grad_stride + (i % dof)] = gq[i];
                    return;
                }
#endif
                if (!r1_done) {  // unsplit launches (and blocks too small to run it beside the counter)
                    dh2_vjp_r1_sel(fw.dh, dhb, sF + lane, scr, wave);
                    __syncthreads();
                }
                dh2_vjp_r1b(fw.dh, dhb, sRed + CC * 64 + lane, scale, scr, gq + lane * dof, dof, wave, nw);
                __syncthreads();
                dh2_vjp_r2_sel(fw.dh, dhb, sF + lane, scr, gq + lane * dof, wave);
                if (dhb.n_chains > 1 && !dhb.shared_q) __syncthreads();  // chain 1's part of the row came from wave 1
                if (wave != 0) return;
                DCX_TS(5);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                float* gdst = b.grad + b0 * b.grad_stride + (b.nz > 1 ? (size_t)blockIdx.z * dof : 0);
                const int n = nb * dof;
                if (b.grad_stride == dof) {
                    for (int i = lane; i < n; i += 64) gdst[i] = gq[i];
                } else {
                    for (int i = lane; i < n; i += 64) gdst[(int64_t)(i / dof) * b.grad_stride + (i % dof)] = gq[i];
                }
                DCX_TSB(3);
                return;
            }
            if (wave != 0) return;
#pragma unroll
            for (int c = 0; c < CC; ++c) sc[c] = sRed[c * 64 + lane];
#pragma unroll
            for (int k = 0; k < D; ++k) gx[k] = sRed[(CC + k) * 64 + lane];
        }
    } else {
    // ---- the one-wave forms (one wave per block, wide shapes folding through one LDS row, the finish-kernel mode) ----
    if (nw > 1 && b.red_slots == 1) {
        // one LDS row: waves 1 .. nw-1 hand their partial sums to wave 0 in turn (same summation order as below)
        for (int w = 1; w < nw; ++w) {
            if (wave == w) {
#pragma unroll
                for (int c = 0; c < CC; ++c) sRed[c * 64 + lane] = sc[c];
                if constexpr (GRAD) {
#pragma unroll
                    for (int k = 0; k < D; ++k) sRed[(CC + k) * 64 + lane] = gx[k];
                }
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int c = 0; c < CC; ++c) sc[c] += sRed[c * 64 + lane];
                if constexpr (GRAD) {
#pragma unroll
                    for (int k = 0; k < D; ++k) gx[k] += sRed[(CC + k) * 64 + lane];
                }
            }
            __syncthreads();
        }
        if (wave != 0) return;
    } else if (nw > 1) {
        float* mine = sRed + (size_t)wave * ACC * 64 + lane;
#pragma unroll
        for (int c = 0; c < CC; ++c) mine[c * 64] = sc[c];
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) mine[(CC + k) * 64] = gx[k];
        }
        __syncthreads();
        fold_partial_rows<ACC>(sRed, wave, lane, nw);
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int c = 0; c < CC; ++c) sc[c] = sRed[c * 64 + lane];
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) gx[k] = sRed[(CC + k) * 64 + lane];
        }
    }

    DCX_TS(4);
    if (__builtin_expect(b.partial != nullptr, 0)) {
        // split launch (small batches): this block saw only its super-chunk
        float* out = b.partial + (tile * b.ys + blockIdx.y) * ACC * 64 + lane;
        if (b.tile_done == nullptr) {
            // score_finish_kernel adds the ys partial rows in a fixed order (deterministic) and applies J^T
#pragma unroll
            for (int c = 0; c < CC; ++c) out[c * 64] = sc[c];
            if constexpr (GRAD) {
#pragma unroll
                for (int k = 0; k < D; ++k) out[(CC + k) * 64] = gx[k];
            }
            return;
        }
        // the in-launch hand-over on ONE wave (blocks whose fold goes through one LDS row, or one wave per block): publish
        // the row write-through, drain, count; the last block to arrive re-reads every row in the order y = 0, 1, ...
#pragma unroll
        for (int c = 0; c < CC; ++c) __hip_atomic_store(out + c * 64, sc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k)
                __hip_atomic_store(out + (CC + k) * 64, gx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef DCX_HANDOVER_FENCE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        unsigned int arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(b.tile_done + tile * kCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived != (unsigned int)b.ys - 1u) return;
        if (lane == 0) b.tile_done[tile * kCounterStride] = 0u;  // ready for the next launch on this stream
        asm volatile("" ::: "memory");
        const float* part = b.partial + tile * b.ys * ACC * 64 + lane;
        // a cold path (wide shapes, one wave per block): four accumulators x four rows in flight per pass, the sums through
        // row 0 of the LDS scratch, so that nothing here claims registers of the hot paths above
#pragma unroll 1
        for (int e0 = 0; e0 < ACC; e0 += 4) {
            float tot[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) tot[u] = 0.0f;
#pragma unroll 1
            for (int y = 0; y < b.ys; y += 4) {
                float r[4][4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int yy = (y + v < b.ys) ? y + v : y;  // past the end: re-read row y, not added
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = (e0 + u < ACC) ? e0 + u : ACC - 1;
                        r[v][u] = __hip_atomic_load(part + ((size_t)yy * ACC + e) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (y + v < b.ys) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) tot[u] += r[v][u];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u < ACC) sRed[(e0 + u) * 64 + lane] = tot[u];
        }
#pragma unroll
        for (int c = 0; c < CC; ++c) sc[c] = sRed[c * 64 + lane];
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) gx[k] = sRed[(CC + k) * 64 + lane];
        }
    }

    if (b.score != nullptr && lane < nb && blockIdx.z == 0) {
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (c < b.c_out) b.score[(b0 + lane) * b.c_out + c] = sc[c];
    }
    }  // one-wave forms

    // ---- wave 0 alone: G to LDS, J^T, gradient rows out ----
    if constexpr (GRAD) {
        float scale = 1.0f;
        if constexpr (CC == 1 && MODE == MODE_GRAD_ROW) {
            if (b.upstream != nullptr) scale = b.upstream[b0 + (lane < nb ? lane : nb - 1)];
            if (b.hinge) scale = (sc[0] - b.hinge_margin > 0.0f) ? b.hinge_weight : 0.0f;
        }
        if (b.d_fk == D) {
#pragma unroll
            for (int k = 0; k < D; ++k) sG[k * 64 + lane] = gx[k] * scale;
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (k < b.d_fk) sG[k * 64 + lane] = gx[k] * scale;
        }
        DCX_FK_TS(12, 2);
        // J^T gX per lane.  The gradient row is built in place of the lane's own q row: every
        // fk_vjp branch reads what it needs from the q row before its first write to gq.
        float* gq = smem + lp.q;
#if defined(DCX_ABLATE) && (DCX_ABLATE & 1)  // timing ablation only (wrong results): no J^T
        for (int i = 0; i < dof; ++i) gq[lane * dof + i] = sG[(i % b.d_fk) * 64 + lane];
#else
        fk_vjp_sel(fw, dhb, sQ + lane * dof, sF + lane, sG + lane, gq + lane * dof, dof);
#endif
        DCX_TS(5);
        // rows -> HBM, coalesced (LDS ops of one wave complete in order; no other wave is alive)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        float* gdst = b.grad + b0 * b.grad_stride + (b.nz > 1 ? (size_t)blockIdx.z * dof : 0);
        const int n = nb * dof;
        if (b.grad_stride == dof) {
            for (int i = lane; i < n; i += 64) gdst[i] = gq[i];
        } else {
            for (int i = lane; i < n; i += 64) gdst[(int64_t)(i / dof) * b.grad_stride + (i % dof)] = gq[i];
        }
        DCX_TSB(3);
    }
    }  // epilogue
}

// Second half of a split launch: one wave per 64-configuration tile adds the ys partial rows, redoes the
// (cheap) FK for its frames and applies J^T.  Runtime D and C — this kernel is not on the VALU-bound path.
struct FinishArgs {
    const float* partial;
    const FkProg* fk;
    const float* q;
    const float* upstream;   // C == 1 only: scales the gradient row
    float* score;
    float* grad;
    int64_t B;
    int64_t grad_stride;
    int32_t ys, acc, C, Dt, dof, d_fk, frame_floats, want_grad;   // C: the compiled class count (layout of the partial rows)
    int32_t c_out;           // the caller's class count (score row stride, columns written)
    int32_t hinge;
    float hinge_margin, hinge_weight;
};

}  // namespace dcx
