// dcx_api.hip — the C ABI declared in include/dcx.h: model lifetime, argument checking,
// launch geometry.  No torch types cross this boundary.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "dcx_internal.h"
#include "pack_kernels.h"
#include "solve_kernels.h"

using namespace dcx;

struct dcx_model {
    int device = 0;
    dcx_fk_desc fk{};              // host copy
    FkProg* fk_dev = nullptr;      // device copy of the compiled program
    DhProg* dh_dev = nullptr;      // DCX_FK_DH: device copy of the step table (fk_device.h), null when the robot has none
    DhArgs dh{};                   // its control part, copied into every launch's arguments
    int32_t fk_dwords = 0;         // dwords of FkProg the transform uses
    float* rows_dev = nullptr;     // [S_active][RS]
    float* rows_p2_dev = nullptr;  // the PAIR-INTERLEAVED copy of rows_dev the two-rows-per-instruction sweeps read (score_kernel.h pair2,
                                   // p2_applies models only; padded to an even row count with a zero-weight row)
    float* rows_x2_dev = nullptr;  // the pair-interleaved copy of rows_xf_dev for the expanded sweeps that take two rows per packed
                                   // instruction (score_kernel.h X2, x2_applies models only)
    float* rows_xf_dev = nullptr;  // what the expanded-form sweeps read (score_kernel.h): the rows shifted by `centre`,
                                   // |s - c|^2 in their last column
    unsigned short* aplanes_dev = nullptr;  // XM sweep (score_kernel.h): the centred supports as bf16 planes, MFMA A-operand layout
    float* centre_dev = nullptr;   // [Dt]: the support centroid for features an FK transform produced, zero for raw inputs
    int64_t S_in = 0;
    int32_t S_active = 0;
    int64_t cap = 0;               // supports the row storage holds (dcx_model_create_ex capacity; dcx_model_update refills in place)
    size_t aplanes_cap = 0;
    float fold = 1.0f;             // factor folded into the row weights: 1/eps (Polyharmonic(1)), (2/gamma)^2 (RQ2), else 1
    int32_t* info_dev = nullptr;   // 16 bytes the packing kernel reports in (kept rows, max |s - c|^2) ...
    int32_t* info_host = nullptr;  // ... and their pinned host copy
    int32_t D = 0, Dt = 0, C = 0, RS = 0;
    int32_t Cc = 0;                // the class count the kernels are compiled for (compiled_classes(C) >= C): row layout, accumulators
    int32_t kind = 0, kf = 0;
    float kp0 = 0, kp1 = 0;
    float kp0_sweep = 0;           // what the sweeps get as ScoreArgs::kp0: 2/gamma for RQ2 (constants folded, score_kernel.h sweep_eval), else kp0
    int32_t xf_rq_ok = 0;          // RQ2: the centred features are small enough for the expanded form (score_kernel.h xf_applies)
    int32_t frame_floats = 0;
    int32_t prog_floats = 0;       // LDS floats of the staged FK program
    launch_fn launch = nullptr;
    int32_t max_threads = 0;
    int32_t n_cu = 256;
    // scratch for split launches (small batches): one fixed-size buffer per stream that has used this model, so that
    // launches on different streams never share partial rows.  Guarded by `mu`; freed only by dcx_model_destroy.
    struct Scratch {
        hipStream_t stream;
        float* ptr;
        size_t rows_bytes;   // capacity of the float rows region (the (value, tag) words behind it hold twice that)
        uint32_t epoch;      // launches that used the owner-polls words: their tags never repeat (0 = an empty word)
    };
    int32_t* giveup_host = nullptr;   // pinned, device-visible: an owner-polls launch that gave up waiting sets it (sticky)
    int32_t* giveup_dev = nullptr;
    mutable std::mutex mu;
    mutable std::vector<Scratch> scratch;
    // exchange rows of the persistent trajectory kernel's cluster form (traj_fused.h traj_exchange): one buffer per stream,
    // zeroed once; `epoch` numbers the launches that have used it (their tags never repeat)
    struct TrajExch {
        hipStream_t stream;
        unsigned long long* ptr;
        size_t bytes;
        uint32_t epoch;
    };
    mutable std::vector<TrajExch> traj_exch;
    // the sums of dcx_score_hess's moments form (hess_kernel.hip hess_moments_kernel): one buffer per stream, allocated at the
    // first such call on it, sized for the largest chunk the launcher forms (kHessMomentRows rows of nacc x 64 floats)
    struct HessMom {
        hipStream_t stream;
        float* ptr;
        size_t bytes;
    };
    mutable std::vector<HessMom> hess_mom;
};

#ifdef DCX_TIMING
static unsigned long long* g_ts_dev = nullptr;
// [32 slots][16 waves] phase stamps of one block, then [4096 blocks][4]: start, sweep start, sweep end, end of every block
// (wave 0) with the hardware id (XCC, SE, CU, SIMD) packed into the top bits of the first
constexpr size_t kTsWords = 32 * 16 + 4096 * 4;
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int fail_hip(hipError_t e, const char* what) {
    return fail(DCX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define DCX_HIP(call)                                        \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return fail_hip(e__, #call);  \
    } while (0)

traj_fused_fn traj_fused_for(int Dt) {
    switch (Dt) {
#define DCX_CASE(D) case D: return launch_traj_fused_D##D;
        DCX_CASE(2) DCX_CASE(4) DCX_CASE(6) DCX_CASE(8) DCX_CASE(12) DCX_CASE(16) DCX_CASE(18) DCX_CASE(21)
        DCX_CASE(24) DCX_CASE(27) DCX_CASE(30) DCX_CASE(32) DCX_CASE(36) DCX_CASE(42) DCX_CASE(48) DCX_CASE(54)
        DCX_CASE(60) DCX_CASE(64) DCX_CASE(72) DCX_CASE(84) DCX_CASE(96)
#undef DCX_CASE
    default: return nullptr;
    }
}

jac_fn jac_for(int Dt) {
    switch (Dt) {
#define DCX_CASE(D) case D: return launch_jac_D##D;
        DCX_CASE(2) DCX_CASE(4) DCX_CASE(6) DCX_CASE(8) DCX_CASE(12) DCX_CASE(16) DCX_CASE(18) DCX_CASE(21)
        DCX_CASE(24) DCX_CASE(27) DCX_CASE(30) DCX_CASE(32) DCX_CASE(36) DCX_CASE(42) DCX_CASE(48) DCX_CASE(54)
        DCX_CASE(60) DCX_CASE(64) DCX_CASE(72) DCX_CASE(84) DCX_CASE(96)
#undef DCX_CASE
    default: return nullptr;
    }
}

launch_fn launch_for(int Dt) {
    switch (Dt) {
#define DCX_CASE(D) case D: return launch_score_D##D;
        DCX_CASE(2) DCX_CASE(4) DCX_CASE(6) DCX_CASE(8) DCX_CASE(12) DCX_CASE(16) DCX_CASE(18) DCX_CASE(21)
        DCX_CASE(24) DCX_CASE(27) DCX_CASE(30) DCX_CASE(32) DCX_CASE(36) DCX_CASE(42) DCX_CASE(48) DCX_CASE(54)
        DCX_CASE(60) DCX_CASE(64) DCX_CASE(72) DCX_CASE(84) DCX_CASE(96)
#undef DCX_CASE
    default: return nullptr;
    }
}

int check_fk(const dcx_fk_desc& fk) {
    if (fk.dof < 1 || fk.dof > DCX_MAX_DOF) return fail(DCX_ERR_INVALID, "fk.dof out of range");
    const int D = fk.n_points * fk.point_dim;
    if (D < 1 || D > DCX_MAX_D) return fail(DCX_ERR_UNSUPPORTED, "fk feature width n_points*point_dim must be in [1, DCX_MAX_D]");
    switch (fk.kind) {
    case DCX_FK_NONE:
        if (fk.n_points * fk.point_dim != fk.dof) return fail(DCX_ERR_INVALID, "DCX_FK_NONE needs n_points*point_dim == dof");
        break;
    case DCX_FK_PLANAR:
        if (fk.point_dim != 2 || fk.n_points != fk.dof) return fail(DCX_ERR_INVALID, "DCX_FK_PLANAR needs point_dim 2, n_points == dof");
        break;
    case DCX_FK_DH: {
        if (fk.point_dim != 3 || fk.n_points > DCX_MAX_POINTS) return fail(DCX_ERR_INVALID, "DCX_FK_DH needs point_dim 3, n_points <= DCX_MAX_POINTS");
        if (fk.n_chains < 1 || fk.n_chains > DCX_MAX_CHAINS) return fail(DCX_ERR_INVALID, "DCX_FK_DH n_chains out of range");
        for (int c = 0; c < fk.n_chains; ++c) {
            if (fk.chain_len[c] < 1 || fk.chain_len[c] > DCX_MAX_JOINTS) return fail(DCX_ERR_INVALID, "DCX_FK_DH chain_len out of range");
            for (int i = 0; i < fk.chain_len[c]; ++i)
                if (fk.joint_q[c][i] < 0 || fk.joint_q[c][i] >= fk.dof) return fail(DCX_ERR_INVALID, "DCX_FK_DH joint_q out of range");
        }
        for (int k = 0; k < fk.n_points; ++k) {
            const int c = fk.pt_chain[k];
            if (c < 0 || c >= fk.n_chains || fk.pt_frame[k] < 0 || fk.pt_frame[k] >= fk.chain_len[c])
                return fail(DCX_ERR_INVALID, "DCX_FK_DH control point refers to a missing frame");
        }
    } break;
    case DCX_FK_TREE: {
        if (fk.point_dim != 3 || fk.n_points > DCX_MAX_POINTS) return fail(DCX_ERR_INVALID, "DCX_FK_TREE needs point_dim 3, n_points <= DCX_MAX_POINTS");
        if (fk.t_n_chains < 1 || fk.t_n_chains > DCX_MAX_TREE_CHAINS) return fail(DCX_ERR_UNSUPPORTED, "DCX_FK_TREE t_n_chains must be in [1, DCX_MAX_TREE_CHAINS]");
        int total = 0;
        for (int c = 0; c < fk.t_n_chains; ++c) {
            if (fk.t_chain_len[c] < 1) return fail(DCX_ERR_INVALID, "DCX_FK_TREE empty chain");
            total += fk.t_chain_len[c];
            if (total > DCX_MAX_TREE_JOINTS) return fail(DCX_ERR_UNSUPPORTED, "DCX_FK_TREE more than DCX_MAX_TREE_JOINTS joints over all chains");
        }
        {
            int n_bases = 0, rep[DCX_MAX_TREE_CHAINS];
            for (int c = 0; c < fk.t_n_chains; ++c) {
                bool seen = false;
                for (int k = 0; k < n_bases && !seen; ++k) seen = std::memcmp(fk.t_base[c], fk.t_base[rep[k]], sizeof(fk.t_base[0])) == 0;
                if (!seen) rep[n_bases++] = c;
            }
            if (n_bases > DCX_MAX_TREE_BASES) return fail(DCX_ERR_UNSUPPORTED, "DCX_FK_TREE more than DCX_MAX_TREE_BASES distinct base transforms");
        }
        for (int j = 0; j < total; ++j) {
            if (fk.t_type[j] < DCX_J_FIXED || fk.t_type[j] > DCX_J_PRISMATIC) return fail(DCX_ERR_INVALID, "DCX_FK_TREE unknown joint type");
            if (fk.t_type[j] != DCX_J_FIXED && (fk.t_q[j] < 0 || fk.t_q[j] >= fk.dof)) return fail(DCX_ERR_INVALID, "DCX_FK_TREE t_q out of range");
        }
        for (int k = 0; k < fk.n_points; ++k) {
            const int c = fk.pt_chain[k];
            if (c < 0 || c >= fk.t_n_chains || fk.pt_frame[k] < 0 || fk.pt_frame[k] >= fk.t_chain_len[c])
                return fail(DCX_ERR_INVALID, "DCX_FK_TREE control point refers to a missing frame");
        }
    } break;
    case DCX_FK_SE2:
        if (fk.dof != 3 || fk.point_dim != 2 || fk.n_points > DCX_MAX_POINTS) return fail(DCX_ERR_INVALID, "DCX_FK_SE2 needs dof 3, point_dim 2");
        break;
    case DCX_FK_SE3:
        if (fk.dof != 6 || fk.point_dim != 3 || fk.n_points > DCX_MAX_POINTS) return fail(DCX_ERR_INVALID, "DCX_FK_SE3 needs dof 6, point_dim 3");
        break;
    default: return fail(DCX_ERR_INVALID, "unknown fk.kind");
    }
    return DCX_OK;
}

int check_kernel(int kind, const float* kp) {
    if (!kp) return fail(DCX_ERR_INVALID, "kparams is NULL");
    switch (kind) {
    case DCX_K_RQ:
        if (!(kp[1] > 0.f)) return fail(DCX_ERR_INVALID, "RQ kernel needs p > 0");
        break;
    case DCX_K_POLY:
        if (kp[0] < 1.f || kp[0] != std::floor(kp[0]) || kp[0] > 64.f) return fail(DCX_ERR_INVALID, "Polyharmonic needs integer 1 <= k <= 64");
        if (kp[1] == 0.f) return fail(DCX_ERR_INVALID, "Polyharmonic needs epsilon != 0");
        break;
    case DCX_K_MQ:
        if (kp[0] == 0.f) return fail(DCX_ERR_INVALID, "MultiQuadratic needs epsilon != 0");
        break;
    default: return fail(DCX_ERR_INVALID, "unknown kernel_kind");
    }
    return DCX_OK;
}

int set_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(DCX_ERR_NO_DEVICE, "no HIP device visible (libdcx has no CPU path)");
    if (device < 0 || device >= n) return fail(DCX_ERR_INVALID, "device index out of range");
    DCX_HIP(hipSetDevice(device));
    return DCX_OK;
}

// a tiny cache of device copies of FK descriptions for the stateless fkine entry points
struct FkCacheEntry {
    int device;
    dcx_fk_desc host;
    FkProg* dev;
};
thread_local std::vector<FkCacheEntry> g_fk_cache;

int upload_fk_prog(const dcx_fk_desc& fk, FkProg** out) {
    FkProg prog;
    build_fk_prog(fk, prog);
    FkProg* d = nullptr;
    DCX_HIP(hipMalloc((void**)&d, sizeof(FkProg)));
    hipError_t e = hipMemcpy(d, &prog, sizeof(FkProg), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail_hip(e, "upload of the FK program");
    }
    *out = d;
    return DCX_OK;
}

int fk_device_copy(int device, const dcx_fk_desc& fk, FkProg** out) {
    for (auto& e : g_fk_cache)
        if (e.device == device && std::memcmp(&e.host, &fk, sizeof(fk)) == 0) {
            *out = e.dev;
            return DCX_OK;
        }
    FkProg* d = nullptr;
    if (int rc = upload_fk_prog(fk, &d)) return rc;
    if (g_fk_cache.size() >= 64) {  // bounded: drop the oldest
        (void)hipFree(g_fk_cache.front().dev);
        g_fk_cache.erase(g_fk_cache.begin());
    }
    g_fk_cache.push_back({device, fk, d});
    *out = d;
    return DCX_OK;
}

// Developer knobs: overrides of the geometry rules below for tests and A/B tools.  The DCX_* environment variables are
// read ONCE, when the library first needs them; afterwards only dcx_debug_set() changes a knob.  Nothing on the launch
// path calls getenv.  -1 = "use the rule".
struct Knobs {
    std::atomic<int64_t> ys{-1}, nw{-1}, min_rows{-1}, split_finish_kernel{-1}, inlaunch_tiles{-1}, jac_per_class{-1},
        mfma{-1}, traj_fused{-1}, xf{-1}, jac_one_sweep{-1}, train_grid{-1}, fkk{-1}, jt_waves{-1}, hess_ys{-1}, hess_form{-1}, xm{-1},
        traj_ys{-1}, traj_across{-1}, owner_poll{-1}, solve_threads{-1}, qt{-1}, giveup_inject{-1}, skew{-1}, skew8{-1};
    Knobs() {
        auto rd = [](const char* name, std::atomic<int64_t>& dst, bool flag) {
            if (const char* e = std::getenv(name)) dst = flag ? 1 : std::atoll(e);
        };
        rd("DCX_YS", ys, false);
        rd("DCX_NW", nw, false);
        rd("DCX_MIN_ROWS", min_rows, false);
        rd("DCX_SPLIT_FINISH_KERNEL", split_finish_kernel, true);
        rd("DCX_INLAUNCH_TILES", inlaunch_tiles, false);
        rd("DCX_JAC_PER_CLASS", jac_per_class, true);
        rd("DCX_MFMA", mfma, false);
        rd("DCX_TRAJ_FUSED", traj_fused, false);
        rd("DCX_XF", xf, false);
        rd("DCX_JAC_ONE_SWEEP", jac_one_sweep, false);
        rd("DCX_TRAIN_GRID", train_grid, false);
        rd("DCX_FKK", fkk, false);
        rd("DCX_JT_WAVES", jt_waves, false);
        rd("DCX_HESS_YS", hess_ys, false);
        rd("DCX_HESS_FORM", hess_form, false);
        rd("DCX_XM", xm, false);
        rd("DCX_TRAJ_YS", traj_ys, false);
        rd("DCX_TRAJ_ACROSS", traj_across, false);
        rd("DCX_OWNER_POLL", owner_poll, false);
        rd("DCX_SOLVE_THREADS", solve_threads, false);
        rd("DCX_QT", qt, false);
        rd("DCX_SKEW", skew, false);
        rd("DCX_SKEW8", skew8, false);
#ifndef DCX_WITH_MATRIX_FORMS
        if (mfma > 0) mfma = -1;   // (forms this build does not carry)
        if (xm > 0) xm = -1;
#endif
    }
};
Knobs& knobs() {
    static Knobs k;
    return k;
}

// Launch geometry.  nw = waves per block (support slices inside a block), ys = support super-chunks across
// blocks (split launch, finished inside the launch by the last block of a tile to arrive, or by score_finish_kernel).
// Measured on MI355X (profiles/r01_sweep_variants.txt, r01_split_grid.txt):
//  * occupancy is what hides the scalar-load latency of the sweep, so even a huge batch wants 8 waves per block
//    (B=1M: nw=1 436, nw=8 687 M evals/s) and a mid-size one 16 (B=65536: 548 vs 555);
//  * a batch with fewer 64-configuration tiles than CUs cannot fill the chip with one block per tile: the
//    supports are then also split across blocks (B=4096: 64 tiles on 256 CUs -> 4 blocks per tile).
struct Geometry {
    int nw, ys;
    int red_slots;  // LDS rows of the cross-wave fold: nw (parallel fold) or 1 (waves take turns), see score_kernel.h
};
Geometry pick_geometry(const dcx_model* m, int64_t B, int acc_floats, bool allow_split) {
    const int cap = m->max_threads / 64;
    const int64_t tiles = (B + 63) / 64;
    Geometry g;
    g.ys = 1;
    if (allow_split && 2 * tiles <= m->n_cu) {
        // Small batches: split the supports over ys blocks per tile until there is one block per CU, but keep ~250
        // supports per block, 16 waves each.  Graph-replay grid over ys x nw (profiles/r01_split_grid.txt): headline
        // (S=2000) B=1024 18.0 us at ys=8, B=4096 20.9 us at ys=4, B=8192 27.0 us at ys=2; config #2 (S=1000) B=4096
        // 17.7 us at ys=4 (19.2 at ys=2, 23.5 at ys=8)
        const int64_t want = std::min<int64_t>((int64_t)m->n_cu / tiles, m->S_active / 250);
        while (2 * g.ys <= want && g.ys < 32) g.ys *= 2;
        g.nw = std::min(16, cap);
    } else if (allow_split && 3 * tiles <= 2 * (int64_t)m->n_cu && (int64_t)m->S_active * m->Dt >= 18000) {
        // between n_cu / 2 and 2 n_cu / 3 tiles: thirds of the supports put at most two blocks (2/3 of a tile's sweep)
        // on a CU instead of one whole tile: headline B=9216 36.5 -> 32.1 us, B=10240 36.8 -> 32.6 us; beyond that
        // (B=12800, ys = 3, 4, 5, 8) the replicated FK and the finish cost more than the better balance gives, and so
        // they do for a light sweep (config #2, S * D = 12 000: 24.0 -> 25.0 us)
        g.ys = 3;
        g.nw = std::min(16, cap);
    } else {
        // beyond four tiles per CU, 8-wave blocks beat 16-wave ones (expanded sweep, B = 262144: 364.6 -> 350.0 us in round 2;
        // round 3 at steady clocks: B = 131072 160.7 vs 162.5 us, B = 1 M 1189 vs 1236).  At exactly four tiles per CU
        // (B = 65536 on 256 CUs) round 2 measured the same and round 3 - whose epilogue runs on all waves of a block -
        // measures the opposite: 16 waves 84.4 us, 8 waves 85.5 (three interleaved runs); five and six tiles per CU are level
        g.nw = std::min((tiles <= 4 * (int64_t)m->n_cu) ? 16 : 8, cap);
    }
    const Knobs& kn = knobs();
    if (const int64_t v = kn.ys; v >= 1 && allow_split) g.ys = (int)std::min<int64_t>(v, 64);
    if (const int64_t v = kn.nw; v >= 1) g.nw = (int)std::min<int64_t>(v, cap);
    // keep >= 15 supports per wave slice
    int min_rows = 15;
    if (const int64_t v = kn.min_rows; v >= 1) min_rows = (int)v;
    while (g.ys > 1 && m->S_active / (g.ys * g.nw) < min_rows) g.ys /= 2;
    while (g.nw > 1 && m->S_active / (g.ys * g.nw) < min_rows) g.nw /= 2;
    const int d_fk = m->fk.n_points * m->fk.point_dim;
    auto lds_bytes = [&](int nw, int slots) {
        return (size_t)(lds_plan(m->fk.dof, d_fk, m->frame_floats, nw > 1 ? slots : 0, acc_floats, true).total + m->prog_floats) * sizeof(float);
    };
    g.red_slots = g.nw;
    // LDS a block may take: a split launch is one block per CU by construction (up to 150 KB); otherwise two 64 KB blocks.
    // Round 3: a block whose nw partial rows do not fit first tries HALF the waves with the parallel fold (cross-wave fold in
    // one barrier, every wave publishing / re-reading its share of a hand-over) before it falls back to handing the rows
    // to wave 0 one at a time - config #3 (C = 5: 17 accumulators, 70 KB at 16 waves) spent 9 k cycles in that serial fold
    // and 14 k in the one-wave hand-over (profiles/r03_dev_e_phase.txt).
    const size_t lds_cap = (g.ys > 1 ? 150 : 64) * 1024;
    if (lds_bytes(g.nw, g.nw) > lds_cap && g.nw >= 8 && lds_bytes(g.nw / 2, g.nw / 2) <= lds_cap && kn.nw < 1) {  // (not against an explicit knob)
        g.nw /= 2;
        g.red_slots = g.nw;
    }
    if (lds_bytes(g.nw, g.nw) > lds_cap) {
        // Wide shapes (URDF hands, dual arms): nw partial rows of D + C floats per lane would leave one or two waves
        // per CU.  Fold through ONE row instead (LDS no longer grows with nw) and take the block size that keeps the
        // most waves resident: registers allow 4 * wps waves per CU, LDS 160 KB / block.
        const int wps = sweep_min_waves(m->Dt, m->Cc, m->kf);
        int best_nw = 1, best_waves = 0;
        for (int nw = g.nw; nw >= 1; nw /= 2) {
            const size_t lds = lds_bytes(nw, 1);
            if (lds > 64 * 1024) continue;
            const int by_lds = (int)((160 * 1024) / lds), by_regs = std::max(1, (4 * wps) / nw);
            const int waves = std::min(by_lds, by_regs) * nw;
            if (waves > best_waves) {
                best_waves = waves;
                best_nw = nw;
            }
        }
        g.nw = best_nw;
        g.red_slots = 1;
    }
    // (Round 4, tried: for a chip-filling batch of a C > 1 model the fold through ONE row instead of nw - 31 KB instead of 63 KB
    // per 8-wave block at config #3, i.e. four resident blocks per CU instead of two.  Same time (95.7 vs 96.8 us at B = 65536;
    // 4- and 16-wave blocks 102 / 97 us): residency is not what holds that sweep at 61 % VALU-busy - the two-row depth of its
    // scalar pipeline is, and a third row buffer does not fit the SGPRs.  profiles/r04_cfg3_residency.txt)
    return g;
}

// This stream's scratch buffer for the partial rows of a split launch.  It is allocated ONCE per (model, stream), at
// the size of the largest split geometry the rules above can produce for this model (at most 2 * n_cu blocks, each
// with one row of Dt + C accumulators x 64 lanes), and lives until dcx_model_destroy: a HIP graph captured on the
// stream keeps a valid pointer for the model's lifetime, and nothing on this path synchronises, frees or reallocates.
// Returns nullptr when the buffer cannot be provided right now (first use of this stream happens during a graph
// capture, the allocation failed, or a developer knob asks for more rows than the rules ever would): the caller then
// uses the unsplit geometry, which is always valid.
constexpr size_t kTileCounters = 1024;                       // arrival counters at the head of a scratch buffer
constexpr size_t kScratchHead = kTileCounters * kCounterStride * sizeof(unsigned int);  // one counter per 128-byte line
float* split_scratch(const dcx_model* m, hipStream_t st, size_t bytes, uint32_t* ptag) {
    std::lock_guard<std::mutex> lock(m->mu);
    // (`bytes` is what the launch's partial ROWS take; an existing buffer is held to the same limit as a new one - its rows
    // region - so that a knob-sized request can never spill into the tag words behind it, which must stay zero: ADVICE r4)
    for (auto& sc : m->scratch)
        if (sc.stream == st) {
            if (sc.rows_bytes < bytes) return nullptr;
            if (++sc.epoch == 0u) sc.epoch = 1u;
            *ptag = sc.epoch;
            return sc.ptr;
        }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    // [counters | float rows of the counter protocol | 8-byte (value, tag) words of the owner-polls protocol (score_kernel.h)]
    const size_t rows_bytes = (size_t)2 * m->n_cu * (m->Dt + m->Cc) * 64 * sizeof(float);
    const size_t fixed = kScratchHead + 3 * rows_bytes;
    if (bytes > rows_bytes) return nullptr;
    float* p = nullptr;
    if (hipMalloc((void**)&p, fixed) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    // zero the arrival counters and the tag words on the SAME stream (the kernels leave them at zero themselves afterwards)
    if (hipMemsetAsync(p, 0, fixed, st) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return nullptr;
    }
    m->scratch.push_back({st, p, rows_bytes, 1u});
    *ptag = 1u;
    return p;
}

// The owner-polls hand-over makes the owning blocks of a launch WAIT for the publishing ones, which is only safe while
// the publishers can always be dispatched: run_score's rule keeps one launch's owners on at most half the CUs, but launches
// that run side by side - several streams - add up.  So one stream per device may use the protocol: the first
// that asks; every other stream keeps the arrival counters, whose blocks never wait for anything.  (Another PROCESS on the
// same GPU is beyond this rule: there the bounded wait and the sticky give-up flag - dcx_model::giveup_host - apply.)
std::mutex g_opoll_mu;
struct OpollOwner { int device; hipStream_t stream; };
std::vector<OpollOwner> g_opoll_owner;
bool opoll_stream_ok(int device, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_opoll_mu);
    for (auto& o : g_opoll_owner)
        if (o.device == device) return o.stream == st;
    g_opoll_owner.push_back({device, st});
    return true;
}

// This stream's exchange rows for the cluster form of the persistent trajectory kernel, and the tag base of the launch that
// is about to use them.  Sized once for the largest grid the rule can pick (n_cu workgroups, two parities); null when
// the buffer cannot be provided now (the stream is being captured, allocation failed): the caller runs one workgroup per path.
// (value, tag) words per workgroup and parity: D + 1, or for several classes CC + 2 D (traj_fused.h, SPECULATION)
inline size_t traj_exchange_words(int Dt, int Cc) { return Cc > 1 ? (size_t)Cc + 2 * (size_t)Dt : (size_t)Dt + 1; }
unsigned long long* traj_exchange_rows(const dcx_model* m, hipStream_t st, size_t bytes, uint32_t* tag_base) {
    std::lock_guard<std::mutex> lock(m->mu);
    dcx_model::TrajExch* hit = nullptr;
    for (auto& ex : m->traj_exch)
        if (ex.stream == st) hit = &ex;
    if (hit && hit->bytes < bytes) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    if (!hit) {
        const size_t fixed = (size_t)2 * m->n_cu * traj_exchange_words(m->Dt, m->Cc) * 64 * sizeof(unsigned long long);
        if (bytes > fixed) return nullptr;
        unsigned long long* p = nullptr;
        if (hipMalloc((void**)&p, fixed) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (hipMemsetAsync(p, 0, fixed, st) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
            return nullptr;
        }
        m->traj_exch.push_back({st, p, fixed, 0u});
        hit = &m->traj_exch.back();
    }
    if (hit->epoch >= 0x00fffffeu) {  // 2^24 launches: start the tags over on zeroed rows
        if (hipMemsetAsync(hit->ptr, 0, hit->bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        hit->epoch = 0;
    }
    hit->epoch += 1;
    *tag_base = hit->epoch << 8;   // tags tag_base + 1 .. + kTrajFusedMaxIters (< 256) belong to this launch
    return hit->ptr;
}

// This stream's buffer for the sums of dcx_score_hess's moments form; null when it cannot be provided now (the stream is being
// captured at its first use, or the allocation failed): the caller then runs the form that needs none.
float* hess_moment_rows(const dcx_model* m, hipStream_t st, size_t bytes) {
    std::lock_guard<std::mutex> lock(m->mu);
    for (auto& hm : m->hess_mom)
        if (hm.stream == st) return hm.bytes >= bytes ? hm.ptr : nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    float* p = nullptr;
    if (hipMalloc((void**)&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    m->hess_mom.push_back({st, p, bytes});
    return p;
}

// The shares of a 16-wave block's rows its four wave groups (waves 0-3, 4-7, 8-11, 12-15: one wave per SIMD each) sweep, packed
// for score_kernel.h wave_slice.  A SIMD issues oldest-first, so with equal slices a block's first waves leave the sweep after
// 22 k cycles and its last after 51 k, running nearly alone at the end (profiles/r06_wave_skew.txt): 48 / 32 / 15 / 5 % instead of
// 25 % each - headline 85.4 -> 83.0 us, B = 16384 29.3 -> 25.8, config #3's shard 20.9 -> 19.7, config #5 29.3 -> 26.6 us per
// iteration; chip-filling batches of many rounds and the five-class sweeps at B = 65536 are unchanged.
// A sweep of scores alone (score_only: dcx_score - 11 VALU per pair and no gradient fold behind it) wants milder shares: with the
// gradient sweeps' 48 / 32 / 15 / 5 % it is SLOWER than with equal slices (headline 57.5 us against 56.3, B = 16384 20.0 against 19.3,
// config #3's model at B = 8192 15.9 against 13.7); 32 / 30 / 24 / 14 % reads 54.5, 17.1 and 13.1 us (profiles/r06_score_only_skew.txt).
// The persistent trajectory kernel of a several-class model slices ONCE for its score sweep and its gradient sweep (traj: it and the
// loop of launches that must stay bit-identical to it): 34 / 30 / 24 / 12 % - config #5's loop on config #3's model 55.3 -> 51.0 us
// per iteration where the indicators flip, 29.5 -> 24.6 in free space (score sweeps only), unchanged where it speculates.
inline int32_t skew_rule(int Cc = 1, bool score_only = false, bool traj = false) {
    const int64_t k = knobs().skew;
    if (k >= 0) return (int32_t)k;                     // 0: equal slices;  > 0: w0 | w1 << 10 | w2 << 20 (tests, A/B tools)
    if (traj && Cc > 1) return 340 | (300 << 10) | (240 << 20);
    if (score_only) return 320 | (300 << 10) | (240 << 20);
    // (a heavier pair body - several classes - leaves the young waves more: config #3's shard 20.7 us equal, 19.6 with the one-class
    // shares, 19.4 with 46 / 30 / 18, 18.9 with these)
    return Cc > 1 ? (420 | (300 << 10) | (200 << 20)) : (480 | (320 << 10) | (150 << 20));
}

// the same for 8-wave blocks (two wave groups): minus the per-mille share of waves 0-3 (knob skew8: 0 = equal slices)
inline int32_t skew8_rule(bool score_only = false) {
    const int64_t k = knobs().skew8;
    if (k >= 0) return -(int32_t)k;
    if (score_only) return -540;   // (scores alone at B = 262144: headline 201 us equal, 199 with 54 %, 203 with 60 %; five classes 222 / 220 / 229)
    return -600;   // (config #3's model at B = 65536: 99.7 -> 96.9 us, Panda's 21 features 79.8 -> 77.5: profiles/r06_wave_skew.txt)
}

// Which FK walk a launch of this model uses (fk_device.h FkWalk).  DH arms: the step table where the model has one, else
// the FkProg through scalar loads; knob fkk = 0 / 1 / 2 forces a walk for tests (0: the LDS walks every other kind uses).
void set_fk_walk(const dcx_model* m, ScoreArgs& a) {
    a.fk = m->fk_dev;
    a.fk_dwords = m->fk_dwords;
    a.dh = m->dh;
    a.fkk = 0;
    if (m->fk.kind == DCX_FK_DH) {
        const int64_t k = knobs().fkk;
        a.fkk = (k == 0) ? 0 : (k == 1) ? 1 : (m->dh_dev ? 2 : 1);
    }
}

struct Hinge {
    int on = 0;                 // 1: one class, applied to the gradient row;  2: several classes, `upstream` holds the scores (score_kernel.h)
    float margin = 0.f, weight = 0.f;
    float margin_c[DCX_MAX_C] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int traj_slices = 0;        // the trajectory loop's launches: both sweeps sliced like the persistent kernel slices (bit-identical to it)
};

// nz > 1 (MODE_GRAD_UP): the nz classes' one-hot sweeps in ONE launch (gridDim.z), rows to grad + z * dof.  Returns
// DCX_ERR_UNSUPPORTED without launching when that form is not available (split launch without arrival counters).
int run_score(const dcx_model* m, const float* q, int64_t B, const float* upstream, float* score, float* grad,
              int mode, int one_hot, int64_t grad_stride, hipStream_t st, Hinge hinge = Hinge(), int nz = 1) {
    if (B == 0) return DCX_OK;
    const int d_fk = m->fk.n_points * m->fk.point_dim;
    const int acc = (mode == MODE_SCORE ? 0 : m->Dt) + m->Cc;
    int64_t nblk = (B + 63) / 64;
    // the split rule sees nz launches' worth of tiles: the classes fill the chip too
    // the expanded form of the sweep (centred data, score_kernel.h) where it is compiled and not switched off
    const bool xf_able = knobs().xf != 0 && xf_applies(m->Dt, m->Cc, m->kf) && m->rows_xf_dev != nullptr &&
                         (m->kf != KF_RQ2 || m->xf_rq_ok || knobs().xf >= 2);   // (knob xf = 2 forces it past the rule: tools/xf_rq_rule.py)
    // The 16-configuration tile (score_kernel.h QT, round 4): a batch of at most 16 configurations per CU runs as blocks of 16
    // that sweep ALL the supports from an LDS copy - one block per CU, no cross-block hand-over.  One class with row weights,
    // D = 12 / 24, the specialised kernel functions; the rows and the block's own state must fit the CU's LDS with all 16 waves
    // (config #2: 64 + 53 KB; a 2000-row model does not fit and keeps the split launch).  Measured (profiles/r04_qt.txt): config #2's
    // model 10.9 - 11.0 -> 10.2 - 10.4 us at every batch from 256 to 4096.  Knob qt: 0 = never, 1 = whenever it is compiled and
    // fits with at least 4 waves.
    constexpr int QT_SLICES = kQtSlices;   // row slices per wave (score_kernel.h)
    Geometry g{};
    bool qt = false;
    int qt_per = 0;
    size_t qt_lds = 0;
    // (a knob that asks for a particular form of the split launch or of the sweep - tests, A/B tools - is honoured: the rule yields)
    const Knobs& kn0 = knobs();
    const bool asked_otherwise = kn0.ys >= 1 || kn0.xf >= 0 || kn0.mfma > 0 || kn0.xm > 0 || kn0.owner_poll >= 0 || kn0.split_finish_kernel > 0 ||
                                 kn0.inlaunch_tiles >= 0 || kn0.min_rows >= 1;
    if (const int64_t kq = kn0.qt; kq != 0 && (kq > 0 || !asked_otherwise) && qt_applies(m->Dt, m->Cc, m->kf, mode) && nz == 1 &&
                                   m->S_active >= 64 && B <= 16LL * m->n_cu) {
        int nw = std::min(16, m->max_threads / 64);
        if (const int64_t v = knobs().nw; v >= 2) nw = (int)std::min<int64_t>(v, m->max_threads / 64);
        const int nw_rule = nw;
        for (; nw >= 4; nw /= 2) {
            qt_per = (m->S_active + QT_SLICES * nw - 1) / (QT_SLICES * nw);
            const size_t plan = lds_plan(m->fk.dof, d_fk, m->frame_floats, nw, acc, true).total + m->prog_floats;
            qt_lds = sizeof(float) * (((plan + 3) & ~(size_t)3) + (size_t)QT_SLICES * nw * (qt_per * row_stride(m->Dt, m->Cc) + 4) +
                                      (size_t)(12 * m->dh.n_pt + 1) * 64);   // + J^T's own scratch columns (phase R1 beside the sweep)
            if (qt_lds <= 156 * 1024 && qt_per >= 2) break;
            if (kq < 0) { nw = 0; break; }   // (the rule: all the waves or not at all)
        }
        if (nw >= 4 && (kq > 0 || nw == nw_rule)) {
            qt = true;
            g.nw = nw;
            g.ys = 1;
            g.red_slots = nw;
        }
    }
    if (!qt) g = pick_geometry(m, B * nz, acc, true);
    float* part = nullptr;
    uint32_t ptag = 0;
    if (knobs().giveup_inject > 0 && m->giveup_host) {   // fault injection (tests): as if an owner of the previous launch had given up
        knobs().giveup_inject = -1;
        *m->giveup_host = 1;
    }
    if (m->giveup_host && *m->giveup_host)
        return fail(DCX_ERR_HIP, "an earlier split launch of this model gave up waiting for its peer workgroups (owner-polls hand-over: "
                                 "they were never scheduled - the GPU is shared with another process?); its results were NaN. "
                                 "dcx_debug_set(\"owner_poll\", 0) selects the hand-over that never waits");
    if (g.ys > 1) {
        part = split_scratch(m, st, (size_t)nblk * nz * g.ys * acc * 64 * sizeof(float), &ptag);
        if (!part) g = pick_geometry(m, B * nz, acc, false);
    }
    unsigned int* counters = nullptr;
    if (part) {
        const bool second_launch = knobs().split_finish_kernel > 0;  // A/B and tests
        // graph-replay timings (profiles/r01_sweep_small_batch_graph.txt): finishing inside the launch saves the second
        // launch and its FK.  With the rows written through L2 and re-read two at a time it wins at every batch the
        // split is used for (<= n_cu / 2 tiles): B=1024 23.3 -> 19.1 us, B=4096 29.9 -> 21.8 us, B=8192 32.5 -> 30.0 us
        // headline; 19.9 -> 18.2, 23.6 -> 19.4, 24.2 -> 21.2 us config #2
        int64_t inlaunch_max = 256;
        if (const int64_t v = knobs().inlaunch_tiles; v >= 0) inlaunch_max = std::min<int64_t>(v, (int64_t)kTileCounters);
        if (nblk * nz <= inlaunch_max && !second_launch) counters = reinterpret_cast<unsigned int*>(part);
        part = reinterpret_cast<float*>(reinterpret_cast<char*>(part) + kScratchHead);
    }
    if (nblk > 0x7fffffffLL) return fail(DCX_ERR_UNSUPPORTED, "batch too large for one launch");
    ScoreArgs a{};
    a.rows = m->rows_dev;
    set_fk_walk(m, a);
    a.q = q;
    a.upstream = upstream;
    a.score = score;
    a.grad = grad;
    a.B = B;
    a.S = m->S_active;
    a.ys = g.ys;
    a.s_super = (m->S_active + g.ys - 1) / g.ys;
    a.s_chunk = (a.s_super + g.nw - 1) / g.nw;
    // two rows per packed instruction (score_kernel.h pair2: the direct form of a narrow one-class model): the pair-interleaved
    // rows, every slice starting on an even row
    const bool p2 = !qt && !xf_able && m->rows_p2_dev != nullptr && p2_applies(m->Dt, m->Cc, m->kf);
    if (p2) {
        a.rows = m->rows_p2_dev;
        a.s_super = (a.s_super + 1) & ~1;
        a.s_chunk = ((a.s_super + g.nw - 1) / g.nw + 1) & ~1;
    }
    // 16-wave blocks: the slices of a block's four wave groups are not equal (score_kernel.h wave_slice; knob skew: 0 = equal, > 0 = packed shares)
    // (slices of fewer than ~24 rows stay equal unless a knob asks: a three-row slice is all pipeline prologue - config #5's 32-restart
    // shard, 16 rows per wave, read 13.7 us skewed against 13.0)
    const bool so = mode == MODE_SCORE && !hinge.traj_slices;   // (scores alone: their own shares, see skew_rule)
    a.s_skew = (g.nw == 16 && !qt && (a.s_chunk >= 24 || knobs().skew > 0)) ? skew_rule(m->Cc, so, hinge.traj_slices != 0) : (g.nw == 8 && !qt && (a.s_chunk >= 24 || knobs().skew8 > 0)) ? skew8_rule(so) : 0;
    a.red_slots = g.red_slots;
    a.dof = m->fk.dof;
    a.d_fk = d_fk;
    a.frame_floats = m->frame_floats;
    a.kind = m->kind;
    a.c_out = m->C;
    a.one_hot = one_hot;
    a.nz = nz;
    a.grad_stride = grad_stride;
    a.kp0 = m->kp0_sweep;
    a.kp1 = m->kp1;
    a.hinge = hinge.on;
    a.hinge_margin = hinge.margin;
    a.hinge_weight = hinge.weight;
    for (int c = 0; c < DCX_MAX_C; ++c) a.hinge_margin_c[c] = hinge.margin_c[c];
    // Expanded form of the sweep (score_kernel.h XF) wherever it is compiled (Polyharmonic(1), rows <= 37 floats): 13-17 %
    // faster for chip-filling batches, 1-3 % for split launches (profiles/r02_xf_probe.txt).  Knob xf = 0: direct form.
    a.xf = xf_able ? 1 : 0;
    // MFMA form of the gradient fold: compiled for even D <= 16 with the two specialised kernel functions; needs every
    // wave to own a slice of the LDS reduction scratch (nw > 1, parallel fold)
    a.mfma = (mode != MODE_SCORE && (m->Cc == 1 || m->Cc == 5 || m->Cc == 8) && (m->Dt == 12 || m->Dt == 16) && (m->Dt % 2) == 0 && m->kf != KF_GEN && g.nw > 1 && g.red_slots == g.nw &&
              knobs().mfma != 0 && knobs().mfma > 0) ? 1 : 0;
    if (qt) {   // 16-configuration blocks, the direct form on the model's own rows
        a.mfma = 0;
        a.xf = 0;
        a.qt = 1;
        a.qt_per = qt_per;
        // the wave groups' shares of the rows (skew_rule: a SIMD issues oldest-first, see score_kernel.h wave_slice); every group
        // keeps 16 slices, their lengths differ - the same rows, the same LDS
        for (int gq = 0; gq < 4; ++gq) a.qt_per_g[gq] = qt_per;
        if (g.nw == 16 && skew_rule() > 0) {
            const int32_t sk = skew_rule();
            const int w3[3] = {sk & 1023, (sk >> 10) & 1023, (sk >> 20) & 1023};
            int left = 4 * qt_per;
            for (int gq = 0; gq < 3; ++gq) {
                a.qt_per_g[gq] = std::min(left, (int)((int64_t)4 * qt_per * w3[gq] / 1000));
                left -= a.qt_per_g[gq];
            }
            a.qt_per_g[3] = left;
        }
        a.qt_off = (int32_t)((lds_plan(a.dof, d_fk, m->frame_floats, g.nw, acc, true).total + m->prog_floats + 3) & ~3);
        a.qt_scr = a.qt_off + QT_SLICES * g.nw * (qt_per * row_stride(m->Dt, m->Cc) + 4);
        // rows copied before the first barrier: ~45 % (the FK chain hides the rest behind its 2 k cycles; tried: the loads of that part
        // issued before anything else - the FK description's own wait then waits for them too: 10.46 -> 11.5 us).  Knob qt >= 100:
        // qt - 100 percent.
        const int64_t kq = knobs().qt;
        a.qt_front = (int32_t)((int64_t)QT_SLICES * g.nw * qt_per * (kq >= 100 ? std::min<int64_t>(kq - 100, 100) : 45) / 100);
        nblk = (B + 15) / 16;
    }
    if (a.mfma) a.xf = 0;
    if (a.xf) {  // the XF kernel: the centred rows and the centroid
        a.rows = m->rows_xf_dev;
        a.centre = m->centre_dev;
    }
    // XM: the expanded form's distance on the matrix cores (score_kernel.h sweep_rows, XM).  Slices then start on the
    // 16-row blocks of the A planes.
    a.xm = (a.xf && mode == MODE_GRAD_ROW && m->aplanes_dev != nullptr && xm_applies(m->Dt, m->Cc, m->kf) && knobs().xm > 0) ? 1 : 0;
    if (a.xm) {
        a.s_skew = 0;   // (slices on the 16-row blocks of the A planes)
        a.aplanes = m->aplanes_dev;
        a.s_super = (a.s_super + 15) / 16 * 16;
        a.s_chunk = ((a.s_super + g.nw - 1) / g.nw + 15) / 16 * 16;
    } else if (a.xf && m->rows_x2_dev != nullptr && x2_applies(m->Dt, m->Cc, m->kf)) {
        // the expanded form with two rows per packed instruction (score_kernel.h X2): the pair-interleaved centred rows, slices
        // that start on even rows
        a.rows = m->rows_x2_dev;
        a.s_super = (a.s_super + 1) & ~1;
        a.s_chunk = ((a.s_super + g.nw - 1) / g.nw + 1) & ~1;
    }
    // J^T on several waves (fk_device.h dh2_vjp_waves): the step table, a parallel fold (its scratch rows 1 .. nw-1 hold 12
    // columns per point step), and not the finish-kernel mode.  Knob jt_waves = 0: wave 0 alone (tests: identical bits).
    a.jt_rows = (a.fkk == 2 && g.nw >= 2 * m->dh.n_chains && m->dh.n_chains <= 2 && m->dh.end0 <= kDhUnroll &&
                 m->dh.n_steps - m->dh.end0 <= kDhUnroll && knobs().jt_waves != 0) ? 1 : 0;
    a.jt_waves = (a.jt_rows && mode != MODE_SCORE && g.red_slots == g.nw && (g.nw - 1) * acc >= 12 * m->dh.n_pt + 1 &&
                  (g.ys == 1 || counters != nullptr)) ? 1 : 0;
#ifdef DCX_TIMING
    if (!g_ts_dev && hipMalloc((void**)&g_ts_dev, sizeof(unsigned long long) * kTsWords) == hipSuccess)
        (void)hipMemset(g_ts_dev, 0, sizeof(unsigned long long) * kTsWords);
    a.ts = g_ts_dev;
    a.ts_block = std::getenv("DCX_TS_BLOCK") ? (unsigned)std::atoi(std::getenv("DCX_TS_BLOCK")) : 0u;
#endif
    size_t lds = sizeof(float) * (lds_plan(a.dof, d_fk, m->frame_floats, g.nw > 1 ? g.red_slots : 0, acc, true).total + m->prog_floats);
    if (qt) lds = qt_lds;
    if (g.ys == 1) {
        hipError_t e = m->launch(m->kf, m->Cc, mode, g.nw, lds, nblk, a, st);
        if (e != hipSuccess) return fail_hip(e, "score kernel launch");
        return DCX_OK;
    }
    if (nz > 1 && counters == nullptr) return DCX_ERR_UNSUPPORTED;  // score_finish_kernel has no class dimension
    a.partial = part;
    a.tile_done = counters;
    // the owner-polls hand-over (score_kernel.h): in-launch finish, the parallel epilogue, every block resident at once
    // (a split launch is one block per CU) with owners on at most half the CUs.  Knob owner_poll = 0: the counter protocol.
    // Measured (profiles/r04_owner_poll.txt): config #2 11.57 -> 11.06 us, config #3's shard 20.6 -> 20.1, Polyharmonic nodes
    // 22.35 -> 21.8; Panda (22 accumulators on 8 waves: three dependent polls per wave) 15.4 -> 15.6, hence acc <= 2 nw.
    if (counters != nullptr && g.red_slots == g.nw && g.nw > 1 && knobs().owner_poll != 0 && (acc <= 2 * g.nw || knobs().owner_poll > 0) &&
        nblk * nz * g.ys <= (int64_t)m->n_cu && 2 * nblk * nz <= (int64_t)m->n_cu && m->giveup_dev != nullptr &&
        (knobs().owner_poll > 0 || opoll_stream_ok(m->device, st))) {   // (knob owner_poll = 1: the caller vouches for the stream - tests)
        a.pwords = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(part) + (size_t)2 * m->n_cu * (m->Dt + m->Cc) * 64 * sizeof(float));
        a.ptag = ptag;           // this launch's tag: words of any other launch never match (score_kernel.h)
        a.giveup = m->giveup_dev;
    }
    hipError_t e = m->launch(m->kf, m->Cc, mode, g.nw, lds, nblk, a, st);
    if (e == hipSuccess && counters == nullptr) {
        FinishArgs f{};
        f.partial = part;
        f.fk = m->fk_dev;
        f.q = q;
        f.upstream = (m->C == 1) ? upstream : nullptr;
        f.score = score;
        f.grad = grad;
        f.B = B;
        f.grad_stride = grad_stride;
        f.ys = g.ys;
        f.acc = acc;
        f.C = m->Cc;
        f.c_out = m->C;
        f.Dt = m->Dt;
        f.dof = m->fk.dof;
        f.d_fk = d_fk;
        f.frame_floats = m->frame_floats;
        f.want_grad = (mode != MODE_SCORE);
        f.hinge = hinge.on;
        f.hinge_margin = hinge.margin;
        f.hinge_weight = hinge.weight;
        e = launch_score_finish(f, nblk, sizeof(float) * (lds_plan(a.dof, d_fk, m->frame_floats, 0, 0).total + m->prog_floats), st);
    }
    if (e != hipSuccess) return fail_hip(e, "split score launch");
    return DCX_OK;
}

}  // namespace

extern "C" {

#ifdef DCX_TIMING
int dcx_debug_ts_words(void) { return (int)kTsWords; }
int dcx_debug_read_ts(unsigned long long* out) {  // developer builds only
    if (!g_ts_dev) return 1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, g_ts_dev, sizeof(unsigned long long) * kTsWords, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 3;
}
#endif

int dcx_version(void) { return DCX_VERSION; }

int dcx_debug_set(const char* name, int64_t value) {
    if (!name) return fail(DCX_ERR_INVALID, "knob name is NULL");
    Knobs& k = knobs();
    const std::string n(name);
    std::atomic<int64_t>* dst = n == "ys" ? &k.ys : n == "nw" ? &k.nw : n == "min_rows" ? &k.min_rows
        : n == "split_finish_kernel" ? &k.split_finish_kernel : n == "inlaunch_tiles" ? &k.inlaunch_tiles
        : n == "jac_per_class" ? &k.jac_per_class : n == "mfma" ? &k.mfma : n == "traj_fused" ? &k.traj_fused : n == "xf" ? &k.xf : n == "jac_one_sweep" ? &k.jac_one_sweep : n == "train_grid" ? &k.train_grid : n == "fkk" ? &k.fkk : n == "jt_waves" ? &k.jt_waves : n == "hess_ys" ? &k.hess_ys : n == "hess_form" ? &k.hess_form : n == "xm" ? &k.xm : n == "traj_ys" ? &k.traj_ys : n == "traj_across" ? &k.traj_across : n == "owner_poll" ? &k.owner_poll : n == "solve_threads" ? &k.solve_threads : n == "qt" ? &k.qt : n == "giveup_inject" ? &k.giveup_inject : n == "skew" ? &k.skew : n == "skew8" ? &k.skew8 : nullptr;
    if (!dst) return fail(DCX_ERR_INVALID, "unknown knob: " + n);
#ifndef DCX_WITH_MATRIX_FORMS
    if ((dst == &k.mfma || dst == &k.xm) && value > 0)
        return fail(DCX_ERR_UNSUPPORTED, "this build of libdcx does not carry the matrix-core forms of the sweep (measured slower, "
                                         "profiles/r03_mfma_ab.txt): make EXTRA=-DDCX_WITH_MATRIX_FORMS");
#endif
    *dst = value;
    return DCX_OK;
}

const char* dcx_last_error(void) { return g_err.c_str(); }

int dcx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- the model's rows: storage for `cap` supports, filled from host staging or by the packing kernel ---------------------
static constexpr double kXfRqRule = 32.0;   // RQ2 takes the expanded form iff gamma * max |s - c|^2 <= this (see model_fill_host)
static size_t rows_tail_floats(const dcx_model* m) { return 8 * (size_t)m->RS + 16; }  // the MFMA B-operand loads run up to 7 rows + 15 floats ahead
static size_t aplane_shorts(int64_t rows) { return ((size_t)(rows + 15) / 16 + 2) * 3 * 16 * 4 * 8; }

static void model_free_rows(dcx_model* m) {
    if (m->rows_dev) (void)hipFree(m->rows_dev);
    if (m->rows_xf_dev) (void)hipFree(m->rows_xf_dev);
    if (m->rows_p2_dev) (void)hipFree(m->rows_p2_dev);
    if (m->rows_x2_dev) (void)hipFree(m->rows_x2_dev);
    if (m->aplanes_dev) (void)hipFree(m->aplanes_dev);
    m->rows_dev = m->rows_xf_dev = m->rows_p2_dev = m->rows_x2_dev = nullptr;
    m->aplanes_dev = nullptr;
    m->cap = 0;
}

// room for `cap` rows (both copies, the same tail padding: the two are interchangeable), the centroid, the read-back words
static int model_alloc_rows(dcx_model* m, int64_t cap) {
    if (cap < 1) cap = 1;
    if (cap <= m->cap && m->rows_dev) return DCX_OK;
    // the new buffers first, the old ones freed only when both exist: a failed growth leaves the model as it was (ADVICE r4:
    // freeing first left rows_dev null beside the old S_active, and the next launch read through it)
    const size_t floats = (size_t)cap * m->RS + rows_tail_floats(m);
    float *rows = nullptr, *rows_xf = nullptr, *rows_p2 = nullptr, *rows_x2 = nullptr;
    hipError_t e = hipMalloc((void**)&rows, floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&rows_xf, floats * sizeof(float));
    if (e == hipSuccess && p2_applies(m->Dt, m->Cc, m->kf)) e = hipMalloc((void**)&rows_p2, (floats + m->RS) * sizeof(float));
    if (e == hipSuccess && x2_applies(m->Dt, m->Cc, m->kf) && xf_applies(m->Dt, m->Cc, m->kf)) e = hipMalloc((void**)&rows_x2, (floats + m->RS) * sizeof(float));
    if (e == hipSuccess && !m->centre_dev) e = hipMalloc((void**)&m->centre_dev, (size_t)m->Dt * sizeof(float));
    if (e == hipSuccess && !m->info_dev) e = hipMalloc((void**)&m->info_dev, 16);
    if (e == hipSuccess && !m->info_host) e = hipHostMalloc((void**)&m->info_host, 16, hipHostMallocDefault);
    if (e == hipSuccess && !m->giveup_host) {
        e = hipHostMalloc((void**)&m->giveup_host, sizeof(int32_t), hipHostMallocMapped);
        if (e == hipSuccess) {
            *m->giveup_host = 0;
            e = hipHostGetDevicePointer((void**)&m->giveup_dev, m->giveup_host, 0);
        }
    }
    if (e != hipSuccess) {
        if (rows) (void)hipFree(rows);
        if (rows_xf) (void)hipFree(rows_xf);
        if (rows_p2) (void)hipFree(rows_p2);
        if (rows_x2) (void)hipFree(rows_x2);
        return fail_hip(e, "device allocation of the model");
    }
    model_free_rows(m);
    m->rows_dev = rows;
    m->rows_xf_dev = rows_xf;
    m->rows_p2_dev = rows_p2;
    m->rows_x2_dev = rows_x2;
    m->cap = cap;
    return DCX_OK;
}

// Device inputs (round 4): one packing launch on the caller's stream (pack_kernels.hip) and a 16-byte read-back - the kept
// row count decides the launch geometry of everything that follows.  No bulk copy in either direction.
static int model_fill_device(dcx_model* m, const float* feat_dev, const float* w_dev, int64_t S, hipStream_t stream) {
    PackArgs a{};
    a.feat = feat_dev;
    a.w = w_dev;
    a.rows = m->rows_dev;
    a.rows_xf = m->rows_xf_dev;
    a.centre = m->centre_dev;
    a.info = m->info_dev;
    a.S = S;
    a.D = m->D;
    a.Dt = m->Dt;
    a.C = m->C;
    a.Cl = m->Cc;
    a.RS = m->RS;
    a.tail_floats = (int32_t)rows_tail_floats(m);
    a.centred = (m->fk.kind != DCX_FK_NONE) ? 1 : 0;
    a.rq2 = (m->kf == KF_RQ2) ? 1 : 0;
    a.fold = m->fold;
    a.seed = m->kp0_sweep;
    hipError_t e = launch_pack_rows(a, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(m->info_host, m->info_dev, 16, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail_hip(e, "packing of the support rows");
    double ss_max;
    std::memcpy(&ss_max, reinterpret_cast<const char*>(m->info_host) + 8, sizeof(double));
    m->S_in = S;
    m->S_active = m->info_host[0];
    m->xf_rq_ok = (m->kf == KF_RQ2 && m->fk.kind != DCX_FK_NONE && (double)m->kp0 * ss_max <= kXfRqRule) ? 1 : 0;
    return DCX_OK;
}

// Host inputs (or the XM planes asked for: their bf16 split is done here): staged and packed on the host, one copy per array
// into the model's storage.  pack_rows_kernel restates exactly this arithmetic.
static int model_fill_host(dcx_model* m, const float* support_feat, const float* weights, int64_t S, hipStream_t stream) {
    const int32_t D = m->D, C = m->C, Cl = m->Cc;
    std::vector<float> feat((size_t)S * D), w((size_t)S * C);
    if (S > 0) {
        hipError_t e = hipStreamSynchronize(stream);   // (device inputs: whatever produced them on this stream is done)
        if (e == hipSuccess) e = hipMemcpy(feat.data(), support_feat, feat.size() * sizeof(float), hipMemcpyDefault);
        if (e == hipSuccess) e = hipMemcpy(w.data(), weights, w.size() * sizeof(float), hipMemcpyDefault);
        if (e != hipSuccess) return fail_hip(e, "copy of supports/weights");
    }
    // support rows: [D coords | zero pad to Dt | C weights | (C>1) row sum | |s|^2 | pad]; all-zero-weight rows dropped.
    const float fold = m->fold;
    std::vector<float> rows;
    rows.reserve((size_t)S * m->RS + rows_tail_floats(m));
    int32_t kept = 0;
    for (int64_t j = 0; j < S; ++j) {
        bool any = false;
        for (int c = 0; c < C; ++c) any |= (w[j * C + c] != 0.0f);
        if (!any) continue;
        const size_t base = rows.size();
        rows.resize(base + m->RS, 0.0f);
        for (int k = 0; k < D; ++k) rows[base + k] = feat[j * D + k];
        float sum = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = w[j * C + c] * fold;
            rows[base + m->Dt + c] = v;
            sum += v;
        }
        if (Cl > 1) rows[base + m->Dt + Cl] = sum;   // (columns C .. Cl-1: zero weights of the padding classes)
        double ss = 0.0;  // |s|^2 of the fp32 coordinates, rounded once (RowLayout::SS_OFF; the expanded-form sweep)
        for (int k = 0; k < D; ++k) ss += (double)feat[j * D + k] * (double)feat[j * D + k];
        rows[base + m->Dt + Cl + (Cl > 1 ? 1 : 0)] = (float)ss;
        ++kept;
    }
    m->S_in = S;
    m->S_active = kept;
    m->xf_rq_ok = 0;
    // the centred copy for the expanded-form sweeps: c = mean of the kept supports (float64, rounded once); every
    // coordinate s - c formed in fp32 exactly as the kernel forms x - c, so that a query that coincides with a support
    // still gives r = 0 exactly
    std::vector<float> centre(m->Dt, 0.0f), rows_xf(rows);
    const int ss_off = m->Dt + Cl + (Cl > 1 ? 1 : 0);
    if (m->fk.kind != DCX_FK_NONE && kept > 0) {
        for (int k = 0; k < D; ++k) {
            double acc = 0.0;
            for (int32_t j = 0; j < kept; ++j) acc += (double)rows[(size_t)j * m->RS + k];
            centre[k] = (float)(acc / (double)kept);
        }
    }
    double ss_max = 0.0;
    for (int32_t j = 0; j < kept; ++j) {
        float* cen = &rows_xf[(size_t)j * m->RS];
        double ss = 0.0;
        for (int k = 0; k < D; ++k) {
            cen[k] = cen[k] - centre[k];
            ss += (double)cen[k] * (double)cen[k];
        }
        ss_max = std::max(ss_max, ss);
        // RQ2: the expanded sweep's t = d2 + 2/gamma takes its seed from this column (one rounding for the sum)
        cen[ss_off] = (m->kf == KF_RQ2) ? (float)(ss + (double)m->kp0_sweep) : (float)ss;
    }
    // RQ2 in the expanded form: features an FK transform produced, centred, with gamma * max |s - c|^2 <= 32.  The error
    // of the expanded form grows linearly in that number (tools/xf_rq_rule.py, profiles/r04_xf_rq_rule.txt: Baxter and
    // Panda, gamma 2 .. 80: 2e-6 / 2.5e-6 against float64 at 29, 3.5e-6 / 7e-6 at 58, 1e-5 at 117; the direct form
    // stays at 3e-7 .. 1e-6): 32 keeps it a factor of four inside the 1e-5 bar.  Larger gamma or workspace: direct form.
    m->xf_rq_ok = (m->kf == KF_RQ2 && m->fk.kind != DCX_FK_NONE && (double)m->kp0 * ss_max <= kXfRqRule) ? 1 : 0;
    std::vector<unsigned short> aplanes;
    if (kept > 0) {
        // XM sweep: the centred coordinates split into three bf16 planes by truncation and laid out as the A operands of
        // v_mfma_f32_16x16x32_bf16: [16-row block][chunk c][support m][k'][8], K slot 8 k' + e of chunk c = term 2c + slot / 16,
        // feature slot % 16; terms hi.hi hi.mid mid.hi hi.lo lo.hi mid.mid take the s planes hi mid hi lo hi mid
        if (xm_applies(m->Dt, C, m->kf)) {
            const int splane_of_term[6] = {0, 1, 0, 2, 0, 1};
            aplanes.assign(aplane_shorts(kept), 0);
            for (int32_t j = 0; j < kept; ++j) {
                unsigned short pl[3][16] = {};
                for (int k = 0; k < D; ++k) {
                    float r = rows_xf[(size_t)j * m->RS + k];
                    for (int p = 0; p < 3; ++p) {
                        uint32_t u;
                        std::memcpy(&u, &r, 4);
                        u &= 0xFFFF0000u;
                        float h;
                        std::memcpy(&h, &u, 4);
                        pl[p][k] = (unsigned short)(u >> 16);
                        r -= h;
                    }
                }
                for (int c = 0; c < 3; ++c)
                    for (int kq = 0; kq < 4; ++kq)
                        for (int e2 = 0; e2 < 8; ++e2) {
                            const int slot = 8 * kq + e2, term = 2 * c + slot / 16, kk = slot % 16;
                            aplanes[((((size_t)(j / 16) * 3 + c) * 16 + (j % 16)) * 4 + kq) * 8 + e2] = pl[splane_of_term[term]][kk];
                        }
            }
        }
    }
    rows.resize(rows.size() + rows_tail_floats(m), 0.0f);
    rows_xf.resize(rows.size(), 0.0f);
    hipError_t e = hipMemcpy(m->rows_dev, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->rows_xf_dev, rows_xf.data(), rows_xf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->centre_dev, centre.data(), centre.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && !aplanes.empty()) {
        if (m->aplanes_dev && m->aplanes_cap < aplanes.size()) {
            (void)hipFree(m->aplanes_dev);
            m->aplanes_dev = nullptr;
        }
        if (!m->aplanes_dev) {
            m->aplanes_cap = aplane_shorts(m->cap);
            e = hipMalloc((void**)&m->aplanes_dev, m->aplanes_cap * sizeof(unsigned short));
        }
        if (e == hipSuccess) e = hipMemcpy(m->aplanes_dev, aplanes.data(), aplanes.size() * sizeof(unsigned short), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return fail_hip(e, "upload of the model's rows");
    return DCX_OK;
}

// device memory?  (a pageable host pointer is unknown to the runtime: that is an error code here, not a failure)
static bool is_device_ptr(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

// (re)fill the rows of a model whose storage holds >= S supports
// Building or refilling a model allocates and ends in a synchronisation of `stream` (the kept row count comes back to the
// host): refused, before anything is touched, while the stream is being captured.
static int refuse_capture(hipStream_t stream) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return fail(DCX_ERR_UNSUPPORTED, "a model cannot be built or refilled on a stream that is being captured (the kept row count is read back)");
    }
    return DCX_OK;
}

static int model_fill(dcx_model* m, const float* support_feat, const float* weights, int64_t S, hipStream_t stream) {
    // the XM planes (knob xm > 0: measurements only) are split on the host
    const bool on_device = S > 0 && is_device_ptr(support_feat) && is_device_ptr(weights) &&
                           !(knobs().xm > 0 && xm_applies(m->Dt, m->Cc, m->kf));
    int rc = on_device ? model_fill_device(m, support_feat, weights, S, stream) : model_fill_host(m, support_feat, weights, S, stream);
    if (rc == DCX_OK && m->rows_p2_dev) {   // the pair-interleaved copy (stream-ordered behind the rows it reads)
        hipError_t e = launch_interleave_rows(m->rows_dev, m->rows_p2_dev, (int32_t)m->S_active, m->RS, (int32_t)rows_tail_floats(m), stream);
        if (e != hipSuccess) return fail_hip(e, "interleaving of the support rows");
    }
    if (rc == DCX_OK && m->rows_x2_dev) {   // ... and of the centred rows
        hipError_t e = launch_interleave_rows(m->rows_xf_dev, m->rows_x2_dev, (int32_t)m->S_active, m->RS, (int32_t)rows_tail_floats(m), stream);
        if (e != hipSuccess) return fail_hip(e, "interleaving of the centred support rows");
    }
    return rc;
}

int dcx_model_create(dcx_model** out, int device, const dcx_fk_desc* fk, int kernel_kind, const float* kparams,
                     const float* support_feat, const float* weights, int64_t S, int32_t D, int32_t C) {
    return dcx_model_create_ex(out, device, fk, kernel_kind, kparams, support_feat, weights, S, D, C, 0, nullptr);
}

int dcx_model_create_ex(dcx_model** out, int device, const dcx_fk_desc* fk, int kernel_kind, const float* kparams,
                        const float* support_feat, const float* weights, int64_t S, int32_t D, int32_t C,
                        int64_t capacity, void* stream) {
    if (!out) return fail(DCX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (S < 0 || S > 0x7fffffffLL) return fail(DCX_ERR_INVALID, "S out of range");
    if (S > 0 && (!support_feat || !weights)) return fail(DCX_ERR_INVALID, "support_feat / weights is NULL");
    if (D < 1 || D > DCX_MAX_D) return fail(DCX_ERR_UNSUPPORTED, "D must be in [1, DCX_MAX_D]");
    if (C < 1 || C > DCX_MAX_C) return fail(DCX_ERR_UNSUPPORTED, "C must be in [1, DCX_MAX_C]");
    if (int rc = check_kernel(kernel_kind, kparams)) return rc;
    dcx_fk_desc desc;
    std::memset(&desc, 0, sizeof(desc));
    if (fk && fk->kind != DCX_FK_NONE) {
        desc = *fk;
    } else {
        desc.kind = DCX_FK_NONE;
        desc.dof = fk ? fk->dof : D;
        desc.n_points = desc.dof;
        desc.point_dim = 1;
    }
    if (int rc = check_fk(desc)) return rc;
    if (desc.n_points * desc.point_dim != D) return fail(DCX_ERR_INVALID, "D does not match the transform's feature width");
    if (int rc = set_device(device)) return rc;
    if (int rc = refuse_capture((hipStream_t)stream)) return rc;

    dcx_model* m = new (std::nothrow) dcx_model();
    if (!m) return fail(DCX_ERR_INVALID, "out of host memory");
    m->device = device;
    m->fk = desc;
    m->S_in = S;
    m->D = D;
    m->Dt = template_d_for(D);
    m->C = C;
    m->Cc = compiled_classes(C);
    m->RS = row_stride(m->Dt, m->Cc);
    m->kind = kernel_kind;
    m->kp0 = kparams[0];
    m->kp1 = kparams[1];
    m->kf = (kernel_kind == DCX_K_RQ && kparams[1] == 2.0f)     ? KF_RQ2
            : (kernel_kind == DCX_K_POLY && kparams[0] == 1.0f) ? KF_POLY1
                                                                : KF_GEN;
    m->kp0_sweep = (m->kf == KF_RQ2) ? 2.0f / m->kp0 : m->kp0;
    m->frame_floats = fk_frame_floats(desc);
    m->prog_floats = fk_prog_floats(desc);
    m->launch = launch_for(m->Dt);
    m->max_threads = max_threads_for(m->Dt);
    if (!m->launch) {
        delete m;
        return fail(DCX_ERR_UNSUPPORTED, "no compiled sweep for this feature width");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) m->n_cu = prop.multiProcessorCount;

    // Polyharmonic(k=1): the 1/eps factor is folded into the weights (score and gradient are linear in them).
    // RQKernel(p = 2): (2/gamma)^2 is folded into them (score_kernel.h sweep_eval; one rounding per weight, like 1/eps).
    m->fold = (m->kf == KF_POLY1) ? 1.0f / m->kp1 : (m->kf == KF_RQ2) ? (float)(4.0 / ((double)m->kp0 * (double)m->kp0)) : 1.0f;
    if (int rc = upload_fk_prog(m->fk, &m->fk_dev)) {
        dcx_model_destroy(m);
        return rc;
    }
    {
        FkProg prog;
        build_fk_prog(m->fk, prog);
        m->fk_dwords = prog.n_dwords;
        DhProg dh;
        if (build_dh_prog(m->fk, dh)) {
            hipError_t de = hipMalloc((void**)&m->dh_dev, sizeof(DhProg));
            if (de == hipSuccess) de = hipMemcpy(m->dh_dev, &dh, sizeof(DhProg), hipMemcpyHostToDevice);
            if (de != hipSuccess) {
                dcx_model_destroy(m);
                return fail_hip(de, "upload of the DH step table");
            }
            m->dh = dh_args_of(dh, m->dh_dev);
        }
    }
    if (capacity > 0x7fffffffLL) capacity = 0x7fffffffLL;
    int rc = model_alloc_rows(m, std::max<int64_t>(S, capacity));
    if (rc == DCX_OK) rc = model_fill(m, support_feat, weights, S, (hipStream_t)stream);
    if (rc != DCX_OK) {
        const std::string msg = g_err;
        dcx_model_destroy(m);
        g_err = msg;
        return rc;
    }
    *out = m;
    return DCX_OK;
}

int dcx_model_update(dcx_model* m, const float* support_feat, const float* weights, int64_t S, void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (S < 0 || S > 0x7fffffffLL) return fail(DCX_ERR_INVALID, "S out of range");
    if (S > 0 && (!support_feat || !weights)) return fail(DCX_ERR_INVALID, "support_feat / weights is NULL");
    if (int rc = set_device(m->device)) return rc;
    if (int rc = refuse_capture((hipStream_t)stream)) return rc;
    std::lock_guard<std::mutex> lock(m->mu);
    if (S > m->cap) {
        // more supports than the storage holds: new storage (blocking; ask dcx_model_create_ex for capacity to avoid it).
        // Launches already enqueued may still read the old rows: wait for them first.
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) return fail_hip(e, "dcx_model_update");
        if (int rc = model_alloc_rows(m, S)) return rc;
    }
    return model_fill(m, support_feat, weights, S, (hipStream_t)stream);
}

void dcx_model_destroy(dcx_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->fk_dev) (void)hipFree(m->fk_dev);
    if (m->dh_dev) (void)hipFree(m->dh_dev);
    if (m->rows_dev) (void)hipFree(m->rows_dev);
    if (m->rows_xf_dev) (void)hipFree(m->rows_xf_dev);
    if (m->rows_p2_dev) (void)hipFree(m->rows_p2_dev);
    if (m->rows_x2_dev) (void)hipFree(m->rows_x2_dev);
    if (m->centre_dev) (void)hipFree(m->centre_dev);
    if (m->aplanes_dev) (void)hipFree(m->aplanes_dev);
    if (m->info_dev) (void)hipFree(m->info_dev);
    if (m->info_host) (void)hipHostFree(m->info_host);
    if (m->giveup_host) (void)hipHostFree(m->giveup_host);
    for (auto& sc : m->scratch)
        if (sc.ptr) (void)hipFree(sc.ptr);
    for (auto& hm : m->hess_mom)
        if (hm.ptr) (void)hipFree(hm.ptr);
    for (auto& ex : m->traj_exch)
        if (ex.ptr) (void)hipFree(ex.ptr);
    delete m;
}

int dcx_model_info(const dcx_model* m, int64_t* S_active, int32_t* D, int32_t* C, int32_t* dof, int32_t* device) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (S_active) *S_active = m->S_active;
    if (D) *D = m->D;
    if (C) *C = m->C;
    if (dof) *dof = m->fk.dof;
    if (device) *device = m->device;
    return DCX_OK;
}

int dcx_score(const dcx_model* m, const float* q, int64_t B, float* score, void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (B < 0 || (B > 0 && (!q || !score))) return fail(DCX_ERR_INVALID, "q / score is NULL or B < 0");
    if (int rc = set_device(m->device)) return rc;
    return run_score(m, q, B, nullptr, score, nullptr, MODE_SCORE, -1, m->fk.dof, (hipStream_t)stream);
}

int dcx_score_grad(const dcx_model* m, const float* q, int64_t B, const float* upstream, float* score, float* grad,
                   void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (B < 0 || (B > 0 && (!q || !grad))) return fail(DCX_ERR_INVALID, "q / grad is NULL or B < 0");
    if (int rc = set_device(m->device)) return rc;
    const int mode = (m->C > 1 && upstream) ? MODE_GRAD_UP : MODE_GRAD_ROW;
    return run_score(m, q, B, upstream, score, grad, mode, -1, m->fk.dof, (hipStream_t)stream);
}

// the collision term of the optimisers for any class count: one launch for one class; for several, the class scores first and
// then the sweep whose upstream is weight * 1[score_c > margin_c] (score_kernel.h, ScoreArgs::hinge == 2)
static int run_hinge(const dcx_model* m, const float* q, int64_t B, const float* margin /* host [C] */, float weight, float* score,
                     float* grad, hipStream_t st, bool traj_slices = false) {
    Hinge h;
    h.weight = weight;
    h.traj_slices = traj_slices ? 1 : 0;
    if (m->C == 1) {
        h.on = 1;
        h.margin = margin[0];
        return run_score(m, q, B, nullptr, score, grad, MODE_GRAD_ROW, -1, m->fk.dof, st, h);
    }
    h.on = 2;
    for (int c = 0; c < m->C; ++c) h.margin_c[c] = margin[c];
    Hinge hs;   // (the scores' sweep: its own slices, or - for the trajectory loop - the persistent kernel's)
    hs.traj_slices = traj_slices ? 1 : 0;
    if (int rc = run_score(m, q, B, nullptr, score, nullptr, MODE_SCORE, -1, m->fk.dof, st, hs)) return rc;
    return run_score(m, q, B, score, nullptr, grad, MODE_GRAD_UP, -1, m->fk.dof, st, h);
}

int dcx_score_hinge_grad(const dcx_model* m, const float* q, int64_t B, float margin, float weight, float* score,
                         float* grad, void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (m->C != 1) return fail(DCX_ERR_UNSUPPORTED, "dcx_score_hinge_grad needs a C == 1 model (several classes: dcx_score_hinge_grad_mc)");
    if (B < 0 || (B > 0 && (!q || !grad))) return fail(DCX_ERR_INVALID, "q / grad is NULL or B < 0");
    if (int rc = set_device(m->device)) return rc;
    return run_hinge(m, q, B, &margin, weight, score, grad, (hipStream_t)stream);
}

int dcx_score_hinge_grad_mc(const dcx_model* m, const float* q, int64_t B, const float* margin, float weight, float* score,
                            float* grad, void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (!margin) return fail(DCX_ERR_INVALID, "margin is NULL (C host floats)");
    if (B < 0 || (B > 0 && (!q || !grad))) return fail(DCX_ERR_INVALID, "q / grad is NULL or B < 0");
    if (m->C > 1 && B > 0 && !score) return fail(DCX_ERR_INVALID, "several classes: score must not be NULL (it carries the first sweep's result to the second)");
    if (int rc = set_device(m->device)) return rc;
    return run_hinge(m, q, B, margin, weight, score, grad, (hipStream_t)stream);
}

int dcx_score_hess(const dcx_model* m, const float* q, int64_t B, const float* upstream, float* grad, float* hess,
                   void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (B < 0 || (B > 0 && (!q || !hess))) return fail(DCX_ERR_INVALID, "q / hess is NULL or B < 0");
    if (int rc = set_device(m->device)) return rc;
    if (B == 0) return DCX_OK;
    if (m->S_active == 0) {
        // a model whose weights are all zero (max_num_supports padding before training): the score is identically zero, and
        // so are its derivatives - nothing to sweep, and no slice size to derive the launch geometry from
        const size_t dof = (size_t)m->fk.dof;
        DCX_HIP(hipMemsetAsync(hess, 0, (size_t)B * dof * dof * sizeof(float), (hipStream_t)stream));
        if (grad) DCX_HIP(hipMemsetAsync(grad, 0, (size_t)B * dof * sizeof(float), (hipStream_t)stream));
        return DCX_OK;
    }
    ModelView v{};
    v.rows = m->rows_dev;
    v.fk = m->fk_dev;
    v.S = m->S_active;
    v.Dt = m->Dt;
    v.C = m->Cc;
    v.c_out = m->C;
    v.RS = m->RS;
    v.dof = m->fk.dof;
    v.d_fk = m->fk.n_points * m->fk.point_dim;
    v.frame_floats = m->frame_floats;
    v.prog_floats = m->prog_floats;
    v.kind = m->kind;
    v.kf = m->kf;
    v.kp0 = m->kp0_sweep;
    v.kp1 = m->kp1;
    v.n_cu = m->n_cu;
    v.ys_knob = (int32_t)knobs().hess_ys;
    v.form_knob = (int32_t)knobs().hess_form;
    v.fk_dh = (m->fk.kind == DCX_FK_DH && knobs().fkk != 0) ? 1 : 0;
    uint32_t unused_tag = 0;
    if (float* sc = split_scratch(m, (hipStream_t)stream, 0, &unused_tag)) {  // small batches split the supports across blocks
        v.counters = reinterpret_cast<unsigned int*>(sc);
        v.n_counters = (int32_t)kTileCounters;
        v.counter_stride = kCounterStride;
        v.scratch = sc + kScratchHead / sizeof(float);
        v.scratch_bytes = (size_t)2 * m->n_cu * (m->Dt + m->Cc) * 64 * sizeof(float);
    }
    // the moments form's sums (batches it applies to: hess_kernel.hip launch_hess; knob hess_form 0 = never)
    if (hess_moments_applies(v, B)) {
        v.mom_bytes = hess_moments_bytes(m->Dt);
        v.mom = hess_moment_rows(m, (hipStream_t)stream, v.mom_bytes);
    }
    const hipError_t e = launch_hess(v, q, B, upstream, grad, hess, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return fail(DCX_ERR_UNSUPPORTED, "dcx_score_hess: the transform's feature row does not fit the LDS in duals");
    if (e != hipSuccess) return fail_hip(e, "dcx_score_hess");
    return DCX_OK;
}

int dcx_score_jac(const dcx_model* m, const float* q, int64_t B, float* score, float* jac, void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (B < 0 || (B > 0 && (!q || !jac))) return fail(DCX_ERR_INVALID, "q / jac is NULL or B < 0");
    if (int rc = set_device(m->device)) return rc;
    if (m->C == 1) return run_score(m, q, B, nullptr, score, jac, MODE_GRAD_ROW, -1, m->fk.dof, (hipStream_t)stream);
    // one sweep per class with a one-hot upstream; rows land interleaved in jac[b, c, :].  Up to ~2 waves of blocks the
    // C sweeps go out as ONE launch (grid z = class) and run side by side; a batch that fills the chip by itself gains
    // nothing from that and keeps one launch per class.
    if ((B + 63) / 64 * m->C <= 2 * (int64_t)m->n_cu && !(knobs().jac_per_class > 0)) {
        int rc = run_score(m, q, B, nullptr, score, jac, MODE_GRAD_UP, -1, (int64_t)m->C * m->fk.dof, (hipStream_t)stream,
                           Hinge(), m->C);
        if (rc != DCX_ERR_UNSUPPORTED) return rc;
    }
    // A chip-filling batch: every class in ONE sweep (jac_kernel.h) where it is compiled (D * C + C <= 104 accumulators per
    // lane).  Knob jac_one_sweep: 0 = never, 1 = also for small batches (tests).
    const int64_t one_sweep = knobs().jac_one_sweep;
    if (jac_applies(m->Dt, m->Cc) && B > 0 && one_sweep != 0 && !(knobs().jac_per_class > 0) &&
        (one_sweep > 0 || (B + 63) / 64 * m->C > 2 * (int64_t)m->n_cu)) {
        const int d_fk = m->fk.n_points * m->fk.point_dim;
        int nw = std::min(16, m->max_threads / 64);
        if (const int64_t v = knobs().nw; v >= 1) nw = (int)std::min<int64_t>(v, m->max_threads / 64);
        int min_rows = 15;
        if (const int64_t v = knobs().min_rows; v >= 1) min_rows = (int)v;
        while (nw > 1 && m->S_active / nw < min_rows) nw /= 2;
        ScoreArgs a{};
        a.rows = m->rows_dev;
        set_fk_walk(m, a);
        a.q = q;
        a.score = score;
        a.grad = jac;
        a.B = B;
        a.S = m->S_active;
        a.ys = 1;
        a.s_super = m->S_active;
        a.s_chunk = (m->S_active + nw - 1) / nw;
        a.s_skew = (nw == 16 && (a.s_chunk >= 24 || knobs().skew > 0)) ? skew_rule(m->Cc) : (nw == 8 && (a.s_chunk >= 24 || knobs().skew8 > 0)) ? skew8_rule() : 0;
        a.dof = m->fk.dof;
        a.d_fk = d_fk;
        a.frame_floats = m->frame_floats;
        a.kind = m->kind;
        a.kp0 = m->kp0_sweep;
        a.kp1 = m->kp1;
        a.grad_stride = (int64_t)m->C * m->fk.dof;
        a.c_out = m->C;
        const size_t lds = sizeof(float) * (size_t)(lds_plan_jac(a.dof, d_fk, m->frame_floats, m->Cc * m->Dt + m->Cc, m->Cc).total + m->prog_floats);
        jac_fn fn = jac_for(m->Dt);
        if (fn && lds <= 150 * 1024) {
            hipError_t e = fn(m->kf, m->Cc, nw, lds, (B + 63) / 64, a, (hipStream_t)stream);
            if (e == hipSuccess) return DCX_OK;
            if (e != hipErrorNotSupported) return fail_hip(e, "Jacobian launch");
            (void)hipGetLastError();
        }
    }
    for (int c = 0; c < m->C; ++c) {
        int rc = run_score(m, q, B, nullptr, c == 0 ? score : nullptr, jac + (int64_t)c * m->fk.dof, MODE_GRAD_UP, c,
                           (int64_t)m->C * m->fk.dof, (hipStream_t)stream);
        if (rc) return rc;
    }
    return DCX_OK;
}

static int check_traj(const dcx_traj_state* st, const dcx_traj_opts* opt, int dof) {
    if (!st || !opt) return fail(DCX_ERR_INVALID, "traj state / opts is NULL");
    if (st->n_paths < 0 || st->n_waypoints < 2 || st->n_waypoints > 1024)
        return fail(DCX_ERR_UNSUPPORTED, "trajectory step needs 2 <= n_waypoints <= 1024 and n_paths >= 0");
    if (st->n_paths > 0 && (!st->path || !st->adam_m || !st->adam_v || !st->limits || !st->col_score || !st->col_grad ||
                            !st->stats || !st->lowest_loss || !st->lowest_obj || !st->lowest_path || !st->best_valid_obj ||
                            !st->best_valid_path || !st->done || !st->steps))
        return fail(DCX_ERR_INVALID, "a trajectory state pointer is NULL");
    if (!(opt->lr > 0.f) || !(opt->beta1 >= 0.f && opt->beta1 < 1.f) || !(opt->beta2 >= 0.f && opt->beta2 < 1.f))
        return fail(DCX_ERR_INVALID, "Adam options out of range");
    (void)dof;
    return DCX_OK;
}

static int traj_step(int device, const dcx_fk_desc* fk, const dcx_traj_state* st, const dcx_traj_opts* opt, const float* margin,
                     int32_t C, int32_t step, void* stream) {
    if (!fk) return fail(DCX_ERR_INVALID, "fk is NULL");
    if (C < 1 || C > DCX_MAX_C) return fail(DCX_ERR_UNSUPPORTED, "trajectory step: 1 <= C <= DCX_MAX_C");
    if (step < 1) return fail(DCX_ERR_INVALID, "step is 1-based");
    if (int rc = check_fk(*fk)) return rc;
    if (int rc = check_traj(st, opt, fk->dof)) return rc;
    if (int rc = set_device(device)) return rc;
    FkProg* dev = nullptr;
    if (int rc = fk_device_copy(device, *fk, &dev)) return rc;
    hipError_t e = launch_traj_adam_step(dev, *fk, *st, *opt, step, (hipStream_t)stream, C, margin);
    if (e != hipSuccess) return fail_hip(e, "trajectory step launch");
    return DCX_OK;
}

int dcx_traj_adam_step(int device, const dcx_fk_desc* fk, const dcx_traj_state* st, const dcx_traj_opts* opt,
                       int32_t step, void* stream) {
    return traj_step(device, fk, st, opt, nullptr, 1, step, stream);
}

int dcx_traj_adam_step_mc(int device, const dcx_fk_desc* fk, const dcx_traj_state* st, const dcx_traj_opts* opt,
                          const float* margin, int32_t C, int32_t step, void* stream) {
    return traj_step(device, fk, st, opt, margin, C, step, stream);
}

static int traj_run(const dcx_model* m, const dcx_traj_state* st, const dcx_traj_opts* opt, const float* margin_in, int32_t first_step,
                    int32_t n_iters, void* stream);

int dcx_traj_adam_run(const dcx_model* m, const dcx_traj_state* st, const dcx_traj_opts* opt, int32_t first_step,
                      int32_t n_iters, void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (m->C != 1) return fail(DCX_ERR_UNSUPPORTED, "dcx_traj_adam_run needs a C == 1 model (several classes: dcx_traj_adam_run_mc)");
    return traj_run(m, st, opt, nullptr, first_step, n_iters, stream);
}

int dcx_traj_adam_run_mc(const dcx_model* m, const dcx_traj_state* st, const dcx_traj_opts* opt, const float* margin,
                         int32_t first_step, int32_t n_iters, void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    return traj_run(m, st, opt, margin, first_step, n_iters, stream);
}

static int traj_run(const dcx_model* m, const dcx_traj_state* st, const dcx_traj_opts* opt, const float* margin_in, int32_t first_step,
                    int32_t n_iters, void* stream) {
    if (!opt) return fail(DCX_ERR_INVALID, "traj state / opts is NULL");
    // the per-class margins (several classes: optim.py:88-89 with safety_margin [C]); NULL = opt->safety_margin for every class
    float margin[DCX_MAX_C];
    for (int c = 0; c < DCX_MAX_C; ++c) margin[c] = (margin_in && c < m->C) ? margin_in[c] : opt->safety_margin;
    dcx_traj_opts opt1 = *opt;
    if (m->C == 1) opt1.safety_margin = margin[0];
    opt = &opt1;
    if (first_step < 1 || n_iters < 0) return fail(DCX_ERR_INVALID, "first_step is 1-based, n_iters >= 0");
    if (int rc = check_traj(st, opt, m->fk.dof)) return rc;
    if (int rc = set_device(m->device)) return rc;
    const int64_t B = (int64_t)st->n_paths * st->n_waypoints;
    if (st->n_paths == 0 || n_iters == 0) return DCX_OK;
    // One persistent launch (traj_fused.h) when a path is one tile of the sweep: W <= 64 waypoints, one workgroup per
    // path, the iterations looped inside the launch.  Waves per block as the sweep would pick them for a chip-filling
    // batch (>= 15 supports per slice), fewer if the fold rows would not fit in LDS.
    if (st->n_waypoints <= 64 && m->S_active > 0 && knobs().traj_fused != 0) {
        const int d_fk = m->fk.n_points * m->fk.point_dim;
        int nw = std::min(16, m->max_threads / 64);
        if (const int64_t v = knobs().nw; v >= 1) nw = (int)std::min<int64_t>(v, m->max_threads / 64);
        // Cluster form (traj_fused.h, round 4): with fewer paths than CUs a path's supports are split over ys workgroups that
        // exchange their partial rows once per iteration - ys = the power of two that fills the chip, at most 8, and the
        // sweep's >= 15 supports per wave slice.  Knob traj_ys: 1 = one workgroup per path always, k = k workgroups.
        int ys = 1;
        {
            int64_t want = std::min<int64_t>(8, (int64_t)m->n_cu / std::max(1, st->n_paths));
            if (const int64_t v = knobs().traj_ys; v >= 1) want = std::min<int64_t>(v, 32);
            while (2 * ys <= want) ys *= 2;
            int min_rows = 15;
            if (const int64_t v = knobs().min_rows; v >= 1) min_rows = (int)v;
            // (tried: TWO 8-wave workgroups per CU at 256 paths, so that one path's lone-wave phases run beside another path's
            // sweep - 31.8 us per iteration against 28.4 for one 16-wave workgroup: profiles/r04_traj_cluster.txt)
            while (ys > 1 && (m->S_active / (ys * nw) < min_rows || (int64_t)ys * st->n_paths > m->n_cu)) ys /= 2;
        }
        while (nw > 1 && m->S_active / (ys * nw) < 15) nw /= 2;
        // the step-table walks on several waves (fk_device.h): chains of <= kDhUnroll steps and >= 4 waves per block
        const bool dh_ok = m->fk.kind == DCX_FK_DH && m->dh_dev && knobs().fkk != 0 && knobs().fkk != 1 && knobs().jt_waves != 0 &&
                           m->dh.n_chains <= 2 && m->dh.end0 <= kDhUnroll && m->dh.n_steps - m->dh.end0 <= kDhUnroll;
        auto lds_of = [&](int w) {
            return sizeof(float) * (size_t)(traj_fused_plan(m->fk.dof, d_fk, m->frame_floats, w, m->Dt, (dh_ok && w >= 4) ? m->dh.n_pt : 0, m->Cc).total + m->prog_floats);
        };
        while (nw > 1 && lds_of(nw) > 150 * 1024) nw /= 2;
        traj_fused_fn fn = traj_fused_for(m->Dt);
        bool fused_ok = fn && lds_of(nw) <= 150 * 1024;
        if (fused_ok) {
            TrajFusedArgs a{};
            for (int c = 0; c < DCX_MAX_C; ++c) a.margin_c[c] = margin[c];
            a.sc.rows = m->rows_dev;
            set_fk_walk(m, a.sc);
            a.sc.jt_rows = a.sc.jt_waves = (dh_ok && nw >= 4 && a.sc.fkk == 2) ? 1 : 0;
#ifdef DCX_TIMING
            if (!g_ts_dev && hipMalloc((void**)&g_ts_dev, sizeof(unsigned long long) * kTsWords) == hipSuccess)
                (void)hipMemset(g_ts_dev, 0, sizeof(unsigned long long) * kTsWords);
            a.sc.ts = g_ts_dev;
#endif
            a.sc.S = m->S_active;
            // this launch's slices: ys super-chunks, nw wave slices each; a model whose direct-form sweep takes two rows per
            // instruction (score_kernel.h pair2) reads the pair-interleaved rows on even-aligned slices, like run_score
            bool traj_p2 = false;
            auto slice = [&](int ys_) {
                a.s_super = (m->S_active + ys_ - 1) / ys_;
                a.sc.s_chunk = (a.s_super + nw - 1) / nw;
                if (traj_p2) {
                    a.s_super = (a.s_super + 1) & ~1;
                    a.sc.s_chunk = ((a.s_super + nw - 1) / nw + 1) & ~1;
                }
                // (the wave groups' shares of a block's rows: run_score's rule, so that the loop of launches slices the same way)
                a.sc.s_skew = (nw == 16 && (a.sc.s_chunk >= 24 || knobs().skew > 0)) ? skew_rule(m->Cc, false, true) : (nw == 8 && (a.sc.s_chunk >= 24 || knobs().skew8 > 0)) ? skew8_rule() : 0;
            };
            a.sc.dof = m->fk.dof;
            a.sc.d_fk = d_fk;
            a.sc.frame_floats = m->frame_floats;
            a.sc.kind = m->kind;
            a.sc.c_out = m->C;
            a.sc.kp0 = m->kp0_sweep;
            a.sc.kp1 = m->kp1;
            a.sc.xf = (knobs().xf != 0 && m->kf == KF_POLY1 && xf_applies(m->Dt, m->Cc, KF_POLY1) && m->rows_xf_dev) ? 1 : 0;
            if (a.sc.xf) {
                a.sc.rows = m->rows_xf_dev;
                a.sc.centre = m->centre_dev;
                if (m->rows_x2_dev && x2_applies(m->Dt, m->Cc, m->kf)) {   // (two rows per packed instruction: score_kernel.h X2)
                    a.sc.rows = m->rows_x2_dev;
                    traj_p2 = true;
                }
            } else if (m->rows_p2_dev && p2_applies(m->Dt, m->Cc, m->kf)) {
                a.sc.rows = m->rows_p2_dev;
                traj_p2 = true;
            }
            slice(ys);
            a.st = *st;
            a.opt = *opt;
            a.n_points = m->fk.n_points;
            a.point_dim = m->fk.point_dim;
            a.coord_major = (m->fk.kind == DCX_FK_TREE && m->fk.t_coord_major) ? 1 : 0;
            for (int done = 0; done < n_iters; done += kTrajFusedMaxIters) {
                a.n_iters = std::min(kTrajFusedMaxIters, n_iters - done);
                for (int i = 0; i < a.n_iters; ++i) {
                    const double t = (double)(first_step + done + i);
                    a.bias1[i] = (float)(1.0 - std::pow((double)opt->beta1, t));
                    a.bias2_sqrt[i] = (float)std::sqrt(1.0 - std::pow((double)opt->beta2, t));
                }
                a.ys = 1;
                if (ys > 1) {
                    a.exch = traj_exchange_rows(m, (hipStream_t)stream, (size_t)st->n_paths * 2 * ys * traj_exchange_words(m->Dt, m->Cc) * 64 * sizeof(unsigned long long),
                                                &a.tag_base);
                    if (a.exch) a.ys = ys;
                    a.cl_across = knobs().traj_across > 0 ? 1 : 0;
                }
                if (a.ys != ys) {  // no exchange rows right now: one workgroup per path, the whole support set each
                    ys = 1;
                    slice(1);
                }
                hipError_t e = fn(m->kf, m->Cc, nw, lds_of(nw), st->n_paths, a, (hipStream_t)stream);
                if (e == hipErrorNotSupported && done == 0) {   // this width / kernel function has no multi-class instantiation
                    (void)hipGetLastError();
                    fused_ok = false;
                    break;
                }
                if (e != hipSuccess && a.ys > 1) {
                    // the cooperative launch was refused (the grid does not fit beside what else is resident): nothing ran
                    (void)hipGetLastError();
                    ys = a.ys = 1;
                    slice(1);
                    e = fn(m->kf, m->Cc, nw, lds_of(nw), st->n_paths, a, (hipStream_t)stream);
                }
                if (e != hipSuccess) return fail_hip(e, "fused trajectory launch");
            }
            if (fused_ok) return DCX_OK;
        }
    }
    // the loop as launches per iteration: the collision term (one sweep; several classes: scores, then the hinge-gradient sweep)
    // and the fused step (traj_kernels.hip)
    for (int it = 0; it < n_iters; ++it) {
        int rc = run_hinge(m, st->path, B, margin, opt->w_collision, const_cast<float*>(st->col_score), const_cast<float*>(st->col_grad),
                           (hipStream_t)stream, /*traj_slices=*/true);
        if (rc) return rc;
        hipError_t e = launch_traj_adam_step(m->fk_dev, m->fk, *st, *opt, first_step + it, (hipStream_t)stream, m->C, margin);
        if (e != hipSuccess) return fail_hip(e, "trajectory step launch");
    }
    return DCX_OK;
}

// ---- escape from collision (scripts/escape.py:19-38 as launches on the caller's stream) ----------------------------------------
namespace {
struct EscapeWork {
    size_t m_off, v_off, score_off, grad_off, qa_off[2], idx_off[2], count_off, total;
};
EscapeWork escape_work(int dof, int C, int64_t B) {
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    EscapeWork w;
    const size_t qd = up((size_t)B * dof * sizeof(float)), ib = up((size_t)B * sizeof(int32_t));
    w.m_off = 0;
    w.v_off = qd;
    w.score_off = 2 * qd;
    w.grad_off = w.score_off + up((size_t)B * C * sizeof(float));
    w.qa_off[0] = w.grad_off + qd;      // compaction (compact_every > 0): the running loops' configurations, dense, ping-pong
    w.qa_off[1] = w.qa_off[0] + qd;
    w.idx_off[0] = w.qa_off[1] + qd;    // ... and which configurations of the caller's they are
    w.idx_off[1] = w.idx_off[0] + ib;
    w.count_off = w.idx_off[1] + ib;
    w.total = w.count_off + 256;
    return w;
}
}  // namespace

size_t dcx_escape_work_bytes(const dcx_model* m, int64_t B) {
    if (!m || B <= 0) return 0;
    return escape_work(m->fk.dof, m->C, B).total;
}

int dcx_escape_adam(const dcx_model* m, float* q, int64_t B, const float* margin, const dcx_escape_opts* opt, void* work,
                    size_t work_bytes, float* history, int32_t* steps, void* stream) {
    if (!m) return fail(DCX_ERR_INVALID, "model is NULL");
    if (!opt) return fail(DCX_ERR_INVALID, "escape options are NULL");
    if (B < 0 || (B > 0 && (!q || !steps || !work))) return fail(DCX_ERR_INVALID, "q / steps / work is NULL or B < 0");
    if (opt->n_steps < 1 || opt->record_freq < 0 || opt->compact_every < 0)
        return fail(DCX_ERR_INVALID, "escape needs n_steps >= 1, record_freq >= 0 and compact_every >= 0");
    if (!(opt->lr > 0.f) || !(opt->beta1 >= 0.f && opt->beta1 < 1.f) || !(opt->beta2 >= 0.f && opt->beta2 < 1.f))
        return fail(DCX_ERR_INVALID, "Adam options out of range");
    if (B > INT32_MAX) return fail(DCX_ERR_UNSUPPORTED, "escape: more than 2^31 - 1 configurations");
    if (B == 0) return DCX_OK;
    const EscapeWork w = escape_work(m->fk.dof, m->C, B);
    if (work_bytes < w.total) return fail(DCX_ERR_INVALID, "escape workspace is smaller than dcx_escape_work_bytes");
    if (int rc = set_device(m->device)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int compact = opt->joint ? 0 : opt->compact_every;   // one loop: nothing to take out
    if (compact > 0) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return fail(DCX_ERR_UNSUPPORTED, "dcx_escape_adam with compact_every > 0 reads a count back: not on a stream that is being captured");
        }
    }
    char* base = (char*)work;
    hipError_t e = hipSuccess;   // (no memset: step 0's update launch initialises the moments and the counters, traj_kernels.hip)
    EscapeArgs a{};
    a.q = q;
    a.score = (const float*)(base + w.score_off);
    a.grad = (const float*)(base + w.grad_off);
    a.margin = margin;
    a.adam_m = (float*)(base + w.m_off);
    a.adam_v = (float*)(base + w.v_off);
    a.history = history;
    a.steps = steps;
    a.B = B;
    a.dof = m->fk.dof;
    a.C = m->C;
    a.record_freq = opt->record_freq;
    a.joint = opt->joint ? 1 : 0;
    a.wrap_mask = opt->wrap_mask;
    a.lr = opt->lr;
    a.beta1 = opt->beta1;
    a.beta2 = opt->beta2;
    a.eps = opt->eps;
    a.idx = nullptr;
    a.qa = nullptr;
    a.n_act = B;
    int32_t* count = (int32_t*)(base + w.count_off);
    int side = 0;   // which half of the ping-pong the NEXT compaction writes
    for (int s = 0; s < opt->n_steps; ++s) {
        // upstream = nullptr: the gradient of the row's sum over the classes (the folded weight column)
        if (int rc = run_score(m, a.qa ? a.qa : q, a.n_act, nullptr, const_cast<float*>(a.score), const_cast<float*>(a.grad),
                               MODE_GRAD_ROW, -1, m->fk.dof, st))
            return rc;
        e = launch_escape_step(a, s, s + 1 == opt->n_steps, st);
        if (e != hipSuccess) return fail_hip(e, "escape step launch");
        if (compact > 0 && (s + 1) % compact == 0 && s + 1 < opt->n_steps) {
            int32_t* idx_out = (int32_t*)(base + w.idx_off[side]);
            float* qa_out = (float*)(base + w.qa_off[side]);
            int32_t left = 0;
            e = hipMemsetAsync(count, 0, sizeof(int32_t), st);
            if (e == hipSuccess) e = launch_escape_compact(a, a.idx, a.n_act, idx_out, qa_out, count, st);
            if (e == hipSuccess) e = hipMemcpyAsync(&left, count, sizeof(int32_t), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return fail_hip(e, "escape compaction");
            if (left == 0) break;       // every loop has stopped
            a.idx = idx_out;
            a.qa = qa_out;
            a.n_act = left;
            side ^= 1;
        }
    }
    return DCX_OK;   // (every loop wrote its final record where it stopped, or behind the last step)
}

int dcx_train_perceptron(int device, int kernel_kind, const float* kparams, float beta, const float* feats, int64_t N,
                         int32_t D, const float* y, int32_t C, float* gains, float* hypothesis, float* kernel_matrix,
                         int32_t max_iteration, int32_t* info, void* stream) {
    return dcx_train_perceptron_ex(device, kernel_kind, kparams, beta, feats, N, D, y, C, gains, hypothesis, kernel_matrix,
                                   max_iteration, info, 0, stream);
}

int dcx_train_perceptron_ex(int device, int kernel_kind, const float* kparams, float beta, const float* feats, int64_t N,
                            int32_t D, const float* y, int32_t C, float* gains, float* hypothesis, float* kernel_matrix,
                            int32_t max_iteration, int32_t* info, int32_t flags, void* stream) {
    if (N < 1 || N > 0x7fffffffLL || D < 1 || C < 1 || C > 31 || max_iteration < 0)
        return fail(DCX_ERR_INVALID, "perceptron trainer: N >= 1, D >= 1, 1 <= C <= 31, max_iteration >= 0");
    if (!feats || !y || !gains || !hypothesis || !kernel_matrix || !info) return fail(DCX_ERR_INVALID, "a trainer pointer is NULL");
    if (int rc = check_kernel(kernel_kind, kparams)) return rc;
    if (int rc = set_device(device)) return rc;
    // The register-resident kernels (one label column; one workgroup up to N = 10240, several up to 131072) keep a label as
    // one sign bit; labels other than -1 / +1 (0 / 1 labels, y = 0) must take the generic loop, which evaluates the
    // reference's expressions on y itself.  The kernels decide that themselves, on the device (train_kernels.hip): no
    // entry point of this library synchronises the caller's stream, and the trainer can sit in a captured HIP graph.
    bool sign_labels = (C == 1 && N <= kTrainGridMaxN);
    // Several workgroups (one grid barrier per iteration) pay once a single workgroup would hold more than four samples
    // per thread.  Knob train_grid: 0 = never, 1 = whenever N >= 2048,
    // 2 = the generic kernel whatever the labels.
    const int64_t tg = knobs().train_grid;
    const bool grid = (flags & DCX_TRAIN_ONE_WORKGROUP) ? false : tg == 0 ? false : (tg > 0 ? N >= 2048 : N > 4096);
    if (tg == 2) sign_labels = false;  // tests: the generic one-workgroup kernel as the referee
    hipError_t e = launch_perceptron(kernel_kind, kparams[0], kparams[1], beta, feats, y, gains, hypothesis, kernel_matrix,
                                     info, (int)N, D, C, max_iteration, sign_labels, grid, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "perceptron trainer launch");
    return DCX_OK;
}

int dcx_fkine(int device, const dcx_fk_desc* fk, const float* q, int64_t B, float* X, void* stream) {
    if (!fk) return fail(DCX_ERR_INVALID, "fk is NULL");
    if (B < 0 || (B > 0 && (!q || !X))) return fail(DCX_ERR_INVALID, "q / X is NULL or B < 0");
    if (int rc = check_fk(*fk)) return rc;
    if (int rc = set_device(device)) return rc;
    FkProg* dev = nullptr;
    if (int rc = fk_device_copy(device, *fk, &dev)) return rc;
    hipError_t e = launch_fkine(dev, *fk, q, B, X, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "fkine launch");
    return DCX_OK;
}

int dcx_dh_frames(int device, const float* q, int64_t B, int32_t dof, const float* a, const float* d, const float* sin_alpha,
                  const float* cos_alpha, float* T, void* stream) {
    if (B < 0 || dof < 1 || (B > 0 && (!q || !a || !d || !sin_alpha || !cos_alpha || !T)))
        return fail(DCX_ERR_INVALID, "dcx_dh_frames: a pointer is NULL, B < 0 or dof < 1");
    if (int rc = set_device(device)) return rc;
    hipError_t e = launch_dh_frames(q, B, dof, a, d, sin_alpha, cos_alpha, T, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "dh_frames launch");
    return DCX_OK;
}

int dcx_dh_frames_vjp(int device, const float* q, int64_t B, int32_t dof, const float* a, const float* sin_alpha,
                      const float* cos_alpha, const float* gT, float* gq, void* stream) {
    if (B < 0 || dof < 1 || (B > 0 && (!q || !a || !sin_alpha || !cos_alpha || !gT || !gq)))
        return fail(DCX_ERR_INVALID, "dcx_dh_frames_vjp: a pointer is NULL, B < 0 or dof < 1");
    if (int rc = set_device(device)) return rc;
    hipError_t e = launch_dh_frames_vjp(q, B, dof, a, sin_alpha, cos_alpha, gT, gq, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "dh_frames_vjp launch");
    return DCX_OK;
}

int dcx_euler_frames(int device, const float* phi, int64_t B, float* R, void* stream) {
    if (B < 0 || (B > 0 && (!phi || !R))) return fail(DCX_ERR_INVALID, "dcx_euler_frames: phi / R is NULL or B < 0");
    if (int rc = set_device(device)) return rc;
    hipError_t e = launch_euler_frames(phi, B, R, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "euler_frames launch");
    return DCX_OK;
}

int dcx_euler_frames_vjp(int device, const float* phi, const float* gR, int64_t B, float* gphi, void* stream) {
    if (B < 0 || (B > 0 && (!phi || !gR || !gphi))) return fail(DCX_ERR_INVALID, "dcx_euler_frames_vjp: a pointer is NULL or B < 0");
    if (int rc = set_device(device)) return rc;
    hipError_t e = launch_euler_frames_vjp(phi, gR, B, gphi, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "euler_frames_vjp launch");
    return DCX_OK;
}

int dcx_fkine_vjp(int device, const dcx_fk_desc* fk, const float* q, const float* gX, int64_t B, float* gq,
                  void* stream) {
    if (!fk) return fail(DCX_ERR_INVALID, "fk is NULL");
    if (B < 0 || (B > 0 && (!q || !gX || !gq))) return fail(DCX_ERR_INVALID, "q / gX / gq is NULL or B < 0");
    if (int rc = check_fk(*fk)) return rc;
    if (int rc = set_device(device)) return rc;
    FkProg* dev = nullptr;
    if (int rc = fk_device_copy(device, *fk, &dev)) return rc;
    hipError_t e = launch_fkine_vjp(dev, *fk, q, gX, B, gq, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "fkine_vjp launch");
    return DCX_OK;
}

int dcx_kernel_matrix(int device, int kernel_kind, const float* kparams, const float* x, int64_t B, const float* s,
                      int64_t S, int32_t D, float* K, void* stream) {
    if (B < 0 || S < 0 || D < 1) return fail(DCX_ERR_INVALID, "negative size");
    if (B > 0 && S > 0 && (!x || !s || !K)) return fail(DCX_ERR_INVALID, "x / s / K is NULL");
    if (int rc = check_kernel(kernel_kind, kparams)) return rc;
    if (int rc = set_device(device)) return rc;
    // grid.y carries B/16 strips: split very tall problems
    const int64_t step = 16LL * 65535;
    for (int64_t b = 0; b < B; b += step) {
        const int64_t nb = std::min(step, B - b);
        hipError_t e = launch_kernel_matrix(kernel_kind, kparams[0], kparams[1], x + b * D, nb, s, S, D, K + b * S,
                                            (hipStream_t)stream);
        if (e != hipSuccess) return fail_hip(e, "kernel_matrix launch");
    }
    return DCX_OK;
}

int dcx_debug_clock_probe(int device, uint64_t* out4, uint64_t wall_ticks, int32_t* wall_clock_khz, void* stream) {
    if (!out4 || wall_ticks < 1 || wall_ticks > (1ull << 32)) return fail(DCX_ERR_INVALID, "clock probe: out4 is NULL or wall_ticks out of range");
    if (int rc = set_device(device)) return rc;
    if (wall_clock_khz) {
        int khz = 0;
        DCX_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
        *wall_clock_khz = khz;
    }
    const hipError_t e = launch_clock_probe(reinterpret_cast<unsigned long long*>(out4), wall_ticks, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "clock probe launch");
    return DCX_OK;
}

size_t dcx_solve_work_bytes(int64_t n, int64_t nrhs) {
    if (n < 1 || nrhs < 1) return 0;
    return solve_work_bytes(n, nrhs);
}

int dcx_solve(int device, const float* A, const float* B, int64_t n, int64_t nrhs, float* X, void* work, size_t work_bytes,
              int32_t* info, int32_t flags, void* stream) {
    if (n < 1 || n > DCX_SOLVE_MAX_N || nrhs < 1 || nrhs > 64) return fail(DCX_ERR_INVALID, "solve: 1 <= n <= 4096, 1 <= nrhs <= 64");
    if (!A || !B || !X || !work || !info) return fail(DCX_ERR_INVALID, "a solve pointer is NULL");
    if (work_bytes < solve_work_bytes(n, nrhs) || (reinterpret_cast<uintptr_t>(work) & 15))
        return fail(DCX_ERR_INVALID, "solve: the workspace is smaller than dcx_solve_work_bytes(n, nrhs) or not 16-byte aligned");
    if (int rc = set_device(device)) return rc;
    static int n_cu_of[64] = {0};
    int n_cu = (device >= 0 && device < 64) ? n_cu_of[device] : 0;
    if (n_cu == 0) {
        hipDeviceProp_t prop;
        n_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 64;
        if (device >= 0 && device < 64) n_cu_of[device] = n_cu;
    }
    hipError_t e = launch_solve(A, B, X, (int)n, (int)nrhs, work, info, n_cu, (flags & DCX_SOLVE_ONE_WORKGROUP) != 0,
                                knobs().solve_threads == 256 ? 256 : knobs().solve_threads == 512 ? 512 : 0, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "solve launch");
    return DCX_OK;
}

}  // extern "C"
