// dcx_internal.h — host-side types shared by the translation units of libdcx.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dcx.h"
#include "score_kernel.h"
#include "traj_fused.h"
#include "jac_kernel.h"

namespace dcx {

// feature widths the sweep is compiled for; any D <= DCX_MAX_D is zero-padded up to the next one
static constexpr int kTemplateD[] = {2, 4, 6, 8, 12, 16, 18, 21, 24, 27, 30, 32, 36, 42, 48, 54, 60, 64, 72, 84, 96};
static constexpr int kNumTemplateD = sizeof(kTemplateD) / sizeof(int);

inline int template_d_for(int D) {
    for (int i = 0; i < kNumTemplateD; ++i)
        if (D <= kTemplateD[i]) return kTemplateD[i];
    return -1;
}
// block-size ceiling per compiled width: keeps the VGPR budget (512 / waves-per-SIMD) above
// the ~3*D live registers of the sweep
inline int max_threads_for(int Dt) { return Dt <= 16 ? 1024 : (Dt <= 48 ? 512 : 256); }

// class counts the sweeps are compiled for: 2 and 3 run as 4, 6 and 7 as 8 - the extra classes are zero weight columns in the
// rows, never read from `upstream` and never written to `score` (ScoreArgs::c_out).  Five is BASELINE config #3's count.
// (Round 4: eight compiled class counts x 21 widths x 3 kernel functions x modes x forms had grown to 115 MB and 10 minutes;
// round 6: two classes through the four-class bodies as well - one instantiation in five less to build, two idle fma per pair.)
inline int compiled_classes(int C) { return C <= 1 ? C : C <= 4 ? 4 : C == 5 ? 5 : 8; }

inline int row_stride(int Dt, int C) {   // RowLayout<Dt, C>::RS
    const int al = C > 1 ? DCX_ROW_ALIGN_MULTI : 4;
    return (Dt + C + (C > 1 ? 1 : 0) + 1 + al - 1) / al * al;
}

// one entry point per compiled D (score_inst.hip, built once per width)
typedef hipError_t (*launch_fn)(int kf, int cc, int mode, int nw, size_t lds_bytes, int64_t n_blocks,
                                const ScoreArgs& args, hipStream_t stream);
#define DCX_DECLARE_LAUNCH(D) \
    hipError_t launch_score_D##D(int, int, int, int, size_t, int64_t, const ScoreArgs&, hipStream_t);
DCX_DECLARE_LAUNCH(2)  DCX_DECLARE_LAUNCH(4)  DCX_DECLARE_LAUNCH(6)  DCX_DECLARE_LAUNCH(8)
DCX_DECLARE_LAUNCH(12) DCX_DECLARE_LAUNCH(16) DCX_DECLARE_LAUNCH(18) DCX_DECLARE_LAUNCH(21)
DCX_DECLARE_LAUNCH(24) DCX_DECLARE_LAUNCH(27) DCX_DECLARE_LAUNCH(30) DCX_DECLARE_LAUNCH(32)
DCX_DECLARE_LAUNCH(36) DCX_DECLARE_LAUNCH(42) DCX_DECLARE_LAUNCH(48) DCX_DECLARE_LAUNCH(54)
DCX_DECLARE_LAUNCH(60) DCX_DECLARE_LAUNCH(64) DCX_DECLARE_LAUNCH(72) DCX_DECLARE_LAUNCH(84)
DCX_DECLARE_LAUNCH(96)
#undef DCX_DECLARE_LAUNCH
// the one-sweep Jacobian (jac_kernel.h), one entry point per compiled D; hipErrorNotSupported where it is not instantiated
typedef hipError_t (*jac_fn)(int kf, int cc, int nw, size_t lds_bytes, int64_t n_blocks, const ScoreArgs& args, hipStream_t stream);
#define DCX_DECLARE_JAC(D) hipError_t launch_jac_D##D(int, int, int, size_t, int64_t, const ScoreArgs&, hipStream_t);
DCX_DECLARE_JAC(2)  DCX_DECLARE_JAC(4)  DCX_DECLARE_JAC(6)  DCX_DECLARE_JAC(8)
DCX_DECLARE_JAC(12) DCX_DECLARE_JAC(16) DCX_DECLARE_JAC(18) DCX_DECLARE_JAC(21)
DCX_DECLARE_JAC(24) DCX_DECLARE_JAC(27) DCX_DECLARE_JAC(30) DCX_DECLARE_JAC(32)
DCX_DECLARE_JAC(36) DCX_DECLARE_JAC(42) DCX_DECLARE_JAC(48) DCX_DECLARE_JAC(54)
DCX_DECLARE_JAC(60) DCX_DECLARE_JAC(64) DCX_DECLARE_JAC(72) DCX_DECLARE_JAC(84)
DCX_DECLARE_JAC(96)
#undef DCX_DECLARE_JAC
// config #5 as one persistent launch (traj_fused.h), one entry point per compiled D as well
// (cc > 1: the two-sweep form for several classes, compiled up to D = 24 for RQKernel(p = 2) / Polyharmonic(1); hipErrorNotSupported elsewhere)
typedef hipError_t (*traj_fused_fn)(int kf, int cc, int nw, size_t lds_bytes, int n_paths, const TrajFusedArgs& args, hipStream_t stream);
#define DCX_DECLARE_TRAJ(D) hipError_t launch_traj_fused_D##D(int, int, int, size_t, int, const TrajFusedArgs&, hipStream_t);
DCX_DECLARE_TRAJ(2)  DCX_DECLARE_TRAJ(4)  DCX_DECLARE_TRAJ(6)  DCX_DECLARE_TRAJ(8)
DCX_DECLARE_TRAJ(12) DCX_DECLARE_TRAJ(16) DCX_DECLARE_TRAJ(18) DCX_DECLARE_TRAJ(21)
DCX_DECLARE_TRAJ(24) DCX_DECLARE_TRAJ(27) DCX_DECLARE_TRAJ(30) DCX_DECLARE_TRAJ(32)
DCX_DECLARE_TRAJ(36) DCX_DECLARE_TRAJ(42) DCX_DECLARE_TRAJ(48) DCX_DECLARE_TRAJ(54)
DCX_DECLARE_TRAJ(60) DCX_DECLARE_TRAJ(64) DCX_DECLARE_TRAJ(72) DCX_DECLARE_TRAJ(84)
DCX_DECLARE_TRAJ(96)
#undef DCX_DECLARE_TRAJ

// aux_kernels.hip
hipError_t launch_score_finish(const FinishArgs& a, int64_t n_tiles, size_t lds_bytes, hipStream_t stream);
hipError_t launch_fkine(const FkProg* fk_dev, const dcx_fk_desc& fk_host, const float* q, int64_t B, float* X,
                        hipStream_t stream);
hipError_t launch_fkine_vjp(const FkProg* fk_dev, const dcx_fk_desc& fk_host, const float* q, const float* gX,
                            int64_t B, float* gq, hipStream_t stream);
hipError_t launch_clock_probe(unsigned long long* out, unsigned long long wall_ticks, hipStream_t stream);
// utils.DH2mat / utils.euler2mat (utils.py:66-75, 15-38) and their autograd with respect to the angles
hipError_t launch_dh_frames(const float* q, int64_t B, int dof, const float* a, const float* d, const float* sa, const float* ca,
                            float* T, hipStream_t stream);
hipError_t launch_dh_frames_vjp(const float* q, int64_t B, int dof, const float* a, const float* sa, const float* ca, const float* gT,
                                float* gq, hipStream_t stream);
hipError_t launch_euler_frames(const float* phi, int64_t B, float* R, hipStream_t stream);
hipError_t launch_euler_frames_vjp(const float* phi, const float* gR, int64_t B, float* gphi, hipStream_t stream);
hipError_t launch_kernel_matrix(int kind, float kp0, float kp1, const float* x, int64_t B, const float* s, int64_t S,
                                int D, float* K, hipStream_t stream);

// hess_kernel.hip: what the second-derivative kernel needs to know of a model
struct ModelView {
    const float* rows;
    const FkProg* fk;
    int32_t S, Dt, C, RS, dof, d_fk, frame_floats, prog_floats, kind, kf;   // C: the compiled class count (row layout)
    int32_t c_out = 0;      // the caller's class count
    float kp0, kp1;
    // this stream's split-launch scratch (dcx_api.hip split_scratch) or null: partial rows + zeroed arrival counters
    float* scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned int* counters = nullptr;
    int32_t n_counters = 0, counter_stride = 0, n_cu = 0;
    int32_t fk_dh = 0;      // the transform is a DH arm
    int32_t ys_knob = -1;   // developer knob hess_ys: blocks per tile (1 = never split)
    int32_t form_knob = -1; // developer knob hess_form: 1 = the moments form wherever it is compiled, 0 = never
    float* mom = nullptr;   // this stream's buffer for the moments form's sums (hess_moments_bytes), or null: the other form runs
    size_t mom_bytes = 0;
};
// the moments form of the second-derivative kernel (hess_kernel.hip): widths it is compiled for, the batches it takes, its buffer
constexpr int kHessMomentRows = 512;        // (tile, y) rows per chunk: 32768 configurations, or fewer tiles split over the supports
constexpr int64_t kHessMomentsMinB = 1024;  // smaller batches are launch latency either way: one launch instead of two (measured equal at 1024)
constexpr bool hess_moments_width(int D) { return D == 2 || D == 4 || D == 6 || D == 8 || D == 12 || D == 16; }
constexpr int hess_moments_nacc(int D) { return D + 1 + (D / 2) * (D / 2 + 1) * 2; }
inline size_t hess_moments_bytes(int D) { return (size_t)kHessMomentRows * hess_moments_nacc(D) * 64 * sizeof(float); }
inline bool hess_moments_applies(const ModelView& m, int64_t B) {
    return hess_moments_width(m.Dt) && (m.form_knob == 1 || (m.form_knob < 0 && B >= kHessMomentsMinB));
}
hipError_t launch_hess(const ModelView& m, const float* q, int64_t B, const float* upstream, float* grad, float* hess,
                       hipStream_t stream);

// train_kernels.hip
hipError_t launch_perceptron(int kind, float kp0, float kp1, float beta, const float* feats, const float* y, float* gains,
                             float* hypo, float* K, int32_t* info, int N, int D, int C, int max_iter, bool sign_labels,
                             bool grid, hipStream_t st);
constexpr int kTrainGridMaxN = 128 * 1024;  // 128 workgroups x 1024 samples (train_kernels.hip perceptron_grid_kernel)

// traj_kernels.hip
// (n_class, margin_c: col_score is [R*W, n_class] under per-class margins; margin_c == nullptr: opt.safety_margin for every class)
hipError_t launch_traj_adam_step(const FkProg* fk_dev, const dcx_fk_desc& fk, const dcx_traj_state& st,
                                 const dcx_traj_opts& opt, int step, hipStream_t stream, int n_class = 1, const float* margin_c = nullptr);
// the update half of dcx_escape_adam: pointers into the caller's workspace, one configuration per lane
struct EscapeArgs {
    float* q;             // [B, dof] in/out
    const float* score;   // [B, C]   this step's dist_est(q)
    const float* grad;    // [B, dof] this step's d sum_c score_c / d q
    const float* margin;  // [C] or nullptr
    float *adam_m, *adam_v;
    float* history;       // [n_slots, B, dof] or nullptr
    int32_t* steps;       // [n_loops, 2]: evaluations, Adam steps (traj_kernels.hip)
    int64_t B;
    int32_t dof, C, step, record_freq, joint;
    int32_t last;         // this is the call's last step (the loops still running leave their final configuration behind it)
    uint64_t wrap_mask;
    float lr, beta1, beta2, eps, bias1, bias2_sqrt;
    // after a compaction the sweep's batch is the n_act loops still running: score / grad rows i, configuration idx[i] of q,
    // its current value also in qa[i] (what the sweep reads).  Before the first one: idx = qa = nullptr, n_act = B.
    const int32_t* idx;
    float* qa;
    int64_t n_act;
};
hipError_t launch_escape_step(EscapeArgs a, int step, bool last, hipStream_t stream);   // step is 0-based
hipError_t launch_escape_compact(const EscapeArgs& a, const int32_t* idx_in, int64_t n_in, int32_t* idx_out, float* qa_out,
                                 int32_t* count, hipStream_t stream);

}  // namespace dcx
