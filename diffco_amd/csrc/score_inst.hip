// score_inst.hip — instantiates the fused sweep for ONE feature width (-DDCX_INST_D=<D>) and
// every (kernel function, class count, gradient mode) combination, behind a plain dispatcher.
// Built once per width so the widths compile in parallel (see Makefile).
#include "dcx_internal.h"
#include "traj_fused.h"
#include "jac_kernel.h"

#ifndef DCX_INST_D
#error "compile with -DDCX_INST_D=<feature width>"
#endif

#if (defined(DCX_STUB) && DCX_INST_PART != 0) || (DCX_INST_PART == 2 && DCX_INST_D > 24)
// (the stubs of this width live in its part-0 object; part 2 - the multi-class trajectory kernels - exists up to D = 24)
#elif defined(DCX_STUB)  // developer builds (Makefile ONLY_WIDTHS): this width is not compiled
namespace dcx {
#define DCX_CAT_(a, b) a##b
#define DCX_CAT(a, b) DCX_CAT_(a, b)
hipError_t DCX_CAT(launch_score_D, DCX_INST_D)(int, int, int, int, size_t, int64_t, const ScoreArgs&, hipStream_t) { return hipErrorNotSupported; }
hipError_t DCX_CAT(launch_jac_D, DCX_INST_D)(int, int, int, size_t, int64_t, const ScoreArgs&, hipStream_t) { return hipErrorNotSupported; }
hipError_t DCX_CAT(launch_traj_fused_D, DCX_INST_D)(int, int, int, size_t, int, const TrajFusedArgs&, hipStream_t) { return hipErrorNotSupported; }
}  // namespace dcx
#else
namespace dcx {
namespace {

constexpr int kD = DCX_INST_D;
constexpr int kMaxT = kD <= 16 ? 1024 : (kD <= 48 ? 512 : 256);

// widths / modes that carry the MFMA form of the gradient fold (sweep_rows_mfma)
// (one, five and eight classes: the developer knob takes it for those models, dcx_api.hip; with C > 1 the weight
// contraction K . W runs on the matrix cores beside the gradient fold)
// Round 5: NOT in the shipped library.  Both matrix-core forms (MF: the gradient fold / K . W on v_mfma_f32_16x16x4_f32; XM: the
// distance GEMM as bf16 x 3 on v_mfma_f32_16x16x32_bf16) measured 4 - 47 % slower than the VALU forms (profiles/r03_mfma_ab.txt);
// they are compiled by `make EXTRA=-DDCX_WITH_MATRIX_FORMS` (tools/build_variant.sh matrix ...) for the parity tests' "mfma" leg
// and the A/B tools, and dcx_debug_set("mfma" / "xm", 1) answers DCX_ERR_UNSUPPORTED without them.
#ifdef DCX_WITH_MATRIX_FORMS
template <int KF, int CC, int MODE>
constexpr bool kHasMfma = (kD == 12 || kD == 16) && (CC == 1 || CC == 5 || CC == 8) && (MODE != MODE_SCORE) && (KF != KF_GEN);
constexpr bool kHasXm = true;
#else
template <int KF, int CC, int MODE>
constexpr bool kHasMfma = false;
constexpr bool kHasXm = false;
#endif

// Shapes with an expanded form (score_kernel.h, XF: Polyharmonic(1), rows of <= 37 floats) run it by default, on the
// centred row pairs the host passes with it (ScoreArgs::centre).  The direct form is compiled for them as well and
// selected per launch by ScoreArgs::xf (dcx_debug_set("xf", 0)): the parity tests run every case in both forms,
// tools/xf_probe.py times them side by side.

template <int KF, int CC, int MODE>
hipError_t go(int nw, size_t lds, int64_t nblk, const ScoreArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)nblk, (unsigned)(a.ys > 1 ? a.ys : 1), (unsigned)(a.nz > 1 ? a.nz : 1));
    // more than 64 KB of dynamic LDS (a split launch's 16 partial rows of a C > 1 model) needs the attribute raised once
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        kern<<<grid, dim3(64 * nw), lds, st>>>(a);
        return hipGetLastError();
    };
    if constexpr (qt_applies(kD, CC, KF, MODE)) {   // the quarter tile (16 configurations per block, rows from LDS): nblk counts ITS blocks
        if (a.qt) return launch(score_kernel<kD, KF, CC, MODE, kMaxT, false, false, false, true>);
    }
    if constexpr (kHasXm && xf_applies(kD, CC, KF) && xm_applies(kD, CC, KF) && MODE == MODE_GRAD_ROW) {
        if (!a.mfma && a.xf && a.xm) return launch(score_kernel<kD, KF, CC, MODE, kMaxT, false, true, true>);
    }
    if constexpr (xf_applies(kD, CC, KF)) {
        if (!a.mfma && a.xf) return launch(score_kernel<kD, KF, CC, MODE, kMaxT, false, true>);
    }
    if constexpr (kHasMfma<KF, CC, MODE>) {
        if (a.mfma) return launch(score_kernel<kD, KF, CC, MODE, kMaxT, true>);
    }
    return launch(score_kernel<kD, KF, CC, MODE, kMaxT>);
}

template <int KF, int CC>
hipError_t by_mode(int mode, int nw, size_t lds, int64_t nblk, const ScoreArgs& a, hipStream_t st) {
    switch (mode) {
    case MODE_SCORE: return go<KF, CC, MODE_SCORE>(nw, lds, nblk, a, st);
    case MODE_GRAD_ROW: return go<KF, CC, MODE_GRAD_ROW>(nw, lds, nblk, a, st);
    case MODE_GRAD_UP:
        if constexpr (CC > 1) return go<KF, CC, MODE_GRAD_UP>(nw, lds, nblk, a, st);
    default: return hipErrorInvalidValue;
    }
}

template <int KF>
hipError_t by_cc(int cc, int mode, int nw, size_t lds, int64_t nblk, const ScoreArgs& a, hipStream_t st) {
    switch (cc) {
    case 1: return by_mode<KF, 1>(mode, nw, lds, nblk, a, st);
    case 5: return by_mode<KF, 5>(mode, nw, lds, nblk, a, st);
#ifndef DCX_DEV_FAST  // developer builds (EXTRA=-DDCX_DEV_FAST): one and five classes only
    case 4: return by_mode<KF, 4>(mode, nw, lds, nblk, a, st);   // (2 and 3 run as 4, 6 and 7 as 8: dcx_internal.h compiled_classes)
    case 8: return by_mode<KF, 8>(mode, nw, lds, nblk, a, st);
#endif
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

#define DCX_CAT_(a, b) a##b
#define DCX_CAT(a, b) DCX_CAT_(a, b)
// Every width is built as TWO objects (Makefile, -DDCX_INST_PART=0 / 1; round 5: 21 objects of up to 155 s each left a
// 16-core build two rounds of the longest ones): part 0 = the sweep with the two specialised kernel functions, part 1 = the
// generic kernel function, the one-sweep Jacobian and the persistent trajectory kernel; round 6: part 2 (widths up to 24) = the
// persistent trajectory kernel for several classes.
#ifndef DCX_INST_PART
#error "compile with -DDCX_INST_PART=0, 1 and (D <= 24) 2"
#endif
hipError_t DCX_CAT(launch_score_gen_D, DCX_INST_D)(int cc, int mode, int nw, size_t lds, int64_t nblk, const ScoreArgs& a, hipStream_t st);
namespace {
// the persistent trajectory kernel's launch: plain, or cooperative for the cluster form
template <class K>
hipError_t traj_go(K kern, int nw, size_t lds, int n_paths, const TrajFusedArgs& a, hipStream_t st) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (a.ys > 1) {
        // the cluster form: the ys workgroups of a path wait for each other once per iteration, so the whole grid must be
        // resident at once - a cooperative launch checks that and keeps other cooperative grids off the device meanwhile
        TrajFusedArgs copy = a;
        void* params[] = {&copy};
        const dim3 grid = a.cl_across ? dim3((unsigned)a.ys, (unsigned)n_paths) : dim3((unsigned)n_paths, (unsigned)a.ys);
        return hipLaunchCooperativeKernel((const void*)kern, grid, dim3(64 * nw), params, (unsigned int)lds, st);
    }
    kern<<<dim3((unsigned)n_paths), dim3(64 * nw), lds, st>>>(a);
    return hipGetLastError();
}

}  // namespace
#if DCX_INST_PART == 0
hipError_t DCX_CAT(launch_score_D, DCX_INST_D)(int kf, int cc, int mode, int nw, size_t lds, int64_t nblk,
                                               const ScoreArgs& a, hipStream_t st) {
    switch (kf) {
    case KF_RQ2: return by_cc<KF_RQ2>(cc, mode, nw, lds, nblk, a, st);
    case KF_POLY1: return by_cc<KF_POLY1>(cc, mode, nw, lds, nblk, a, st);
    case KF_GEN: return DCX_CAT(launch_score_gen_D, DCX_INST_D)(cc, mode, nw, lds, nblk, a, st);
    default: return hipErrorInvalidValue;
    }
}
#elif DCX_INST_PART == 1
hipError_t DCX_CAT(launch_score_gen_D, DCX_INST_D)(int cc, int mode, int nw, size_t lds, int64_t nblk, const ScoreArgs& a, hipStream_t st) {
    return by_cc<KF_GEN>(cc, mode, nw, lds, nblk, a, st);
}

namespace {
// the one-sweep Jacobian (jac_kernel.h); hipErrorNotSupported when this width / class count has no instantiation
template <int KF, int CC>
hipError_t jac_go(int nw, size_t lds, int64_t nblk, const ScoreArgs& a, hipStream_t st) {
    if constexpr (jac_applies(kD, CC)) {
        auto kern = score_jac_kernel<kD, KF, CC, kMaxT>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        kern<<<dim3((unsigned)nblk), dim3(64 * nw), lds, st>>>(a);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}
template <int KF>
hipError_t jac_by_cc(int cc, int nw, size_t lds, int64_t nblk, const ScoreArgs& a, hipStream_t st) {
    switch (cc) {
    case 5: return jac_go<KF, 5>(nw, lds, nblk, a, st);
#ifndef DCX_DEV_FAST
    case 4: return jac_go<KF, 4>(nw, lds, nblk, a, st);
    case 8: return jac_go<KF, 8>(nw, lds, nblk, a, st);
#endif
    default: return hipErrorNotSupported;
    }
}
}  // namespace

hipError_t DCX_CAT(launch_jac_D, DCX_INST_D)(int kf, int cc, int nw, size_t lds, int64_t nblk, const ScoreArgs& a,
                                             hipStream_t st) {
    switch (kf) {
    case KF_RQ2: return jac_by_cc<KF_RQ2>(cc, nw, lds, nblk, a, st);
    case KF_POLY1: return jac_by_cc<KF_POLY1>(cc, nw, lds, nblk, a, st);
    case KF_GEN: return jac_by_cc<KF_GEN>(cc, nw, lds, nblk, a, st);
    default: return hipErrorInvalidValue;
    }
}

#if DCX_INST_PART == 1
// several classes (part 2 of this width, D <= 24): two sweeps per iteration, see traj_fused.h
#if DCX_INST_D <= 24
hipError_t DCX_CAT(launch_traj_fused_mc_D, DCX_INST_D)(int kf, int cc, int nw, size_t lds, int n_paths, const TrajFusedArgs& a, hipStream_t st);
#endif
hipError_t DCX_CAT(launch_traj_fused_D, DCX_INST_D)(int kf, int cc, int nw, size_t lds, int n_paths, const TrajFusedArgs& a,
                                                    hipStream_t st) {
    if (cc > 1) {
#if DCX_INST_D <= 24
        return DCX_CAT(launch_traj_fused_mc_D, DCX_INST_D)(kf, cc, nw, lds, n_paths, a, st);
#else
        return hipErrorNotSupported;   // (the caller runs the three-launch loop)
#endif
    }
    auto go_t = [&](auto kern) { return traj_go(kern, nw, lds, n_paths, a, st); };
    if (a.ys > 1) {
        if constexpr (xf_applies(kD, 1, KF_POLY1)) {
            if (kf == KF_POLY1 && a.sc.xf) return go_t(traj_fused_kernel<kD, KF_POLY1, kMaxT, true, true>);
        }
        switch (kf) {
        case KF_RQ2: return go_t(traj_fused_kernel<kD, KF_RQ2, kMaxT, false, true>);
        case KF_POLY1: return go_t(traj_fused_kernel<kD, KF_POLY1, kMaxT, false, true>);
        case KF_GEN: return go_t(traj_fused_kernel<kD, KF_GEN, kMaxT, false, true>);
        default: return hipErrorInvalidValue;
        }
    }
    if constexpr (xf_applies(kD, 1, KF_POLY1)) {
        if (kf == KF_POLY1 && a.sc.xf) return go_t(traj_fused_kernel<kD, KF_POLY1, kMaxT, true>);
    }
    switch (kf) {
    case KF_RQ2: return go_t(traj_fused_kernel<kD, KF_RQ2, kMaxT>);
    case KF_POLY1: return go_t(traj_fused_kernel<kD, KF_POLY1, kMaxT>);
    case KF_GEN: return go_t(traj_fused_kernel<kD, KF_GEN, kMaxT>);
    default: return hipErrorInvalidValue;
    }
}
#endif  // part 1

#else   // ---- part 2: the persistent trajectory kernel for several classes (D <= 24; RQKernel(p = 2) and Polyharmonic(1)) ----
namespace {
template <int CC>
hipError_t traj_mc(int kf, int nw, size_t lds, int n_paths, const TrajFusedArgs& a, hipStream_t st) {
    auto go_t = [&](auto kern) { return traj_go(kern, nw, lds, n_paths, a, st); };
    if (a.ys > 1) {
        if constexpr (xf_applies(kD, CC, KF_POLY1)) {
            if (kf == KF_POLY1 && a.sc.xf) return go_t(traj_fused_kernel<kD, KF_POLY1, kMaxT, true, true, CC>);
        }
        switch (kf) {
        case KF_RQ2: return go_t(traj_fused_kernel<kD, KF_RQ2, kMaxT, false, true, CC>);
        case KF_POLY1: return go_t(traj_fused_kernel<kD, KF_POLY1, kMaxT, false, true, CC>);
        default: return hipErrorNotSupported;   // (any other kernel function: the three-launch loop)
        }
    }
    if constexpr (xf_applies(kD, CC, KF_POLY1)) {
        if (kf == KF_POLY1 && a.sc.xf) return go_t(traj_fused_kernel<kD, KF_POLY1, kMaxT, true, false, CC>);
    }
    switch (kf) {
    case KF_RQ2: return go_t(traj_fused_kernel<kD, KF_RQ2, kMaxT, false, false, CC>);
    case KF_POLY1: return go_t(traj_fused_kernel<kD, KF_POLY1, kMaxT, false, false, CC>);
    default: return hipErrorNotSupported;
    }
}
}  // namespace
hipError_t DCX_CAT(launch_traj_fused_mc_D, DCX_INST_D)(int kf, int cc, int nw, size_t lds, int n_paths, const TrajFusedArgs& a, hipStream_t st) {
    switch (cc) {
    case 5: return traj_mc<5>(kf, nw, lds, n_paths, a, st);
#ifndef DCX_DEV_FAST
    case 4: return traj_mc<4>(kf, nw, lds, n_paths, a, st);
    case 8: return traj_mc<8>(kf, nw, lds, n_paths, a, st);
#endif
    default: return hipErrorNotSupported;
    }
}
#endif  // DCX_INST_PART
}  // namespace dcx
#endif  // DCX_STUB
