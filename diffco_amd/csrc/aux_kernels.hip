// aux_kernels.hip — the pieces of the path that are also callable on their own:
//   fkine / fkine_vjp  : model.*.fkine and its autograd   (reference model.py:40-48, 90-93, 156-159,
//                        225-241, 366-383, 430-453, 486-502)
//   kernel_matrix      : KernelFunc.__call__ -> K[B,S]     (reference kernel.py:17-29, 49-57, 73-79),
//                        used by the perceptron trainer's row fill and by fit_poly.
#include "dcx_internal.h"

namespace dcx {
namespace {

// one wave per block, one lane per configuration; same LDS staging as the fused kernel
__global__ __launch_bounds__(64) void fkine_kernel(const FkProg* fkd, const float* q, int64_t B, float* X,
                                                   int dof, int d_fk, int frame_floats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int nb = (int)((B - b0) < 64 ? (B - b0) : 64);
    float* sQ = smem;
    float* sX = sQ + ((64 * dof + 3) & ~3);
    float* sF = sX + 64 * d_fk;
    const fk_cptr fk = stage_fk_prog(fkd, sF + 64 * frame_floats, lane, 64);  // program last (variable size)
    const float* qsrc = q + b0 * dof;
    const int n = nb * dof;
    for (int i = lane; i < 64 * dof; i += 64) sQ[i] = qsrc[i < n ? i : (i % dof) + (nb - 1) * dof];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    fk_forward_trig(fk, sQ + lane * dof, sF + lane, 0, 1);
    fk_forward_chain(fk, sQ + lane * dof, sX + lane, sF + lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // [64][d_fk] rows out, coalesced
    float* dst = X + b0 * d_fk;
    const int m = nb * d_fk;
    for (int i = lane; i < m; i += 64) dst[i] = sX[(i % d_fk) * 64 + (i / d_fk)];
}

__global__ __launch_bounds__(64) void fkine_vjp_kernel(const FkProg* fkd, const float* q, const float* gX,
                                                       int64_t B, float* gq, int dof, int d_fk, int frame_floats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int nb = (int)((B - b0) < 64 ? (B - b0) : 64);
    float* sQ = smem;
    float* sX = sQ + ((64 * dof + 3) & ~3);
    float* sG = sX + 64 * d_fk;
    float* sF = sG + 64 * d_fk;
    const fk_cptr fk = stage_fk_prog(fkd, sF + 64 * frame_floats, lane, 64);  // program last (variable size)
    const float* qsrc = q + b0 * dof;
    const int n = nb * dof;
    for (int i = lane; i < 64 * dof; i += 64) sQ[i] = qsrc[i < n ? i : (i % dof) + (nb - 1) * dof];
    const float* gsrc = gX + b0 * d_fk;
    const int m = nb * d_fk;
    for (int i = lane; i < 64 * d_fk; i += 64) {
        const int ii = i < m ? i : (i % d_fk) + (nb - 1) * d_fk;
        sG[(i % d_fk) * 64 + (i / d_fk)] = gsrc[ii];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    fk_forward_trig(fk, sQ + lane * dof, sF + lane, 0, 1);
    fk_forward_chain(fk, sQ + lane * dof, sX + lane, sF + lane);
    fk_vjp(fk, sQ + lane * dof, sF + lane, sG + lane, sQ + lane * dof);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float* dst = gq + b0 * dof;
    for (int i = lane; i < n; i += 64) dst[i] = sQ[i];
}

// K[b, j]: thread = one support column j, loop over a strip of TB configurations.
// HBM-write bound (4 B per pair); supports are re-read through L1/L2.
constexpr int KM_TB = 16;
__global__ __launch_bounds__(256) void kernel_matrix_kernel(ScoreArgs a, const float* x, int64_t B, const float* s,
                                                            int64_t S, int D, float* K) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.y * KM_TB;
    if (j >= S) return;
    const float* sj = s + j * D;
    for (int t = 0; t < KM_TB; ++t) {
        const int64_t b = b0 + t;
        if (b >= B) break;
        const float* xb = x + b * D;
        float d2 = 0.f;
        for (int k = 0; k < D; ++k) {
            const float dl = xb[k] - sj[k];
            d2 = fmaf(dl, dl, d2);
        }
        float val, g;
        if (a.kind == DCX_K_RQ && a.kp1 == 2.0f) {
            kernel_eval<KF_RQ2>(d2, a, val, g);
        } else if (a.kind == DCX_K_POLY && a.kp0 == 1.0f) {
            // exact zero at coincident points, 1/eps applied here (nothing to fold it into)
            val = (d2 > 0.f ? d2 * __builtin_amdgcn_rsqf(d2) : 0.f) / a.kp1;
        } else {
            kernel_eval<KF_GEN>(d2, a, val, g);
            if (a.kind == DCX_K_POLY && d2 == 0.f) val = 0.f;
        }
        K[b * S + j] = val;
    }
}

__global__ __launch_bounds__(64) void score_finish_kernel(const FinishArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int nb = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
    const int dof = a.dof;
    const LdsPlan lp = lds_plan(dof, a.d_fk, a.frame_floats, 0, 0);
    float* sQ = smem + lp.q;
    float* sX = smem + lp.x;
    float* sG = smem + lp.g;
    float* sF = smem + lp.f;
    const float* part = a.partial + (size_t)blockIdx.x * a.ys * a.acc * 64 + lane;
    // v[u] = rows y = 0, 1, ... of accumulators e0 .. e0+n-1 added in that order; 8 accumulators x 2 rows of loads in
    // flight at a time (one load per round trip made this kernel latency-bound: ys * acc dependent L2 reads)
    constexpr int EC = 8;
    auto fold = [&](int e0, int n, float* v) {
#pragma unroll
        for (int u = 0; u < EC; ++u) v[u] = 0.0f;
        int y = 0;
        for (; y + 1 < a.ys; y += 2) {
            float r0[EC], r1[EC];
#pragma unroll
            for (int u = 0; u < EC; ++u) {
                const int e = e0 + (u < n ? u : 0);
                r0[u] = part[((size_t)y * a.acc + e) * 64];
                r1[u] = part[((size_t)(y + 1) * a.acc + e) * 64];
            }
#pragma unroll
            for (int u = 0; u < EC; ++u) v[u] = (v[u] + r0[u]) + r1[u];
        }
        if (y < a.ys) {
#pragma unroll
            for (int u = 0; u < EC; ++u) v[u] += part[((size_t)y * a.acc + e0 + (u < n ? u : 0)) * 64];
        }
    };
    float score0 = 0.0f;
    for (int c0 = 0; c0 < a.C; c0 += EC) {
        float v[EC];
        const int n = (a.C - c0) < EC ? (a.C - c0) : EC;
        fold(c0, n, v);
        if (c0 == 0) score0 = v[0];
#pragma unroll
        for (int u = 0; u < EC; ++u)
            if (u < n && c0 + u < a.c_out && a.score != nullptr && lane < nb) a.score[(b0 + lane) * a.c_out + c0 + u] = v[u];
    }
    if (!a.want_grad) return;
    const fk_cptr fk = stage_fk_prog(a.fk, smem + lp.fk, lane, 64);
    {
        const float* qsrc = a.q + b0 * dof;
        const int n = nb * dof;
        for (int i = lane; i < 64 * dof; i += 64) sQ[i] = qsrc[i < n ? i : (i % dof) + (nb - 1) * dof];
    }
    float scale = 1.0f;
    if (a.C == 1 && a.upstream != nullptr) scale = a.upstream[b0 + (lane < nb ? lane : nb - 1)];
    if (a.C == 1 && a.hinge) scale = (score0 - a.hinge_margin > 0.0f) ? a.hinge_weight : 0.0f;
    for (int k0 = 0; k0 < a.d_fk; k0 += EC) {
        float v[EC];
        const int n = (a.d_fk - k0) < EC ? (a.d_fk - k0) : EC;
        fold(a.C + k0, n, v);
#pragma unroll
        for (int u = 0; u < EC; ++u)
            if (u < n) sG[(k0 + u) * 64 + lane] = v[u] * scale;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    fk_forward_trig(fk, sQ + lane * dof, sF + lane, 0, 1);
    fk_forward_chain(fk, sQ + lane * dof, sX + lane, sF + lane);
    fk_vjp(fk, sQ + lane * dof, sF + lane, sG + lane, sQ + lane * dof);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float* gdst = a.grad + b0 * a.grad_stride;
    const int n = nb * dof;
    if (a.grad_stride == dof) {
        for (int i = lane; i < n; i += 64) gdst[i] = sQ[i];
    } else {
        for (int i = lane; i < n; i += 64) gdst[(int64_t)(i / dof) * a.grad_stride + (i % dof)] = sQ[i];
    }
}


size_t fk_lds_bytes(const dcx_fk_desc& fk, bool with_g) {
    const int d_fk = fk.n_points * fk.point_dim;
    return sizeof(float) * (fk_prog_floats(fk) + ((64 * fk.dof + 3) & ~3) + 64 * d_fk * (with_g ? 2 : 1) + 64 * fk_frame_floats(fk));
}

// ---- utils.DH2mat / utils.euler2mat as callables of their own (reference utils.py:66-75, 15-38; round 6) ---------------------
// User-written robot classes build their FK from these (model.py:230, 437 do: DH2mat -> a chain of bmm); the fused kernels
// never materialise link frames, so these exist for that API only.  HBM-bound by construction: 4 B in, 64 B out per joint.
// One lane = one (configuration, joint) entry; the frames leave through LDS as whole 16-byte pieces, coalesced.
typedef float v4f_aux __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void dh_frames_kernel(const float* q, int64_t n, int dof, const float* a, const float* d,
                                                        const float* sa, const float* ca, float* T) {
    __shared__ v4f_aux sT[256 * 4];
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * 256, e = e0 + tid;
    if (e < n) {
        const int j = (int)(e % dof);
        float s, c;
        sincos_f32(q[e], &s, &c);
        const float aj = a[j], saj = sa[j], caj = ca[j];
        // T = Rz(theta) Tz(d) Tx(a) Rx(alpha)  (utils.py:69-74, row by row)
        sT[tid * 4 + 0] = v4f_aux{c, -s * caj, s * saj, aj * c};
        sT[tid * 4 + 1] = v4f_aux{s, c * caj, -c * saj, aj * s};
        sT[tid * 4 + 2] = v4f_aux{0.0f, saj, caj, d[j]};
        sT[tid * 4 + 3] = v4f_aux{0.0f, 0.0f, 0.0f, 1.0f};
    }
    __syncthreads();
    const int64_t left = n - e0;
    const int rows = (int)(left < 256 ? left : 256) * 4;
    v4f_aux* dst = reinterpret_cast<v4f_aux*>(T + e0 * 16);
    for (int i = tid; i < rows; i += 256) dst[i] = sT[i];
}
// autograd of the above w.r.t. the joint angles: only rows 0 and 1 of a frame depend on theta
__global__ __launch_bounds__(256) void dh_frames_vjp_kernel(const float* q, int64_t n, int dof, const float* a, const float* sa,
                                                            const float* ca, const float* gT, float* gq) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int j = (int)(e % dof);
    float s, c;
    sincos_f32(q[e], &s, &c);
    const v4f_aux g0 = reinterpret_cast<const v4f_aux*>(gT + e * 16)[0], g1 = reinterpret_cast<const v4f_aux*>(gT + e * 16)[1];
    const float aj = a[j], saj = sa[j], caj = ca[j];
    float r = g0.x * -s;
    r = fmaf(g0.y, -c * caj, r);
    r = fmaf(g0.z, c * saj, r);
    r = fmaf(g0.w, -aj * s, r);
    r = fmaf(g1.x, c, r);
    r = fmaf(g1.y, -s * caj, r);
    r = fmaf(g1.z, s * saj, r);
    r = fmaf(g1.w, aj * c, r);
    gq[e] = r;
}
// R = Rz(yaw) Ry(pitch) Rx(roll) for phi = (roll, pitch, yaw)  (utils.py:15-38: `rz @ ry @ rx`, left to right)
__device__ __forceinline__ void euler_parts(const float* phi, float (&M)[3][3], float& sx, float& cx, float& sy, float& cy, float& sz, float& cz) {
    sincos_f32(phi[0], &sx, &cx);
    sincos_f32(phi[1], &sy, &cy);
    sincos_f32(phi[2], &sz, &cz);
    // M = Rz Ry
    M[0][0] = cz * cy; M[0][1] = -sz; M[0][2] = cz * sy;
    M[1][0] = sz * cy; M[1][1] = cz;  M[1][2] = sz * sy;
    M[2][0] = -sy;     M[2][1] = 0.f; M[2][2] = cy;
}
__global__ __launch_bounds__(256) void euler_frames_kernel(const float* phi, int64_t B, float* R) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float M[3][3], sx, cx, sy, cy, sz, cz;
    euler_parts(phi + b * 3, M, sx, cx, sy, cy, sz, cz);
    float* o = R + b * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) {     // (M Rx)[i] = (M_i0, M_i1 cx + M_i2 sx, -M_i1 sx + M_i2 cx)
        o[3 * i] = M[i][0];
        o[3 * i + 1] = fmaf(M[i][2], sx, M[i][1] * cx);
        o[3 * i + 2] = fmaf(M[i][2], cx, -(M[i][1] * sx));
    }
}
__global__ __launch_bounds__(256) void euler_frames_vjp_kernel(const float* phi, const float* gR, int64_t B, float* gphi) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float M[3][3], sx, cx, sy, cy, sz, cz;
    euler_parts(phi + b * 3, M, sx, cx, sy, cy, sz, cz);
    const float* g = gR + b * 9;
    // roll: R = M Rx, dRx/droll = [[0,0,0],[0,-sx,-cx],[0,cx,-sx]]
    float gr = 0.f, gp = 0.f, gy = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gr = fmaf(g[3 * i + 1], fmaf(M[i][2], cx, -(M[i][1] * sx)), gr);
        gr = fmaf(g[3 * i + 2], -fmaf(M[i][2], sx, M[i][1] * cx), gr);
    }
    // pitch: dM/dpitch = Rz dRy: [[-cz sy, 0, cz cy], [-sz sy, 0, sz cy], [-cy, 0, -sy]];  yaw: dM/dyaw = dRz Ry: [[-sz cy, -cz, -sz sy], [cz cy, -sz, cz sy], [0, 0, 0]]
    const float P[3][3] = {{-cz * sy, 0.f, cz * cy}, {-sz * sy, 0.f, sz * cy}, {-cy, 0.f, -sy}};
    const float Y[3][3] = {{-sz * cy, -cz, -sz * sy}, {cz * cy, -sz, cz * sy}, {0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gp = fmaf(g[3 * i], P[i][0], gp);
        gp = fmaf(g[3 * i + 1], fmaf(P[i][2], sx, P[i][1] * cx), gp);
        gp = fmaf(g[3 * i + 2], fmaf(P[i][2], cx, -(P[i][1] * sx)), gp);
        gy = fmaf(g[3 * i], Y[i][0], gy);
        gy = fmaf(g[3 * i + 1], fmaf(Y[i][2], sx, Y[i][1] * cx), gy);
        gy = fmaf(g[3 * i + 2], fmaf(Y[i][2], cx, -(Y[i][1] * sx)), gy);
    }
    gphi[b * 3] = gr;
    gphi[b * 3 + 1] = gp;
    gphi[b * 3 + 2] = gy;
}

}  // namespace

hipError_t launch_fkine(const FkProg* fk_dev, const dcx_fk_desc& fk, const float* q, int64_t B, float* X,
                        hipStream_t st) {
    if (B == 0) return hipSuccess;
    const int d_fk = fk.n_points * fk.point_dim;
    fkine_kernel<<<dim3((unsigned)((B + 63) / 64)), dim3(64), fk_lds_bytes(fk, false), st>>>(
        fk_dev, q, B, X, fk.dof, d_fk, fk_frame_floats(fk));
    return hipGetLastError();
}

hipError_t launch_fkine_vjp(const FkProg* fk_dev, const dcx_fk_desc& fk, const float* q, const float* gX,
                            int64_t B, float* gq, hipStream_t st) {
    if (B == 0) return hipSuccess;
    const int d_fk = fk.n_points * fk.point_dim;
    fkine_vjp_kernel<<<dim3((unsigned)((B + 63) / 64)), dim3(64), fk_lds_bytes(fk, true), st>>>(
        fk_dev, q, gX, B, gq, fk.dof, d_fk, fk_frame_floats(fk));
    return hipGetLastError();
}

hipError_t launch_score_finish(const FinishArgs& a, int64_t n_tiles, size_t lds_bytes, hipStream_t st) {
    score_finish_kernel<<<dim3((unsigned)n_tiles), dim3(64), lds_bytes, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_kernel_matrix(int kind, float kp0, float kp1, const float* x, int64_t B, const float* s, int64_t S,
                                int D, float* K, hipStream_t st) {
    if (B == 0 || S == 0) return hipSuccess;
    ScoreArgs a{};
    a.kind = kind;
    a.kp0 = kp0;
    a.kp1 = kp1;
    dim3 grid((unsigned)((S + 255) / 256), (unsigned)((B + KM_TB - 1) / KM_TB));
    kernel_matrix_kernel<<<grid, dim3(256), 0, st>>>(a, x, B, s, S, D, K);
    return hipGetLastError();
}


// ---- measurement aid: the clock the shaders have right now --------------------------------------------------------------------
// One wave samples clock64 (s_memtime: the shader clock) and wall_clock64 (s_memrealtime: a fixed-rate counter) and sleeps until
// `wall_ticks` of the latter have passed; launched on a side stream BESIDE a loop of sweeps it reports the clock the sweep runs
// at (bench.py `roofline.clock`; profiles/r05_clock_under_load.txt is why the line carries it).
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long wall_ticks) {
    const unsigned long long r0 = wall_clock64(), t0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < wall_ticks) {
        __builtin_amdgcn_s_sleep(64);
        r1 = wall_clock64();
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) {
        out[0] = t0;
        out[1] = t1;
        out[2] = r0;
        out[3] = r1;
    }
}
hipError_t launch_dh_frames(const float* q, int64_t B, int dof, const float* a, const float* d, const float* sa, const float* ca,
                            float* T, hipStream_t st) {
    const int64_t n = B * dof;
    if (n == 0) return hipSuccess;
    dh_frames_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(q, n, dof, a, d, sa, ca, T);
    return hipGetLastError();
}
hipError_t launch_dh_frames_vjp(const float* q, int64_t B, int dof, const float* a, const float* sa, const float* ca, const float* gT,
                                float* gq, hipStream_t st) {
    const int64_t n = B * dof;
    if (n == 0) return hipSuccess;
    dh_frames_vjp_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(q, n, dof, a, sa, ca, gT, gq);
    return hipGetLastError();
}
hipError_t launch_euler_frames(const float* phi, int64_t B, float* R, hipStream_t st) {
    if (B == 0) return hipSuccess;
    euler_frames_kernel<<<dim3((unsigned)((B + 255) / 256)), 256, 0, st>>>(phi, B, R);
    return hipGetLastError();
}
hipError_t launch_euler_frames_vjp(const float* phi, const float* gR, int64_t B, float* gphi, hipStream_t st) {
    if (B == 0) return hipSuccess;
    euler_frames_vjp_kernel<<<dim3((unsigned)((B + 255) / 256)), 256, 0, st>>>(phi, gR, B, gphi);
    return hipGetLastError();
}

hipError_t launch_clock_probe(unsigned long long* out, unsigned long long wall_ticks, hipStream_t st) {
    clock_probe_kernel<<<1, 64, 0, st>>>(out, wall_ticks);
    return hipGetLastError();
}

}  // namespace dcx
