// pack_kernels.hip — the model's support rows built ON THE DEVICE (round 4; dcx_model_create_ex / dcx_model_update).
//
// dcx_model_create used to stage everything through the host: a blocking device-to-host copy of the supports and the
// weights, the row packing in a C++ loop, five hipMalloc + hipMemcpy back.  The active-learning loop
// of the reference (collision_checkers.py:220-252: update -> train -> fit_poly -> score) rebuilds the model every round,
// from tensors that already live in HBM.  pack_rows_kernel does the same packing there, on the caller's stream:
//   * rows whose C weights are all zero are dropped (what max_num_supports padding produces, kernel_perceptrons.py:
//     159-196): an in-order stream compaction (wave ballots + a block scan), so the kept rows stand in the order the
//     host loop leaves them in;
//   * a row = [D coordinates | zero pad to Dt | C weights x fold | (C > 1) their sum | |s|^2 | pad] with the same
//     arithmetic as the host loop (float products and sums in the same order, |s|^2 accumulated in double and rounded
//     once): the device-built rows are bit-identical to the host-built ones (tests/test_gpu_api.py);
//   * the centred copy for the expanded-form sweeps: centroid per feature as a SEQUENTIAL double sum over the kept rows
//     (one lane per feature - the host's order, so the same float), s - c in fp32, |s - c|^2 (+ 2/gamma for RQ2) in the last
//     column, and the largest |s - c|^2 for the RQ rule (dcx_api.hip xf_rq_ok).
// The host reads back 16 bytes (kept rows, max |s - c|^2) - the launch geometry depends on the row count - and nothing else.
// One workgroup: a model is a few thousand rows; the packing is latency, not bandwidth
// (dcx_model_update at S = 2000: ~0.1 ms including the read-back; profiles/r04_model_latency.txt).
#include <algorithm>

#include "pack_kernels.h"

namespace dcx {
namespace {

constexpr int kPackThreads = 1024;

__global__ __launch_bounds__(kPackThreads) void pack_rows_kernel(const PackArgs a) {
    __shared__ int s_wave_cnt[kPackThreads / 64];
    __shared__ int s_base;
    __shared__ double s_max[kPackThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ss_off = a.Dt + a.Cl + (a.Cl > 1 ? 1 : 0);
    if (tid == 0) s_base = 0;
    __syncthreads();
    // ---- compaction + direct rows ---------------------------------------------------------------------------------------
    for (int64_t j0 = 0; j0 < a.S; j0 += kPackThreads) {
        const int64_t j = j0 + tid;
        bool keep = false;
        if (j < a.S)
            for (int c = 0; c < a.C; ++c) keep |= (a.w[j * a.C + c] != 0.0f);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (lane == 0) s_wave_cnt[wave] = __builtin_popcountll(m);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += s_wave_cnt[w];
        const int dst = before + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (keep) {
            float* r = a.rows + (size_t)dst * a.RS;
            const float* f = a.feat + j * a.D;
            double ss = 0.0;
            for (int k = 0; k < a.D; ++k) {
                const float v = f[k];
                r[k] = v;
                ss += (double)v * (double)v;
            }
            for (int k = a.D; k < a.RS; ++k) r[k] = 0.0f;
            float sum = 0.0f;
            for (int c = 0; c < a.C; ++c) {
                const float v = a.w[j * a.C + c] * a.fold;
                r[a.Dt + c] = v;
                sum += v;
            }
            if (a.Cl > 1) r[a.Dt + a.Cl] = sum;
            r[ss_off] = (float)ss;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < kPackThreads / 64; ++w) tot += s_wave_cnt[w];
            s_base += tot;
        }
        __syncthreads();
    }
    const int kept = s_base;
    // the tail padding behind the kept rows (look-ahead loads of the sweeps), both arrays
    for (int i = tid; i < a.tail_floats; i += kPackThreads) {
        a.rows[(size_t)kept * a.RS + i] = 0.0f;
        a.rows_xf[(size_t)kept * a.RS + i] = 0.0f;
    }
    __syncthreads();
    // ---- centroid: one lane per feature, the kept rows in order, double accumulation (dcx_model_create's loop) ------------
    if (tid < a.Dt) {
        float c = 0.0f;
        if (a.centred && tid < a.D && kept > 0) {
            double acc = 0.0;
            const float* col = a.rows + tid;
            int j = 0;
            for (; j + 32 <= kept; j += 32) {   // 32 loads in flight, the additions in row order
                float v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = col[(size_t)(j + u) * a.RS];
#pragma unroll
                for (int u = 0; u < 32; ++u) acc += (double)v[u];
            }
            for (; j < kept; ++j) acc += (double)col[(size_t)j * a.RS];
            c = (float)(acc / (double)kept);
        }
        a.centre[tid] = c;
    }
    __syncthreads();
    // ---- centred rows ---------------------------------------------------------------------------------------------------
    double mx = 0.0;
    for (int j = tid; j < kept; j += kPackThreads) {
        const float* r = a.rows + (size_t)j * a.RS;
        float* x = a.rows_xf + (size_t)j * a.RS;
        double ss = 0.0;
        for (int k = 0; k < a.D; ++k) {
            const float v = r[k] - a.centre[k];
            x[k] = v;
            ss += (double)v * (double)v;
        }
        for (int k = a.D; k < a.RS; ++k) x[k] = r[k];
        x[ss_off] = a.rq2 ? (float)(ss + (double)a.seed) : (float)ss;
        mx = ss > mx ? ss : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(mx, o, 64);
        mx = other > mx ? other : mx;
    }
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < kPackThreads / 64; ++w) t = s_max[w] > t ? s_max[w] : t;
        a.info[0] = kept;
        a.info[1] = 0;
        reinterpret_cast<double*>(a.info)[1] = t;
    }
}

__global__ __launch_bounds__(256) void interleave_rows_kernel(const float* rows, float* il, int32_t kept, int32_t RS, int32_t tail) {
    const int32_t even = (kept + 1) & ~1;
    const int64_t n = (int64_t)even * RS + tail;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= (int64_t)even * RS) {
            il[i] = 0.0f;
            continue;
        }
        const int32_t j = (int32_t)(i / RS), e = (int32_t)(i % RS);
        il[(size_t)(j >> 1) * 2 * RS + 2 * e + (j & 1)] = j < kept ? rows[i] : 0.0f;
    }
}

}  // namespace

hipError_t launch_interleave_rows(const float* rows, float* rows_il, int32_t kept, int32_t RS, int32_t tail, hipStream_t stream) {
    const int64_t n = (int64_t)((kept + 1) & ~1) * RS + tail;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 1024);
    interleave_rows_kernel<<<dim3(blocks ? blocks : 1), 256, 0, stream>>>(rows, rows_il, kept, RS, tail);
    return hipGetLastError();
}

hipError_t launch_pack_rows(const PackArgs& a, hipStream_t stream) {
    pack_rows_kernel<<<dim3(1), dim3(kPackThreads), 0, stream>>>(a);
    return hipGetLastError();
}

}  // namespace dcx
