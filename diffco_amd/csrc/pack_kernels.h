// pack_kernels.h — host view of the device-side row packing (pack_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcx {

struct PackArgs {
    const float* feat;   // [S, D] device
    const float* w;      // [S, C] device
    float* rows;         // [cap + tail] x RS: the direct-form rows
    float* rows_xf;      // the centred copy
    float* centre;       // [Dt]
    int32_t* info;       // 16 bytes out: [0] kept rows, [1] 0, then a double: max |s - c|^2 over the kept rows
    int64_t S;
    int32_t D, Dt, C, RS;
    int32_t Cl;          // class count of the ROW LAYOUT (>= C: the compiled count; columns C .. Cl-1 are zero weights)
    int32_t tail_floats; // zeroed floats behind the last kept row (look-ahead loads of the sweeps)
    int32_t centred;     // features an FK transform produced: centre = centroid of the kept supports (else zero)
    int32_t rq2;         // RQKernel(p = 2): the centred rows' last column carries |s - c|^2 + seed
    float fold, seed;    // weight factor (1/eps, (2/gamma)^2 or 1); 2/gamma
};
hipError_t launch_pack_rows(const PackArgs& a, hipStream_t stream);
// the pair-interleaved copy of `kept` rows for the two-rows-per-instruction sweeps (score_kernel.h pair2): row j's element e at
// rows_il[(j >> 1) * 2 RS + 2 e + (j & 1)]; an odd count is padded with a zero row; `tail` zeroed floats behind
hipError_t launch_interleave_rows(const float* rows, float* rows_il, int32_t kept, int32_t RS, int32_t tail, hipStream_t stream);

}  // namespace dcx
