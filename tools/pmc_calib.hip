// pmc_calib.hip — known-byte-count kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// for the access widths the score kernel uses (MI355X_MICROARCH.md §HBM: FETCH_SIZE is only calibrated
// for 16 B/lane streams; other widths must be calibrated on a known byte count).
//   calib_copy4  : N floats read, N floats written, 4 B per lane  (q staging / score+grad stores)
//   calib_copy16 : the same bytes at 16 B per lane                (reference point: FETCH_SIZE reads 1/2)
// Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_copy4(const float* __restrict__ a, float* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i] + 1.0f;
}
__global__ void calib_copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) { float4 v = a[i]; v.x += 1.0f; b[i] = v; }
}
int main() {
    const size_t n = (size_t)1 << 28;  // 1 GiB read + 1 GiB written per kernel: far beyond the 256 MiB L3
    float *a, *b;
    if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&b, n * 4) != hipSuccess) return 1;
    (void)hipMemset(a, 0, n * 4);
    for (int r = 0; r < 3; ++r) {
        calib_copy4<<<(unsigned)(n / 256), 256>>>(a, b, n);
        calib_copy16<<<(unsigned)(n / 4 / 256), 256>>>((const float4*)a, (float4*)b, n / 4);
    }
    (void)hipDeviceSynchronize();
    printf("each kernel: %zu bytes read, %zu bytes written\n", n * 4, n * 4);
    return 0;
}
