// valu_ubench.hip — measures the issue cost of the VALU instructions the sweep is made of, on the
// device it runs on.  Used to state the roofline of the fused score kernel in instruction slots
// (DESIGN.md): is v_pk_fma_f32 / v_pk_add_f32 one slot or two, what does v_rsq_f32 / v_rcp_f32 cost.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o build/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    // 8 independent chains per lane so latency never limits issue
    float a[8];
    float2_ p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = float2_{a[i], a[i] + 1.f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const float2_ m2 = {m, m}, c2 = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (OP == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
                if (OP == 7) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int OP>
double run(const char* name, int waves_per_simd, double flops_per_lane_instr) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    const int iters = 20000;
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 32;
    const double wave_instr = instr_per_wave * blocks * 4;
    const double per_simd_per_s = wave_instr / (cus * 4.0) / (ms * 1e-3);
    const double clk = prop.clockRate * 1e3;  // kHz -> Hz (max clock; real clock may be lower)
    printf("%-14s waves/SIMD=%d  %.3f ms  wave-instr/s/SIMD=%.3e  cycles/instr@%.2fGHz=%.2f  chip %.1f T lane-instr/s",
           name, waves_per_simd, ms, per_simd_per_s, clk / 1e9, clk / per_simd_per_s, wave_instr * 64 / (ms * 1e-3) / 1e12);
    if (flops_per_lane_instr > 0) printf("  = %.1f TFLOP/s", wave_instr * 64 * flops_per_lane_instr / (ms * 1e-3) / 1e12);
    printf("\n");
    hipFree(out);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s  CUs=%d  clock=%.0f MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, 2);
        run<1>("v_pk_fma_f32", w, 4);
        run<2>("v_pk_add_f32", w, 2);
        run<6>("v_pk_mul_f32", w, 2);
        run<5>("v_sub_f32", w, 1);
        run<3>("v_rsq_f32", w, 0);
        run<4>("v_rcp_f32", w, 0);
        run<7>("v_sqrt_f32", w, 0);
    }
    return 0;
}
