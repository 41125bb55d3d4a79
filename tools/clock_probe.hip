// tools/clock_probe.hip — developer tool (GPU box): what clock do the shaders really run at under the sweep's load?
// s_memtime (clock64) against s_memrealtime (wall_clock64, a fixed-rate counter) and HIP event time, for a light launch
// (one workgroup) and a chip-filling one (8 waves per SIMD of v_pk_fma_f32 / v_fma_f32), early and late in a 50 ms run.
// Build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o devlibs/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <time.h>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* ts, int iters, float seed) {
    float2_ p[8];
    float a[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = float2_{a[i], a[i] + 1.f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const float2_ m2 = {m, m}, c2 = {c, c};
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
            }
        }
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (threadIdx.x == 0) {
        ts[blockIdx.x * 4 + 0] = t0; ts[blockIdx.x * 4 + 1] = t1; ts[blockIdx.x * 4 + 2] = r0; ts[blockIdx.x * 4 + 3] = r1;
    }
    if (s == 123.456f) ts[0] = (unsigned long long)s;
}

// the clock as a function of the time since the launch began: block 0's first wave stamps both counters every `chunk` instructions
__global__ __launch_bounds__(256) void k_ramp(unsigned long long* ts, int chunks, int chunk, float seed) {
    float2_ p[8];
    for (int i = 0; i < 8; ++i) p[i] = float2_{seed + i + threadIdx.x, seed + i + 1.f};
    const float2_ m2 = {1.0000001f, 1.0000001f}, c2 = {1e-9f, 1e-9f};
    for (int c = 0; c < chunks; ++c) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { ts[2 * c] = clock64(); ts[2 * c + 1] = wall_clock64(); }
        for (int it = 0; it < chunk; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { ts[2 * chunks] = clock64(); ts[2 * chunks + 1] = wall_clock64(); }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 123.456f) ts[0] = (unsigned long long)s;
}
void run_ramp(int blocks, const char* what, int gap_us) {
    const int chunks = 40, chunk = 150;   // 8 x 150 = 1200 pk_fma per chunk and wave: ~2.5 us per chunk at 8 waves per SIMD
    unsigned long long* ts;
    hipMalloc(&ts, sizeof(unsigned long long) * 2 * (chunks + 1));
    std::vector<unsigned long long> h(2 * (chunks + 1));
    for (int l = 0; l < 4; ++l) {   // back to back (gap_us = 0) or with an idle gap in front of each launch
        if (gap_us) { hipDeviceSynchronize(); timespec t{0, gap_us * 1000}; nanosleep(&t, nullptr); }
        k_ramp<<<blocks, 256>>>(ts, chunks, chunk, 1.f);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), ts, sizeof(unsigned long long) * 2 * (chunks + 1), hipMemcpyDeviceToHost);
    printf("%s: GHz in consecutive ~%d-instruction windows of the LAST of four launches (time since its start in us : GHz)\n   ", what, 8 * chunk);
    for (int c = 0; c < chunks; ++c) {
        const double dt = (double)(h[2 * c + 3] - h[2 * c + 1]) * 0.01, ghz = (double)(h[2 * c + 2] - h[2 * c]) / ((double)(h[2 * c + 3] - h[2 * c + 1]) * 10.0);
        printf(" %.0f:%.2f", (double)(h[2 * c + 3] - h[1]) * 0.01, ghz);
        (void)dt;
    }
    printf("\n");
    hipFree(ts);
}

template <int OP>
void run(const char* name, int blocks, int iters, int launches) {
    unsigned long long* ts;
    hipMalloc(&ts, sizeof(unsigned long long) * 4 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<unsigned long long> h(4 * blocks);
    for (int l = 0; l < launches; ++l) {
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(ts, iters, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (l == 0 || l == launches - 1) {
            hipMemcpy(h.data(), ts, sizeof(unsigned long long) * 4 * blocks, hipMemcpyDeviceToHost);
            std::vector<double> ratio, dur;
            for (int b = 0; b < blocks; ++b) {
                ratio.push_back((double)(h[4 * b + 1] - h[4 * b]) / (double)(h[4 * b + 3] - h[4 * b + 2]));
                dur.push_back((double)(h[4 * b + 1] - h[4 * b]));
            }
            std::sort(ratio.begin(), ratio.end());
            std::sort(dur.begin(), dur.end());
            const double instr = (double)iters * 32;   // per wave
            printf("%-13s blocks=%5d launch %3d: event %.3f ms | clock64 ticks per wall_clock64 tick: min %.3f median %.3f max %.3f | "
                   "clock64 ticks per wave-instruction (median block, waves of a SIMD share it): %.2f\n",
                   name, blocks, l, ms, ratio.front(), ratio[ratio.size() / 2], ratio.back(), dur[dur.size() / 2] / instr);
        }
    }
    hipFree(ts);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s  CUs=%d  clockRate=%.0f MHz  wall clock rate=%.1f MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3, wall_khz / 1e3);
    const int cus = prop.multiProcessorCount;
    run<1>("pk_fma light", 1, 20000, 3);
    run<1>("pk_fma full", cus * 8, 20000, 6);      // ~10 ms per launch, 8 waves per SIMD
    run<0>("fma full", cus * 8, 20000, 6);
    run<1>("pk_fma 1/SIMD", cus, 20000, 3);
    run_ramp(cus * 8, "v_pk_fma_f32 on every SIMD, 8 waves each, launches back to back", 0);
    run_ramp(cus * 8, "the same with 200 us of idle GPU before each launch", 200);
    run_ramp(cus * 8, "the same with 20 ms of idle GPU before each launch", 20000);
    return 0;
}
