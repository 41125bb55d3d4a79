// solve_probe.hip — developer harness for csrc/solve_kernels.hip: times dcx's dense solve without torch and prints where a
// block step spends its time (workgroup 0's stamps: panel, barrier, trailing update, barrier).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDCX_SOLVE_TS -Idiffco_amd/csrc tools/solve_probe.hip -o devlibs/solve_probe
//   devlibs/solve_probe [n ...]
#include "../diffco_amd/csrc/solve_kernels.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {438, 1000, 2000};
    const int nt = getenv("NT") ? atoi(getenv("NT")) : 0;   // 256 / 512 / 0 = by size
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    for (int n : sizes) {
        const int D = 24;
        std::vector<float> pts((size_t)n * D), A((size_t)n * n), B(n);
        srand(n);
        for (auto& v : pts) v = (float)rand() / RAND_MAX;
        for (int i = 0; i < n; ++i) {
            B[i] = (rand() & 1) ? 1.f : -1.f;
            for (int j = 0; j < n; ++j) {
                double d2 = 0;
                for (int k = 0; k < D; ++k) { const double d = pts[(size_t)i * D + k] - pts[(size_t)j * D + k]; d2 += d * d; }
                A[(size_t)i * n + j] = (float)std::sqrt(d2);
            }
        }
        float *dA, *dB, *dX; void* work; int32_t* info;
        const size_t wb = dcx::solve_work_bytes(n, 1);
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, n * 4)); CK(hipMalloc(&dX, n * 4)); CK(hipMalloc(&work, wb)); CK(hipMalloc(&info, 8));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), n * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = n <= 1000 ? 20 : 5;
#ifdef DCX_SOLVE_TS
        {
            std::vector<unsigned long long> zero(8 * 2048, 0);
            CK(hipMemcpyToSymbol(HIP_SYMBOL(dcx::g_solve_ts), zero.data(), zero.size() * 8));
        }
#endif
        for (int w = 0; w < 2; ++w) CK(dcx::launch_solve(dA, dB, dX, n, 1, work, info, prop.multiProcessorCount, false, nt, 0));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) CK(dcx::launch_solve(dA, dB, dX, n, 1, work, info, prop.multiProcessorCount, false, nt, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        int32_t hinfo[2]; CK(hipMemcpy(hinfo, info, 8, hipMemcpyDeviceToHost));
        std::vector<float> X(n); CK(hipMemcpy(X.data(), dX, n * 4, hipMemcpyDeviceToHost));
        double res = 0;
        for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += (double)A[(size_t)i * n + j] * X[j]; res = std::fmax(res, std::fabs(s - B[i])); }
        printf("n=%d  %.1f us per solve  info=%d barriers=%d  max residual %.2e\n", n, ms / reps * 1e3, hinfo[0], hinfo[1], res);
#ifdef DCX_SOLVE_TS
        std::vector<unsigned long long> ts(8 * 2048);
        CK(hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(dcx::g_solve_ts), ts.size() * 8));
        const int steps = hinfo[1] - 1;   // barriers: 1 after the copy + 1 per iteration (iteration 0 factorises panel 0 only)
        double trail = 0, panel = 0, bar = 0;
        for (int s = 1; s <= steps; ++s) {
            trail += (double)(ts[8 * s + 1] - ts[8 * s]);             // workgroup 0 waits for the groups that hold the next panel's columns
            panel += (double)(ts[8 * s + 2] - ts[8 * s + 1]);         // the next panel
            bar += (double)(ts[8 * s + 3] - ts[8 * s + 2]);           // the grid barrier (waiting for the others included)
        }
        const double tick = 0.01;   // us (100 MHz)
        const double runs = reps + 2;
        const unsigned long long* pt = &ts[8 * 2047];
        printf("   panel columns, per column: search + exchange %.2f us, publish %.2f us, read + update %.2f us; panel load %.1f us per panel\n",
               pt[0] * tick / runs / n, pt[1] * tick / runs / n, pt[2] * tick / runs / n, pt[3] * tick / runs / steps);
        printf("   workgroup 0, %d iterations: copy %.1f us | waiting for the next panel's columns %.1f  that panel %.1f  barrier %.1f (sums, us) | back-substitution %.1f us | total %.1f us\n",
               steps, (ts[1] - ts[0]) * tick + 0.0, trail * tick, panel * tick, bar * tick, (double)(ts[3] - ts[2]) * tick, (double)(ts[3] - ts[0]) * tick);
#endif
        hipFree(dA); hipFree(dB); hipFree(dX); hipFree(work); hipFree(info);
    }
    return 0;
}
