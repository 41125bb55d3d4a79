// tools/lone_wave_ubench.hip — developer microbenchmark (round 3): what ONE wave pays per instruction on gfx950.
//
// The fused kernels spend their small-batch launches in phases where one wave of a block works and fifteen wait at a
// barrier (FK chain, J^T, hand-over).  This times straight-line instruction streams on a lone wave with s_memtime, so
// that those phases can be priced by instruction count and kind instead of by guessing:
//     hipcc --offload-arch=gfx950 -O3 -o tools/lone_wave_ubench tools/lone_wave_ubench.hip && tools/lone_wave_ubench
// Each row: cycles per instruction (or per round trip) for a stream of N instructions, the median over the blocks of a
// 256-block grid (one block per CU), with `park` extra waves of the block waiting at s_barrier meanwhile.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

enum {
    T_FMA_IND, T_FMA_DEP, T_PKFMA_IND, T_PKFMA_DEP, T_MOV, T_SALU, T_FMA_SALU, T_READLANE, T_READFIRSTLANE, T_MUL_DEP,
    T_LDS_RT, T_LDS_WRITE, T_LDS_READ_PIPE, T_BRANCH, T_SLOAD_RT, T_GLOAD_RT, T_GLOAD_SC1_RT, T_ATOMIC_RT, T_STORE_SC1_DRAIN,
    T_BARRIER, T_FMA_IND2, T_RCP, T_N
};
static const char* kNames[T_N] = {
    "v_fma_f32, 8 independent accumulators", "v_fma_f32, one dependent chain", "v_pk_fma_f32, 8 independent",
    "v_pk_fma_f32, one dependent chain", "v_mov_b32", "s_add_u32 (SALU)", "v_fma_f32 / s_add_u32 alternating (per pair)",
    "v_readlane_b32 (distinct SGPR dst)", "v_readfirstlane_b32", "v_mul_f32 dependent chain",
    "ds_read_b32 -> s_waitcnt -> dependent address (LDS round trip)", "ds_write_b32 back to back",
    "ds_read_b32 back to back, one wait at the end", "s_cbranch_scc taken (s_cmp + branch per iteration)",
    "s_load_dword -> wait (scalar cache hit round trip)", "global_load_dword -> wait, same line (L1/L2 hit)",
    "global_load_dword sc1 -> wait (agent-scope load round trip)", "global_atomic_add returning, agent scope (round trip)",
    "global_store sc1 + s_waitcnt vmcnt(0) (write-through drain)", "s_barrier, all waves of the block (per barrier)",
    "v_fma_f32, 2 independent accumulators", "v_rcp_f32 dependent chain"};

__global__ __launch_bounds__(1024) void ubench(int test, unsigned long long* out, float* sink, unsigned int* gmem, int park) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.0f;   // every LDS word holds 0 (a valid byte offset)
    __syncthreads();
    if (test == T_BARRIER) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        REP64(__builtin_amdgcn_s_barrier();)
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
        return;
    }
    if (wave != 0) {  // the parked waves wait until the lone wave is done
        __syncthreads();
        return;
    }
    float a0 = lane, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f, b = 1.0001f, c = 0.5f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, 1.f}, p1 = {1.f, 2.f}, p2 = p1, p3 = p1, p4 = p1, p5 = p1, p6 = p1, p7 = p1, pb = {b, b}, pc = {c, c};
    unsigned int s0 = 1, s1 = 2, s2 = 3, s3 = 4, addr = (unsigned)(uintptr_t)lds & 0xffff;
    unsigned int* gp = gmem + blockIdx.x * 64;
    const __attribute__((address_space(4))) unsigned int* kp = (const __attribute__((address_space(4))) unsigned int*)(uintptr_t)gp;
    unsigned int vaddr = addr + lane * 4, vtmp = 0, voff = lane * 4;
    unsigned long long t0, t1;
    int n = 256;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    t0 = __builtin_readcyclecounter();
    switch (test) {
    case T_FMA_IND:
        REP4(REP4(REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)))
        n = 2048;
        break;
    case T_FMA_IND2:
        REP256(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));)
        n = 512;
        break;
    case T_FMA_DEP: REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));) break;
    case T_MUL_DEP: REP256(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b));) break;
    case T_RCP: REP256(asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));) break;
    case T_PKFMA_IND:
        REP64(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                           "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                           : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        n = 512;
        break;
    case T_PKFMA_DEP: REP256(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pb), "v"(pc));) break;
    case T_MOV: REP256(asm volatile("v_mov_b32 %0, %1\n" : "=v"(a1) : "v"(a0));) break;
    case T_SALU: REP256(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");) break;
    case T_FMA_SALU: REP256(asm volatile("v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 1" : "+v"(a0), "+s"(s0) : "v"(b), "v"(c) : "scc");) break;
    case T_READLANE: REP64(asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 4\n v_readlane_b32 %2, %4, 5\n v_readlane_b32 %3, %4, 6"
                                        : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0));) break;
    case T_READFIRSTLANE: REP256(asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(a0));) break;
    case T_LDS_RT:
        REP64(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "=&v"(vtmp), "+v"(vaddr) :: "memory");)
        n = 64;
        break;
    case T_LDS_WRITE: REP256(asm volatile("ds_write_b32 %0, %1" :: "v"(vaddr), "v"(a0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); break;
    case T_LDS_READ_PIPE:
        REP64(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768"
                           : "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4) : "v"(vaddr) : "memory");)
        asm volatile("s_waitcnt lgkmcnt(0)");
        break;
    case T_BRANCH:
        asm volatile("s_mov_b32 %0, 256\n 1: s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b" : "=s"(s0) :: "scc");
        break;
    case T_SLOAD_RT:
        REP64(asm volatile("s_load_dword %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(s0) : "s"(kp) : "memory");)
        n = 64;
        break;
    case T_GLOAD_RT:
        REP64(asm volatile("global_load_dword %0, %1, %2\n s_waitcnt vmcnt(0)\n v_add_u32 %1, %1, %0" : "=&v"(vtmp), "+v"(voff) : "s"(gp) : "memory");)
        n = 64;
        break;
    case T_GLOAD_SC1_RT:
        REP64(asm volatile("global_load_dword %0, %1, %2 sc1\n s_waitcnt vmcnt(0)\n v_add_u32 %1, %1, %0" : "=&v"(vtmp), "+v"(voff) : "s"(gp) : "memory");)
        n = 64;
        break;
    case T_ATOMIC_RT:
        if (lane == 0) {
            REP16(asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=&v"(vtmp) : "v"(gp + 32), "v"(0u) : "memory");)
        }
        n = 16;
        break;
    case T_STORE_SC1_DRAIN:
        REP16(asm volatile("global_store_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" :: "v"(gp + lane), "v"(0u) : "memory");)
        n = 16;
        break;
    default: break;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x] = (t1 - t0) * 1000ull / n;  // milli-cycles per instruction
    sink[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.y + p3.x + p4.x + p5.x + p6.x + p7.x + (float)(s0 + s1 + s2 + s3 + vtmp + vaddr + voff) + (float)(uintptr_t)gp;
    __syncthreads();
}

int main() {
    const int grid = 256;
    unsigned long long* out;
    float* sink;
    unsigned int* gmem;
    hipMalloc(&out, grid * sizeof(*out));
    hipMalloc(&sink, grid * 64 * sizeof(float));
    hipMalloc(&gmem, grid * 64 * sizeof(unsigned));
    hipMemset(gmem, 0, grid * 64 * sizeof(unsigned));
    std::vector<unsigned long long> h(grid);
    printf("# lone-wave instruction costs on this device, cycles (s_memtime) per instruction; median over %d blocks\n", grid);
    printf("%-66s %10s %10s\n", "stream", "alone", "+15 parked");
    for (int t = 0; t < T_N; ++t) {
        double res[2];
        for (int pk = 0; pk < 2; ++pk) {
            const int threads = (t == T_BARRIER) ? (pk ? 1024 : 64) : (pk ? 1024 : 64);
            for (int rep = 0; rep < 3; ++rep) ubench<<<grid, threads>>>(t, out, sink, gmem, pk);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, grid * sizeof(*out), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            res[pk] = (t == T_BARRIER) ? h[grid / 2] / 64.0 : h[grid / 2] / 1000.0;
        }
        printf("%-66s %10.1f %10.1f\n", kNames[t], res[0], res[1]);
    }
    return 0;
}
