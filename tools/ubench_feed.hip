// Micro-benchmark of the operand-feed path of the GEMM on gfx950: how fast can one workgroup per CU pull its
// (BM+BN) x 64-byte k-slab per step from L2/HBM into LDS?  Variants:
//   layout  : 0 = row-strided 64-byte segments of [rows, pitch] matrices (what a plain [N,K] int8 weight gives)
//             1 = packed: the slab of a step is one contiguous block
//             2 = row-strided 128-byte segments (BK = 128)
//   path    : 0 = LDS-DMA (global_load_lds_dwordx4), 1 = global_load_dwordx4 -> VGPR -> ds_write_b128,
//             2 = global_load_dwordx4 -> VGPR only (xor-reduced)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_feed.hip -o gpurun_out/ubench_feed
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

struct Args {
    const uint8_t* w; const uint8_t* x; uint32_t* sink;
    int pitch;       // bytes between rows (strided layouts)
    int nk;          // steps
    int tiles_m;     // WGs sharing one weight panel
};

// BM activation rows + BN weight rows, SEG bytes per row and step, NW waves, LOOK stages in flight
template <int BM, int BN, int SEG, int NW, int LOOK, int LAYOUT, int PATH, int READS = 0>
__global__ __launch_bounds__(NW * 64) void feed_kernel(const Args a)
{
    constexpr int CH = SEG / 16;
    constexpr int TI = (BM + BN) * CH / 64;            // 1-KiB pieces per stage
    constexpr int LOADS = TI / NW;
    constexpr int STAGE = (BM + BN) * SEG;
    constexpr int NS = LOOK + 2;
    static_assert(TI % NW == 0, "");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = blockIdx.x / a.tiles_m, tm = blockIdx.x % a.tiles_m;

    const uint8_t* src[LOADS];
    size_t step_stride[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int pw = i * NW + wave, qd = pw * 64 + lane, rr = qd / CH, pc = qd % CH;
        if (LAYOUT == 1) {      // packed: per step one contiguous block per operand: [panel][step][rows*SEG]
            if (rr < BN) { src[i] = a.w + ((size_t)tn * a.nk) * (BN * SEG) + (size_t)qd * 16; step_stride[i] = BN * SEG; }
            else         { src[i] = a.x + ((size_t)tm * a.nk) * (BM * SEG) + (size_t)(qd - BN * CH) * 16; step_stride[i] = BM * SEG; }
        } else {
            if (rr < BN) src[i] = a.w + (size_t)(tn * BN + rr) * a.pitch + pc * 16;
            else         src[i] = a.x + (size_t)(tm * BM + rr - BN) * a.pitch + pc * 16;
            step_stride[i] = SEG;
        }
    }
    uint32_t acc = 0;
    if constexpr (PATH == 0) {
        auto stage = [&](int buf, int kt) {
#pragma unroll
            for (int i = 0; i < LOADS; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * step_stride[i]),
                                                 (__attribute__((address_space(3))) void*)(lds + buf * STAGE + (i * NW + wave) * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < LOOK; ++s) stage(s, s);
        int nxt = LOOK % NS;
        int cur = 0;
        for (int kt = 0; kt < a.nk; ++kt) {
            if (kt + LOOK < a.nk) { stage(nxt, kt + LOOK); wait_vmcnt<LOADS * LOOK>(); }
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if constexpr (READS > 0) {       // concurrent fragment-read traffic: READS x ds_read_b128 per wave and stage
#pragma unroll
                for (int r = 0; r < READS; ++r) {
                    const u32x4 v = *(const u32x4*)(lds + cur * STAGE + ((r * 64 + lane) * 16 + wave * 1024) % STAGE);
                    acc ^= v[0] ^ v[3];
                }
            }
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
            cur = (cur + 1 == NS) ? 0 : cur + 1;
        }
        acc ^= *(uint32_t*)(lds + tid * 4);
    } else {
        u32x4 r[LOOK][LOADS];
#pragma unroll
        for (int s = 0; s < LOOK; ++s)
#pragma unroll
            for (int i = 0; i < LOADS; ++i) r[s][i] = *(const u32x4*)(src[i] + (size_t)s * step_stride[i]);
        // rotate through LOOK register stages with a fully unrolled inner loop so indices stay static
        for (int kt0 = 0; kt0 < a.nk; kt0 += LOOK) {
#pragma unroll
            for (int s = 0; s < LOOK; ++s) {
                const int kt = kt0 + s;
                if (kt < a.nk) {
                    if constexpr (PATH == 1) {
#pragma unroll
                        for (int i = 0; i < LOADS; ++i) *(u32x4*)(lds + (kt & 1) * STAGE + ((i * NW + wave) * 64 + lane) * 16) = r[s][i];
                    } else {
#pragma unroll
                        for (int i = 0; i < LOADS; ++i) acc ^= r[s][i][0] ^ r[s][i][1] ^ r[s][i][2] ^ r[s][i][3];
                    }
                    if (kt + LOOK < a.nk) {
#pragma unroll
                        for (int i = 0; i < LOADS; ++i) r[s][i] = *(const u32x4*)(src[i] + (size_t)(kt + LOOK) * step_stride[i]);
                    }
                    if constexpr (PATH == 1) __syncthreads();
                }
            }
        }
        if constexpr (PATH == 1) acc = *(uint32_t*)(lds + tid * 4);
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

template <int BM, int BN, int SEG, int NW, int LOOK, int LAYOUT, int PATH, int READS = 0>
void run(const char* name, Args a, int ntiles, int M, int N, int K)
{
    constexpr int NS = LOOK + 2;
    const size_t shm = (size_t)(BM + BN) * SEG * (PATH == 0 ? NS : 2);
    auto k = feed_kernel<BM, BN, SEG, NW, LOOK, LAYOUT, PATH, READS>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    a.nk = K / SEG;
    a.tiles_m = M / BM;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(ntiles), dim3(NW * 64), shm, 0, a);
    CHECK(hipDeviceSynchronize());
    const int iters = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(ntiles), dim3(NW * 64), shm, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, bytes = (double)ntiles * (BM + BN) * K;
    printf("%-44s tiles=%4d  %8.2f us  %7.2f TB/s  %6.1f GB/s/CU(active)\n", name, ntiles, us, bytes / us / 1e6,
           bytes / us / 1e3 / (ntiles < 256 ? ntiles : 256));
}

int main()
{
    const int M = 512, N = 11008, K = 4096;
    uint8_t *w, *x; uint32_t* sink;
    CHECK(hipMalloc(&w, (size_t)(N + 256) * K)); CHECK(hipMalloc(&x, (size_t)M * K)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(w, 1, (size_t)(N + 256) * K)); CHECK(hipMemset(x, 2, (size_t)M * K));
    Args a{w, x, sink, K, 0, 0};
#define RUN(BM, BN, SEG, NW, LOOK, LAYOUT, PATH) run<BM, BN, SEG, NW, LOOK, LAYOUT, PATH>(#BM "x" #BN " seg" #SEG " w" #NW " look" #LOOK " layout" #LAYOUT " path" #PATH, a, (M / BM) * (N / BN), M, N, K)
#define RUNR(BM, BN, SEG, NW, LOOK, LAYOUT, PATH, RD) run<BM, BN, SEG, NW, LOOK, LAYOUT, PATH, RD>(#BM "x" #BN " seg" #SEG " w" #NW " look" #LOOK " layout" #LAYOUT " path" #PATH " reads" #RD, a, (M / BM) * (N / BN), M, N, K)
    RUNR(256, 128, 64, 8, 3, 1, 0, 0);
    RUNR(256, 128, 64, 8, 3, 1, 0, 4);
    RUNR(256, 128, 64, 8, 3, 1, 0, 8);
    RUNR(256, 128, 64, 8, 3, 1, 0, 16);
    RUNR(256, 128, 64, 8, 3, 1, 0, 32);
    RUN(256, 128, 64, 8, 3, 0, 0);
    RUN(256, 128, 64, 8, 3, 1, 0);
    RUN(256, 128, 128, 8, 1, 2, 0);
    RUN(256, 128, 64, 8, 3, 0, 1);
    RUN(256, 128, 64, 8, 3, 1, 1);
    RUN(256, 128, 64, 8, 3, 0, 2);
    RUN(256, 128, 64, 8, 3, 1, 2);
    RUN(256, 128, 128, 8, 2, 2, 2);
    RUN(256, 128, 64, 4, 3, 0, 0);
    RUN(256, 128, 64, 4, 3, 1, 0);
    RUN(256, 128, 64, 12, 3, 1, 0);
    RUN(256, 128, 64, 12, 3, 0, 0);
    RUN(256, 128, 64, 8, 4, 1, 0);
    RUN(256, 128, 64, 8, 4, 0, 0);
    RUN(128, 64, 64, 4, 3, 0, 0);
    RUN(128, 64, 64, 4, 3, 1, 0);
    RUN(128, 128, 64, 4, 3, 1, 0);
    RUN(128, 128, 64, 4, 3, 0, 0);
    return 0;
}
