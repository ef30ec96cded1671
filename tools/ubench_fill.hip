// Micro-benchmark: how fast can dedicated loader waves FILL LDS from L2-resident contiguous data, by
//   mode 0: LDS-DMA (global_load_lds_dwordx4)            mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
// with R "reader" waves issuing ds_read_b128 traffic alongside (0 = none).  One workgroup per CU, 256 CUs, each
// streams `bytes_per_wg` from its own slice of a 64 MiB buffer (L2/MALL resident after the warm-up launch).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_fill.hip -o tools/ubench_fill
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// LW loader waves, RW reader waves.  Ring of 96 KiB.  DEPTH = pieces in flight per loader wave.
template <int MODE, int LW, int RW, int DEPTH, int READS_PER_PIECE>
__global__ __launch_bounds__((LW + RW) * 64) void fill_kernel(const uint8_t* src, int pieces_per_wave, uint32_t* sink, int shared_src)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int RING = 96 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t acc = 0;
    if (wave < LW) {
        const size_t wg_bytes = (size_t)pieces_per_wave * LW * 1024;
        const uint8_t* p = src + (shared_src ? 0 : (size_t)blockIdx.x * wg_bytes) + (size_t)wave * 1024 + lane * 16;
        uint8_t* l = lds + wave * 1024;
        if constexpr (MODE == 0) {
            int off = 0;
            for (int i = 0; i < pieces_per_wave; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)i * LW * 1024),
                                                 (__attribute__((address_space(3))) void*)(l + off), 16, 0, 0);
                off += LW * 1024; if (off >= RING) off = 0;
                if ((i & (DEPTH - 1)) == DEPTH - 1) wait_vmcnt<DEPTH / 2>();
            }
            wait_vmcnt<0>();
        } else {
            u32x4 r[DEPTH];
            int off = 0;
            for (int i0 = 0; i0 < pieces_per_wave; i0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) r[d] = *(const u32x4*)(p + (size_t)(i0 + d) * LW * 1024);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    *(u32x4*)(l + off + lane * 16) = r[d];
                    off += LW * 1024; if (off >= RING) off = 0;
                }
            }
        }
    } else {
        // reader waves: READS_PER_PIECE ds_read_b128 per piece the loaders move (the GEMM reads each byte ~2.7x)
        const int total = pieces_per_wave * LW * READS_PER_PIECE / (RW > 0 ? RW : 1);
        int off = (wave - LW) * 4096 + lane * 16;
        for (int i = 0; i < total; i += 8) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const u32x4 v = *(const u32x4*)(lds + off);
                acc ^= v[0] ^ v[2];
                off += 1024; if (off >= RING) off -= RING;
            }
        }
    }
    __syncthreads();
    acc ^= *(uint32_t*)(lds + tid * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int LW, int RW, int DEPTH, int RPP>
void run(const char* name, const uint8_t* src, uint32_t* sink, int shared_src)
{
    const int grid = 256, pieces_per_wave = 1536 / LW * 2;              // 3 MiB per workgroup
    const size_t shm = 96 * 1024;
    auto k = fill_kernel<MODE, LW, RW, DEPTH, RPP>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3((LW + RW) * 64), shm, 0, src, pieces_per_wave, sink, shared_src);
    CHECK(hipDeviceSynchronize());
    const int iters = 10;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3((LW + RW) * 64), shm, 0, src, pieces_per_wave, sink, shared_src);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, bytes = (double)grid * pieces_per_wave * LW * 1024;
    printf("%-52s %8.2f us  %7.2f TB/s  %6.1f GB/s/CU\n", name, us, bytes / us / 1e6, bytes / us / 1e3 / grid);
}

int main()
{
    uint8_t* src; uint32_t* sink;
    const size_t total = (size_t)256 * 3 * 1024 * 1024 + (1 << 20);
    CHECK(hipMalloc(&src, total)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(src, 3, total));
#define RUN(MODE, LW, RW, DEPTH, RPP, SH) run<MODE, LW, RW, DEPTH, RPP>("mode" #MODE " loaders" #LW " readers" #RW " depth" #DEPTH " reads/piece" #RPP " shared" #SH, src, sink, SH)
    RUN(0, 4, 0, 8, 0, 1); RUN(1, 4, 0, 8, 0, 1);
    RUN(0, 8, 0, 8, 0, 1); RUN(1, 8, 0, 8, 0, 1);
    RUN(0, 4, 4, 8, 3, 1); RUN(1, 4, 4, 8, 3, 1);
    RUN(0, 4, 8, 8, 3, 1); RUN(1, 4, 8, 8, 3, 1);
    RUN(1, 4, 0, 16, 0, 1); RUN(1, 8, 0, 16, 0, 1); RUN(1, 2, 0, 16, 0, 1);
    RUN(0, 4, 0, 8, 0, 0); RUN(1, 4, 0, 8, 0, 0); RUN(1, 8, 0, 16, 0, 0);
    return 0;
}
