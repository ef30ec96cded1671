// Micro-benchmark: do independent VALU instructions of the SAME wave overlap with its int8 MFMAs on gfx950, and does
// it matter whether the accumulators live in arch VGPRs or in AccVGPRs?  One wave per SIMD (4 waves / workgroup),
// 256 workgroups.  Per loop iteration: 12 MFMAs 32x32x32 i8 on 6 accumulators, V VALU ops (v_lshlrev / v_and on
// registers the MFMAs do not touch) behind each MFMA, optionally one ds_read_b128 behind each of the first 5.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_valu.hip -o tools/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int AGPR>
__device__ __forceinline__ void mfma(i32x16& c, const i32x4& a, const i32x4& b)
{
    if constexpr (AGPR) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int V, int AGPR, int LDS, int W = 4>
__global__ __launch_bounds__(W * 64) void k(int iters, int* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[W * 64 * 16];
    const int lane = threadIdx.x & 63;
    i32x4 a = {lane, 1, 2, 3}, b = {3, 2, 1, lane};
    asm volatile("" : "+v"(a), "+v"(b));
    i32x16 c[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0;
    uint32_t x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = lane * (i + 1); y[i] = 0; asm volatile("" : "+v"(x[i])); }
    uint32_t mask;
    asm volatile("s_mov_b32 %0, 0xf0f0f0f0" : "=s"(mask));
    i32x4 rd[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) rd[i] = i32x4{0, 0, 0, 0};
    const uint8_t* lp = lds + threadIdx.x * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            mfma<AGPR>(c[m % 6], a, b);
            if (LDS && m < 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[m]) : "v"((uint32_t)(uintptr_t)lp), "i"(0));
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (v & 1) asm volatile("v_and_b32 %0, %1, %2" : "=v"(y[v % 8]) : "s"(mask), "v"(x[v % 8]));
                else       asm volatile("v_lshlrev_b32 %0, 4, %1" : "=v"(y[v % 8]) : "v"(x[(v + 3) % 8]));
            }
        }
        if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += y[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) s += rd[i][0];
    if (s == 0x7fffffff) out[0] = s;
}

template <int V, int AGPR, int LDS, int W = 4>
void run(int* out)
{
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<V, AGPR, LDS, W>), dim3(grid), dim3(W * 64), 0, 0, iters, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<V, AGPR, LDS, W>), dim3(grid), dim3(W * 64), 0, 0, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_per_mfma = ms * 1e6 / (iters * 12.0 * (W / 4));
    const double tops = (double)grid * W * iters * 12 * 65536.0 / (ms * 1e-3) / 1e12;
    printf("waves=%d valu/mfma=%2d acc=%s lds_reads=%d  %8.1f us  %6.2f ns/MFMA/SIMD (%.1f clk @2.1GHz)  %7.1f TOPS\n", W, V, AGPR ? "AGPR" : "VGPR",
           LDS, ms * 1e3, ns_per_mfma, ns_per_mfma * 2.1, tops);
}

int main()
{
    int* out;
    CHECK(hipMalloc(&out, 64));
    run<0, 0, 0>(out); run<2, 0, 0>(out); run<4, 0, 0>(out); run<6, 0, 0>(out); run<8, 0, 0>(out); run<12, 0, 0>(out);
    run<0, 1, 0>(out); run<2, 1, 0>(out); run<4, 1, 0>(out); run<6, 1, 0>(out); run<8, 1, 0>(out); run<12, 1, 0>(out);
    run<0, 0, 1>(out); run<5, 0, 1>(out); run<0, 1, 1>(out); run<5, 1, 1>(out);
    run<0, 0, 0, 8>(out); run<4, 0, 0, 8>(out); run<8, 0, 0, 8>(out); run<12, 0, 0, 8>(out); run<8, 0, 1, 8>(out);
    return 0;
}
