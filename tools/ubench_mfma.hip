// Micro-benchmark: int8 MFMA issue rate and the price of s_barrier on gfx950.
//   waves per workgroup W (4 = one per SIMD, 8 = two per SIMD), B = MFMAs per wave between barriers (0 = no barrier).
// Prints TOPS, the shader clock measured with s_memtime, and cycles per MFMA per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma.hip -o tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int W, int B, int SHAPE>
__global__ __launch_bounds__(W * 64) void mfma_kernel(int iters, int* out, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    i32x4 a = {lane, 1, 2, 3}, b = {3, 2, 1, lane};
    asm volatile("" : "+v"(a), "+v"(b));
    i32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    i32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if constexpr (SHAPE == 32) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
            } else {
                d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d3, 0, 0, 0);
            }
            if constexpr (B == 4) __builtin_amdgcn_s_barrier();
        }
        if constexpr (B == 8) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    s += d0[0] + d1[1] + d2[2] + d3[3];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W, int B, int SHAPE>
void run(const char* name, int grid, int* out, unsigned long long* cyc)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mfma_kernel<W, B, SHAPE>), dim3(grid), dim3(W * 64), 0, 0, iters, out, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mfma_kernel<W, B, SHAPE>), dim3(grid), dim3(W * 64), 0, 0, iters, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[1024];
    CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
    const double nm = (double)grid * W * iters * 8;                        // MFMAs
    const double ops = nm * (SHAPE == 32 ? 65536.0 : 32768.0);
    const double per_simd = (double)iters * 8 * (W / 4);                    // MFMAs per SIMD
    printf("%-34s grid=%3d  %8.1f us  %7.1f TOPS  clk(memtime)=%.0f MHz  cycles/MFMA/SIMD=%.1f\n", name, grid, ms * 1e3,
           ops / (ms * 1e-3) / 1e12, avg / (ms * 1e3), avg / per_simd);
}

// steady-state form (`ubench_mfma SECONDS`): the same kernel launched back to back for SECONDS so the chip settles into the clock its
// power budget allows; reports the sustained rate over the second half of the run
template <int W, int SHAPE>
void run_long(const char* name, double seconds, int* out, unsigned long long* cyc)
{
    const int iters = 200000, grid = 256;                                  // ~27 ms per launch
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double total_ms = 0, late_ms = 0, late_cyc = 0; int late_n = 0;
    while (total_ms < seconds * 1e3) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_kernel<W, 0, SHAPE>), dim3(grid), dim3(W * 64), 0, 0, iters, out, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        if (total_ms > seconds * 500) {
            unsigned long long h[256];
            CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
            double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
            late_ms += ms; late_cyc += avg; ++late_n;
        }
    }
    const double ops = (double)grid * W * iters * 8 * (SHAPE == 32 ? 65536.0 : 32768.0) * late_n;
    printf("%-28s sustained over %.1f s: %7.1f TOPS  clk(memtime)=%.0f MHz  cycles/MFMA/SIMD=%.2f\n", name, late_ms * 1e-3,
           ops / (late_ms * 1e-3) / 1e12, late_cyc / (late_ms * 1e3), late_cyc / ((double)iters * 8 * (W / 4) * late_n));
    fflush(stdout);
}

int main(int argc, char** argv)
{
    int* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, 8 * 1024));
    if (argc > 1) {
        const double s = atof(argv[1]);
        run_long<4, 16>("w4 mfma_i32_16x16x64_i8", s, out, cyc);
        run_long<8, 16>("w8 mfma_i32_16x16x64_i8", s, out, cyc);
        run_long<4, 32>("w4 mfma_i32_32x32x32_i8", s, out, cyc);
        return 0;
    }
#define RUN(W, B, S, G) run<W, B, S>("w" #W " barrier/" #B " mfma" #S, G, out, cyc)
    RUN(4, 0, 32, 256); RUN(8, 0, 32, 256); RUN(8, 8, 32, 256); RUN(8, 4, 32, 256); RUN(4, 8, 32, 256); RUN(4, 4, 32, 256);
    RUN(8, 0, 32, 172); RUN(8, 8, 32, 172);
    RUN(4, 0, 16, 256); RUN(8, 0, 16, 256); RUN(8, 8, 16, 256);
    RUN(16, 0, 32, 256); RUN(16, 8, 32, 256);
    return 0;
}
