// Micro-benchmark: is v_mfma_i32_16x16x64_i8 more expensive in POWER per MAC than v_mfma_i32_32x32x32_i8?  (Both are 1024 MACs per cycle per SIMD;
// the 16 x 16 shape reads twice the operand registers per MAC.)  One wave per SIMD, NACC independent accumulators (dependent MFMAs of one chain are
// 8 / 16 instructions apart), launched back to back for SECONDS so the chip settles into the clock its power budget allows; prints the sustained
// rate, the shader clock (cycle counter / wall time) and cycles per MFMA per SIMD.  Board power is read by the caller (rocm-smi / tools/yardstick.py).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_energy.hip -o tools/ubench_mfma_energy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(int iters, int* out, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    i32x4 a = {lane * 0x01010101, 0x7f3c1122, 0x55aa55aa, 0x0f1e2d3c}, b = {0x33cc33cc, 0x12345678, lane * 0x01030507, 0x7a6b5c4d};
    asm volatile("" : "+v"(a), "+v"(b));
    i32x4 d[16];
    i32x16 c[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = i32x16{0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d[i], 0, 0, 0);      // 16 x 16384 MACs
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);   // 8 x 32768 MACs
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += d[i][0] + d[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int WAVES>
void run(const char* name, double seconds, int* out, unsigned long long* cyc)
{
    const int iters = 100000, grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double total_ms = 0, late_ms = 0, late_cyc = 0; int late_n = 0;
    while (total_ms < seconds * 1e3) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<SHAPE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, iters, out, cyc);
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
    const double macs_per_iter = 16.0 * 16384.0;                            // both shapes: 262144 MACs per wave and iteration
    const double ops = (double)grid * WAVES * iters * macs_per_iter * 2 * late_n;
    printf("%-34s sustained over %.1f s: %7.1f TOPS  clk=%.0f MHz  MACs/cycle/SIMD=%.0f\n", name, late_ms * 1e-3, ops / (late_ms * 1e-3) / 1e12,
           late_cyc / (late_ms * 1e3), (double)iters * macs_per_iter * (WAVES / 4) * late_n / late_cyc);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    int* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, 8 * 1024));
    const double s = argc > 1 ? atof(argv[1]) : 2.0;
    const int which = argc > 2 ? atoi(argv[2]) : 0;
    if (which == 0 || which == 1) run<16, 4>("16x16x64_i8, 1 wave/SIMD, 16 acc", s, out, cyc);
    if (which == 0 || which == 2) run<32, 4>("32x32x32_i8, 1 wave/SIMD, 4 acc", s, out, cyc);
    if (which == 0 || which == 3) run<16, 8>("16x16x64_i8, 2 waves/SIMD", s, out, cyc);
    if (which == 0 || which == 4) run<32, 8>("32x32x32_i8, 2 waves/SIMD", s, out, cyc);
    return 0;
}
