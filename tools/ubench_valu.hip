// Micro-benchmark: issue cost (shader cycles per wave-instruction, one wave per SIMD) of the VALU instructions the GEMM epilogue is made of
// on gfx950 - alone and with a 16-cycle MFMA (v_mfma_f32_16x16x32_f16) in front of every group of four - so that the dequantise / convert /
// stage phase can be priced instruction by instruction.  Each variant runs 64 independent instructions of one kind per loop iteration
// (16 registers, 4 uses each), 256 iterations, timed with s_memtime by wave 0 of every workgroup; the median over workgroups is printed.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 256;

// KIND: 0 v_mul_f32, 1 v_pk_mul_f32, 2 v_cvt_f32_i32, 3 v_cvt_pk_f16_f32, 4 v_cvt_f16_f32, 5 v_and_b32, 6 v_pk_mul_f32 with a broadcast
// (op_sel_hi:[0,1]) operand, 7 ds_write_b64, 8 v_fma_f32, 9 v_exp_f32, 10 v_rcp_f32, 11 v_mad_i64_i32, 12 v_lshl_add_u64, 13 nothing (loop + MFMA only)
template <int KIND, int MFMA>
__global__ __launch_bounds__(256) void valu_kernel(float* out, unsigned long long* cyc, float seed)
{
    extern __shared__ unsigned char lds[];
    float r[16];
    f32x2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = seed + threadIdx.x + i; p[i] = f32x2{r[i], r[i] + 1.f}; }
    f32x4 acc = {0, 0, 0, 0};
    f16x8 ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ha[i] = static_cast<_Float16>(threadIdx.x & 3); hb[i] = static_cast<_Float16>(i); }
    const float s = seed * 1.0001f;
    f32x2 sb = {s, s};
    unsigned int ldsa = threadIdx.x * 8;
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if constexpr (MFMA) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = (g * 4 + u) & 15;
                if constexpr (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
                if constexpr (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sb));
                if constexpr (KIND == 2) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));
                if constexpr (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
                if constexpr (KIND == 4) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(r[i]));
                if constexpr (KIND == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
                if constexpr (KIND == 6) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(p[i]) : "v"(sb));
                if constexpr (KIND == 7) asm volatile("ds_write_b64 %0, %1" :: "v"(ldsa), "v"(p[i]) : "memory");
                if constexpr (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
                if constexpr (KIND == 9) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                if constexpr (KIND == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
                if constexpr (KIND == 11) asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, %0" : "+v"(p[i]) : "v"(r[i]) : "vcc");
                if constexpr (KIND == 12) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(p[i]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if ((threadIdx.x & 63) == 0) t1 = __builtin_readcyclecounter();
    float sum = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += r[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int MFMA>
static double run(float* out, unsigned long long* cyc, int grid)
{
    hipLaunchKernelGGL((valu_kernel<KIND, MFMA>), dim3(grid), dim3(256), 8192, 0, out, cyc, 1.5f);
    hipLaunchKernelGGL((valu_kernel<KIND, MFMA>), dim3(grid), dim3(256), 8192, 0, out, cyc, 1.5f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    return static_cast<double>(h[grid / 2]) / ITERS;           // cycles per iteration (64 instructions [+ 16 MFMAs])
}

int main()
{
    const int grid = 256;
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, grid * 256 * sizeof(float)));
    CHECK(hipMalloc(&cyc, grid * sizeof(unsigned long long)));
    const double base0 = run<13, 0>(out, cyc, grid), base1 = run<13, 1>(out, cyc, grid);
    printf("one wave per SIMD, 64 independent instructions per iteration; shader cycles (s_memtime) per instruction\n");
    printf("  loop alone: %.1f cycles per iteration; with 16 x v_mfma_f32_16x16x32_f16: %.1f (%.1f per MFMA)\n", base0, base1, base1 / 16);
    printf("  %-34s %10s %28s\n", "instruction", "alone", "4 behind each 16-cycle MFMA: cycles per group of 4 beyond the MFMA");
#define ROW(K, NAME) do { const double a = run<K, 0>(out, cyc, grid), b = run<K, 1>(out, cyc, grid);                       \
        printf("  %-34s %10.2f %14.1f  (iteration %.0f vs %.0f MFMA only)\n", NAME, (a - base0) / 64, (b - base1) / 16, b, base1); } while (0)
    ROW(0, "v_mul_f32");
    ROW(8, "v_fma_f32");
    ROW(1, "v_pk_mul_f32");
    ROW(6, "v_pk_mul_f32 op_sel_hi:[0,1]");
    ROW(2, "v_cvt_f32_i32");
    ROW(4, "v_cvt_f16_f32");
    ROW(3, "v_cvt_pk_f16_f32");
    ROW(5, "v_and_b32");
    ROW(9, "v_exp_f32");
    ROW(10, "v_rcp_f32");
    ROW(11, "v_mad_i64_i32");
    ROW(12, "v_lshl_add_u64");
    ROW(7, "ds_write_b64");
    return 0;
}
