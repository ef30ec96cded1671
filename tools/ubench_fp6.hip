// Micro-benchmark for carrying W4A4 on the FP6 matrix pipe of gfx950 (v_mfma_scale_f32_16x16x128_f8f6f4, FP6 E3M2 operands, unit
// block scales): every integer of [-8, 8] is an E3M2 value (1 2 3 4 5 6 7 8 = 1, 2, 1.5 x 2, 4, 1.25 x 4, 1.5 x 4, 1.75 x 4, 8; E2M3
// stops at 7.5 and the reference's 4-bit weights reach -8), products are integers <= 64 and the fp32 accumulator is exact below
// 2^24, so the int4 x int4 -> int32 contraction should come out BIT-EXACT at up to twice the int8 MFMA rate.  Answered on the hardware:
//   (1) exactness: random ([-8,7] x [-7,7]) and worst-case (all -8 x 7, alternating signs) operands over K up to 262144 against an
//       integer reference;
//   (2) rate: cycles per MFMA per SIMD next to v_mfma_i32_16x16x64_i8.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_fp6.hip -o tools/ubench_fp6
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// E3M2 code of an integer in [-8, 8]: sign | 3-bit exponent (bias 3) | 2-bit mantissa
__host__ __device__ inline uint32_t fp6_code(int v)
{
    const uint32_t mag[9] = {0x00, 0x0c, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18};
    return (v < 0 ? 0x20u : 0u) | mag[v < 0 ? -v : v];
}

// a, b: [16][K] int8 holding values of [-7, 7]; lane l owns row (l & 15), elements 32 (l >> 4) .. + 31 of every 128-element block, packed
// as a 192-bit little-endian bit stream (element e at bit 6 e).  out[16][16] = a b^T as int32 (from the fp32 accumulator).
__global__ __launch_bounds__(64) void exact_kernel(const int8_t* a, const int8_t* b, int K, int* out, int* nonint)
{
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < K; kb += 128) {
        uint32_t pa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int e = 0; e < 32; ++e) {
            const uint32_t ca = fp6_code(a[r * K + kb + 32 * g + e]), cb = fp6_code(b[r * K + kb + 32 * g + e]);
            const int bit = 6 * e, w = bit >> 5, o = bit & 31;
            pa[w] |= ca << o; pb[w] |= cb << o;
            if (o > 26) { pa[w + 1] |= ca >> (32 - o); pb[w + 1] |= cb >> (32 - o); }
        }
        i32x8 va, vb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { va[i] = static_cast<int>(pa[i]); vb[i] = static_cast<int>(pb[i]); }
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, acc, 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    // C/D: col = lane & 15 (B row), row = 4 (lane >> 4) + i (A row)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f = acc[i];
        if (f != static_cast<float>(static_cast<int>(f))) atomicAdd(nonint, 1);
        out[(4 * g + i) * 16 + r] = static_cast<int>(f);
    }
}

typedef int i32x6 __attribute__((ext_vector_type(6)));
template <int MODE>      // 0: fp6 16x16x128, 1: i8 16x16x64, 2: fp6 32x32x64, 3: fp6 16x16x128 through asm (accumulators in place), 4: fp4 16x16x128 asm
__global__ __launch_bounds__(256) void rate_kernel(int iters, float* out, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    i32x8 a = {lane, 1, 2, 3, 4, 5, 0, 0}, b = {3, 2, 1, lane, 7, 9, 0, 0};
    i32x4 a4 = {lane, 1, 2, 3}, b4 = {3, 2, 1, lane};
    asm volatile("" : "+v"(a), "+v"(b), "+v"(a4), "+v"(b4));
    if (iters < 0) { a4[0] = out[0] != 0.f; }
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    i32x4 e0 = {0}, e1 = {0}, e2 = {0}, e3 = {0};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if constexpr (MODE == 0) {
                d0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, d0, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                d1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, d1, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                d2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, d2, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                d3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, d3, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            } else if constexpr (MODE == 1) {
                e0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, e1, 0, 0, 0);
                e2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, e2, 0, 0, 0);
                e3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, e3, 0, 0, 0);
            } else if constexpr (MODE == 3) {
                i32x6 a6 = {a[0], a[1], a[2], a[3], a[4], a[5]}, b6 = {b[0], b[1], b[2], b[3], b[4], b[5]};
                const int sc = 0x7f7f7f7f;
#define M6(D) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:3 blgp:3" : "+v"(D) : "v"(a6), "v"(b6), "v"(sc))
                M6(d0); M6(d1); M6(d2); M6(d3);
#undef M6
            } else if constexpr (MODE == 4) {
                const int sc = 0x7f7f7f7f;
#define M4(D) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(D) : "v"(a4), "v"(b4), "v"(sc))
                M4(d0); M4(d1); M4(d2); M4(d3);
#undef M4
            } else {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = d0[0] + d1[1] + d2[2] + d3[3] + static_cast<float>(e0[0] + e1[1] + e2[2] + e3[3]) + c0[0] + c1[1] + c2[2] + c3[3];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void rate(const char* name, float* out, unsigned long long* cyc)
{
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(grid), dim3(256), 0, 0, iters, out, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(grid), dim3(256), 0, 0, iters, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
    const double macs = MODE == 1 ? 16384.0 : (MODE == 2 ? 65536.0 : 32768.0);
    const double nm = (double)grid * 4 * iters * 8;
    printf("%-36s %8.1f us  %8.1f T(FL)OPS  clk=%.0f MHz  cycles/MFMA/SIMD=%.2f\n", name, ms * 1e3, nm * macs * 2 / (ms * 1e-3) / 1e12,
           avg / (ms * 1e3), avg / ((double)iters * 8));
}

int main()
{
    int total_bad = 0;
    for (int K : {128, 1024, 4096, 11008 / 128 * 128, 32768, 262144}) {
        for (int mode = 0; mode < 3; ++mode) {       // 0 random, 1 all +7 x +7 (largest sums), 2 alternating signs of magnitude 7
            const size_t n = (size_t)16 * K;
            int8_t* ha = (int8_t*)malloc(n); int8_t* hb = (int8_t*)malloc(n);
            uint32_t s = 12345u + K + mode;
            for (size_t i = 0; i < n; ++i) {
                s = s * 1664525u + 1013904223u; const int va = (int)((s >> 8) % 16) - 8;   // weights: [-8, 7]
                s = s * 1664525u + 1013904223u; const int vb = (int)((s >> 8) % 15) - 7;   // activations: [-7, 7]
                ha[i] = mode == 0 ? va : (mode == 1 ? -8 : ((i & 1) ? 7 : -8));
                hb[i] = mode == 0 ? vb : (mode == 1 ? -7 : ((i % 3) ? 7 : -7));
            }
            int8_t *da, *db; int *dout, *dn;
            CHECK(hipMalloc(&da, n)); CHECK(hipMalloc(&db, n)); CHECK(hipMalloc(&dout, 1024)); CHECK(hipMalloc(&dn, 4));
            CHECK(hipMemcpy(da, ha, n, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, n, hipMemcpyHostToDevice));
            CHECK(hipMemset(dn, 0, 4));
            hipLaunchKernelGGL(exact_kernel, dim3(1), dim3(64), 0, 0, da, db, K, dout, dn);
            int hout[256], hn;
            CHECK(hipMemcpy(hout, dout, 1024, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hn, dn, 4, hipMemcpyDeviceToHost));
            int bad = 0; long long maxabs = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    long long ref = 0;
                    for (int k = 0; k < K; ++k) ref += (long long)ha[i * K + k] * hb[j * K + k];
                    if (ref != hout[i * 16 + j]) { if (bad < 3) printf("   mismatch [%d][%d] got %d want %lld\n", i, j, hout[i * 16 + j], ref); ++bad; }
                    if (llabs(ref) > maxabs) maxabs = llabs(ref);
                }
            printf("exactness K=%6d mode %d: %d mismatches of 256, %d non-integer accumulators, max |sum| %lld\n", K, mode, bad, hn, maxabs);
            total_bad += bad + hn;
            CHECK(hipFree(da)); CHECK(hipFree(db)); CHECK(hipFree(dout)); CHECK(hipFree(dn)); free(ha); free(hb);
        }
    }
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, 8 * 256));
    rate<1>("v_mfma_i32_16x16x64_i8", out, cyc);
    rate<0>("v_mfma_scale_f32_16x16x128 fp6 x fp6", out, cyc);
    rate<2>("v_mfma_scale_f32_32x32x64 fp6 x fp6", out, cyc);
    rate<3>("16x16x128 fp6 (E3M2) x fp6, asm in place", out, cyc);
    rate<4>("16x16x128 fp4 x fp4, asm in place", out, cyc);
    printf(total_bad ? "FP6 carrier NOT exact\n" : "FP6 carrier exact on every case\n");
    return total_bad ? 1 : 0;
}
