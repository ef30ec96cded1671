// Micro-benchmark (round 6): what does ONE consumer wave per SIMD pay for the memory instructions it issues between its MFMAs?
// The k loops of gemm_wreg.hip run 24 int8 MFMAs (16 cycles each: 384) per k-step and wave in 530 cycles; the feed ablations price a weight load
// (global_load_dwordx4, 1 KiB per wave) at ~39 cycles of the k-step and a fragment read (ds_read_b128) at ~13 (NOTEBOOK.md round 6).  This kernel is
// that k-step without its memory SYSTEM - the loads hit one hot KiB per wave in the L2 / L1, the LDS reads one block, the waits are a k-step old -
// so what is left is the issue cost: four waves per workgroup (one per SIMD), 256 workgroups, `iters` k-steps of
//   8 groups x 3 MFMAs (24 independent accumulators), with per-k-step extras selected by FORM:
//   0  nothing (MFMA floor)                          1  8 x ds_read_b128, one behind each group's last MFMA (the activation path)
//   2  1 + 3 x global_load_dwordx4 (saddr + voffset) behind the first MFMA of groups 1, 4, 6 (the shipped order of rounds 2-5)
//   3  1 + 3 x buffer_load_dwordx4 (offen + soffset), same places (round 6's form)
//   4  1 + 6 x buffer_load_dwordx2 (the fragment's two 8-byte halves in two different gaps)
//   5  1 + 6 x buffer_load_dwordx4 under a half exec mask (lanes 0-31 / 32-63 in two different gaps; exec restored in front of the next MFMA)
//   6  1 + 3 x ds_read_b128 more (the weights out of LDS instead)
//   7  3 x buffer_load_dwordx4 only (no LDS reads)       8  1 + 3 loads all behind ONE group's MFMAs (groups 1 only: back to back)
//   9  1 + 3 x buffer_load_dwordx4 each behind the LAST MFMA of its group (next to the group's ds_read)
// Prints shader cycles per k-step (median over workgroups of wave 0's s_memtime span / iters).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int FORM>
__global__ __launch_bounds__(256) void kstep_kernel(const uint8_t* w, int iters, int* out, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[16 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16 * 1024 / 4; i += 256) reinterpret_cast<int*>(lds)[i] = i * 0x9E3779B1;
    __syncthreads();
    i32x4 acc[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[j][i] = i32x4{0, 0, 0, 0};
    i32x4 xf[8], wq[2][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xf[j] = i32x4{lane * 0x01010101, j, lane ^ j, 7}; asm volatile("" : "+v"(xf[j])); }
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 3; ++i) { wq[d][i] = i32x4{lane, i, d, lane * 3}; asm volatile("" : "+v"(wq[d][i])); }
    const unsigned lbase = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) uint8_t*)lds)) + lane * 16;
    const uint8_t* wsrc = w + (blockIdx.x * 4 + wave) * 4096;                    // 4 KiB per wave: hot after the first k-step
    const i32x4 rs = {static_cast<int>(reinterpret_cast<size_t>(w)), static_cast<int>((reinterpret_cast<size_t>(w) >> 32) & 0xffff), 1 << 30, 0x00020000};
    const int vo = (blockIdx.x * 4 + wave) * 4096 + lane * 16, vo8 = (blockIdx.x * 4 + wave) * 4096 + lane * 8, l16 = lane * 16;
    unsigned so = 0;
    unsigned long long t0 = 0, t1 = 0;
    auto wload = [&](int d, int i, int part) __attribute__((always_inline)) {
        i32x4& dst = wq[d][i];
        if constexpr (FORM == 2) { if (part == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(l16), "s"(wsrc + i * 1024) : "memory"); }
        if constexpr (FORM == 3 || FORM == 7 || FORM == 8 || FORM == 9) { if (part == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(dst) : "v"(vo), "s"(rs), "s"(so), "i"(i * 1024) : "memory"); }
        if constexpr (FORM == 4) {
            i32x2 h = {dst[part * 2], dst[part * 2 + 1]};
            asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4" : "+v"(h) : "v"(vo8), "s"(rs), "s"(so), "i"(i * 1024 + part * 512) : "memory");
            dst[part * 2] = h[0]; dst[part * 2 + 1] = h[1];
        }
        if constexpr (FORM == 5) {
            if (part == 0) asm volatile("s_mov_b32 exec_hi, 0\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_mov_b32 exec_hi, -1" : "+v"(dst) : "v"(vo), "s"(rs), "s"(so), "i"(i * 1024) : "memory");
            else asm volatile("s_mov_b32 exec_lo, 0\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_mov_b32 exec_lo, -1" : "+v"(dst) : "v"(vo), "s"(rs), "s"(so), "i"(i * 1024) : "memory");
        }
        if constexpr (FORM == 6) { if (part == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(lbase), "i"(8192 + i * 1024) : "memory"); }
    };
    auto kstep = [&](auto d_c) __attribute__((always_inline)) {
        constexpr int d = decltype(d_c)::value;                                   // (ring slots are compile-time registers, as in the real loop)
        // the k-step's waits: everything requested a k-step ago
        if constexpr (FORM >= 2 && FORM != 6) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]), "+v"(wq[d ^ 1][0]), "+v"(wq[d ^ 1][1]), "+v"(wq[d ^ 1][2]) : "i"(FORM == 4 || FORM == 5 ? 6 : 3));
        if constexpr (FORM != 0 && FORM != 7)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]), "+v"(xf[6]), "+v"(xf[7]));
        if constexpr (FORM == 6) asm volatile("" : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[d][0], xf[j], acc[j][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (FORM != 9) {
                if (FORM == 8) { if (j == 1) { wload(d ^ 1, 0, 0); wload(d ^ 1, 1, 0); wload(d ^ 1, 2, 0); } }
                else {
                    if (j == 1) wload(d ^ 1, 0, 0);
                    if (j == 4) wload(d ^ 1, 1, 0);
                    if (j == 6) wload(d ^ 1, 2, 0);
                    if (j == 2) wload(d ^ 1, 0, 1);
                    if (j == 5) wload(d ^ 1, 1, 1);
                    if (j == 7) wload(d ^ 1, 2, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[j][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[d][1], xf[j], acc[j][1], 0, 0, 0);
            acc[j][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[d][2], xf[j], acc[j][2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FORM != 0 && FORM != 7) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(xf[j]) : "v"(lbase), "i"(j * 1024) : "memory");
            if (FORM == 9) { if (j == 1) wload(d ^ 1, 0, 0); if (j == 4) wload(d ^ 1, 1, 0); if (j == 6) wload(d ^ 1, 2, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
        so = (so + 1024) & 2047;                                                 // (a moving scalar offset, as the real loop's)
    };
    for (int it = -2; it < iters; it += 2) {
        if (it == 0) t0 = __builtin_readcyclecounter();
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
    }
    t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(wq[0][0]), "+v"(wq[0][1]), "+v"(wq[0][2]), "+v"(wq[1][0]), "+v"(wq[1][1]), "+v"(wq[1][2]));
    asm volatile("" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]), "+v"(xf[6]), "+v"(xf[7]));
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) s += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FORM> static double run(const uint8_t* w, int* out, unsigned long long* cyc, int iters)
{
    std::vector<unsigned long long> h(256);
    std::vector<double> meds;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(kstep_kernel<FORM>, dim3(256), dim3(256), 0, 0, w, iters, out, cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        meds.push_back(static_cast<double>(h[128]) / iters);
    }
    std::sort(meds.begin(), meds.end());
    return meds[2];
}

int main()
{
    uint8_t* w; int* out; unsigned long long* cyc;
    CHECK(hipMalloc(&w, 8 << 20)); CHECK(hipMemset(w, 3, 8 << 20)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, 256 * 8));
    const int iters = 2000;
    const char* names[10] = {"24 MFMAs only", "+ 8 ds_read_b128 (activation fragments)", "+ 8 ds_read + 3 global_load_dwordx4 (rounds 2-5)", "+ 8 ds_read + 3 buffer_load_dwordx4 (round 6)",
                             "+ 8 ds_read + 6 buffer_load_dwordx2 (halves in two gaps)", "+ 8 ds_read + 6 half-exec buffer_load_dwordx4", "+ 8 ds_read + 3 ds_read_b128 (weights from LDS)",
                             "+ 3 buffer_load_dwordx4 only", "+ 8 ds_read + 3 buffer_load_dwordx4 back to back in one gap", "+ 8 ds_read + 3 buffer_load_dwordx4 behind their group's last MFMA"};
    double r[10];
    r[0] = run<0>(w, out, cyc, iters); r[1] = run<1>(w, out, cyc, iters); r[2] = run<2>(w, out, cyc, iters); r[3] = run<3>(w, out, cyc, iters); r[4] = run<4>(w, out, cyc, iters);
    r[5] = run<5>(w, out, cyc, iters); r[6] = run<6>(w, out, cyc, iters); r[7] = run<7>(w, out, cyc, iters); r[8] = run<8>(w, out, cyc, iters); r[9] = run<9>(w, out, cyc, iters);
    printf("s_memtime ticks per k-step of 24 v_mfma_i32_16x16x64_i8 (one wave per SIMD, 256 workgroups; the counter runs at the nominal 2.4 GHz - 24 MFMAs of 16 shader cycles read 468 at the ~1.97 GHz the part sustains):\n");
    for (int f = 0; f < 10; ++f) printf("  form %d  %-66s %8.1f  (+%.1f over the MFMAs)\n", f, names[f], r[f], r[f] - r[0]);
    return 0;
}
