// Micro-benchmark: what the "launch floor" of a dependent graph node is made of on gfx950.  The metric GEMM (232 workgroups x 384 threads,
// 144 KiB of dynamic LDS, 248 VGPRs, a 232-byte by-value argument block) returning at entry costs 2.17 us per launch in a chain of 20
// (profiles/r03_gemm_ab.txt, abl9_empty) - 8 % of the launch, paid twice per forward.  This sweeps the footprint of an EMPTY kernel
// (grid, threads, dynamic LDS, registers, argument bytes) in one hipGraph of 20 dependent launches each, replayed round-robin.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_launch.hip -o tools/ubench_launch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <string>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
struct Small { int* p; int flag; };
struct Big { int* p; int flag; long long pad[27]; };                            // 232 bytes, as WrArgs

extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

// REGS = 0: a handful of registers; 1: 248 VGPRs (an asm clobber of v247 makes the allocation as big as the GEMM's)
template <int NT, int REGS, class A>
__global__ __launch_bounds__(NT) void empty_kernel(const A a)
{
    if (a.flag != 12345) return;
    if constexpr (REGS) asm volatile("v_mov_b32 v247, 0" ::: "v247");
    lds[threadIdx.x] = 1;
    a.p[threadIdx.x] = lds[threadIdx.x ^ 1];
}
// the same, but every workgroup stores 48 KiB (the GEMM's output tile) with nt stores before it returns: what the END of a storing kernel adds
template <int NT>
__global__ __launch_bounds__(NT) void store_kernel(u32x4v* y, int flag)
{
    if (flag == 12345) return;
    u32x4v v = {threadIdx.x, blockIdx.x, 1u, 2u};
    u32x4v* base = y + static_cast<size_t>(blockIdx.x) * 3072;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v, base + i * NT + threadIdx.x);
}

struct Case { std::string name; hipGraphExec_t exec; };

template <class F>
static hipGraphExec_t capture(hipStream_t st, int launches, F&& launch)
{
    hipGraph_t g; hipGraphExec_t e;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < launches; ++i) launch();
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    return e;
}

int main()
{
    hipStream_t st; CHECK(hipStreamCreate(&st));
    int* p; CHECK(hipMalloc(&p, 4096));
    u32x4v* y; CHECK(hipMalloc(&y, 256 * 3072 * 16));
    const int L = 20;
    std::vector<Case> cases;
    Small s{p, 0}; Big b{p, 0, {0}};
#define ADD(NAME, KERN, GRID, NT, LDS, ARG)                                                                             \
    do {                                                                                                                \
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        cases.push_back({NAME, capture(st, L, [&] { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(NT), LDS, st, ARG); })}); \
    } while (0)
    ADD("g232 t384 lds144K r248 arg232 (the GEMM's footprint)", (empty_kernel<384, 1, Big>), 232, 384, 144 * 1024, b);
    ADD("g232 t384 lds144K r248 arg16", (empty_kernel<384, 1, Small>), 232, 384, 144 * 1024, s);
    ADD("g232 t384 lds144K r8   arg232", (empty_kernel<384, 0, Big>), 232, 384, 144 * 1024, b);
    ADD("g232 t384 lds1K   r248 arg232", (empty_kernel<384, 1, Big>), 232, 384, 1024, b);
    ADD("g232 t384 lds64K  r248 arg232", (empty_kernel<384, 1, Big>), 232, 384, 64 * 1024, b);
    ADD("g232 t384 lds1K   r8   arg16", (empty_kernel<384, 0, Small>), 232, 384, 1024, s);
    ADD("g232 t256 lds1K   r8   arg16", (empty_kernel<256, 0, Small>), 232, 256, 1024, s);
    ADD("g232 t64  lds1K   r8   arg16", (empty_kernel<64, 0, Small>), 232, 64, 1024, s);
    ADD("g232 t256 lds144K r248 arg232", (empty_kernel<256, 1, Big>), 232, 256, 144 * 1024, b);
    ADD("g58  t384 lds144K r248 arg232", (empty_kernel<384, 1, Big>), 58, 384, 144 * 1024, b);
    ADD("g512 t512 lds1K   r8   arg16 (the quantiser's grid)", (empty_kernel<512, 0, Small>), 512, 512, 1024, s);
    ADD("g8   t64  lds1K   r8   arg16", (empty_kernel<64, 0, Small>), 8, 64, 1024, s);
    cases.push_back({"g232 t384 nt stores of 48 KiB per workgroup, no LDS", capture(st, L, [&] { hipLaunchKernelGGL(store_kernel<384>, dim3(232), dim3(384), 0, st, y, 0); })});
    cases.push_back({"g232 t384 the same kernel returning at entry", capture(st, L, [&] { hipLaunchKernelGGL(store_kernel<384>, dim3(232), dim3(384), 0, st, y, 12345); })});
    const int ROUNDS = 40;
    std::vector<std::vector<float>> t(cases.size());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < ROUNDS + 3; ++r)
        for (size_t c = 0; c < cases.size(); ++c) {
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(cases[c].exec, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) t[c].push_back(ms * 1e3f / L);
        }
    printf("empty-kernel launch floor, %d dependent launches per graph, %d interleaved rounds; us per launch median / min\n", L, ROUNDS);
    for (size_t c = 0; c < cases.size(); ++c) {
        std::sort(t[c].begin(), t[c].end());
        printf("  %-62s %6.2f %6.2f\n", cases[c].name.c_str(), t[c][t[c].size() / 2], t[c][0]);
    }
    return 0;
}
