// Micro-benchmark: what a quantise -> GEMM hand-off INSIDE one kernel would cost on gfx950 (the question behind fusing the forward's two
// launches: 5.4 us + 0.5-0.9 us gap + 24.8 us today).  232 co-resident workgroups of 384 threads, as the metric launch.  Per round:
//   phase 0  every workgroup writes its 1/232 of a 2 MiB "quantised activation" (a value that depends on the round and the address)
//            with write-through (sc1) 16-byte stores, drains them, and arrives at a grid-wide counter;
//   (MODE 3: phase 0 writes nothing - the cost of the arrival alone, and of the slab read when the lines are L2-resident)
//   sync     one lane polls the counter (relaxed, agent scope); then, by MODE, 0: nothing, 1: an agent-scope acquire fence (buffer_inv sc1:
//            drops the non-coherent lines of this XCD's L2), 2: nothing, but phase 1 reads with sc1 loads;
//   phase 1  every workgroup reads the 512 KiB slab of "its" M tile (blockIdx % 4) - mostly lines written on OTHER XCDs, which this XCD's
//            L2 may still hold from the previous round - and counts the words that are not this round's;
//   a second grid-wide arrival closes the round (nobody overwrites what a neighbour still reads).
// Reported per MODE: stale words seen (correctness of the protocol), and the median over rounds of (a) first store -> past the sync,
// (b) the slab read, from the 100 MHz wall clock.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_gridsync.hip -o tools/ubench_gridsync
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int G = 232, NT = 384, ROUNDS = 24, XBYTES = 2 << 20, SLAB = XBYTES / 4;
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int word_of(int r, int idx16) { return static_cast<unsigned int>(r) * 0x9E3779B1u + static_cast<unsigned int>(idx16); }

template <int MODE>
__global__ __launch_bounds__(NT) void gridsync_kernel(uint4* x, unsigned int* counter, unsigned long long* t_sync, unsigned long long* t_read,
                                                      unsigned int* stale)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(x, 0, XBYTES, 0x00020000);
    constexpr int PIECES = XBYTES / 16;                                     // 16-byte pieces of the activation
    const int p0 = static_cast<int>(static_cast<long long>(PIECES) * b / G), p1 = static_cast<int>(static_cast<long long>(PIECES) * (b + 1) / G);
    unsigned int bad = 0;
    auto arrive_and_wait = [&](unsigned int target) {
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    };
    for (int r = 1; r <= ROUNDS; ++r) {
        const unsigned long long t0 = wall_clock64();
        for (int p = p0 + tid; p < (MODE == 3 ? p0 : p1); p += NT) {
            const unsigned int w = word_of(r, p);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4v{w, w ^ 1u, w ^ 2u, w ^ 3u}, rs, p * 16, 0, 16 /* sc1: write-through */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        arrive_and_wait(static_cast<unsigned int>(G) * (2 * r - 1));
        if constexpr (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const unsigned long long t1 = wall_clock64();
        const int s0 = (b & 3) * (SLAB / 16);
        for (int p = tid; p < SLAB / 16; p += NT) {
            u32x4v v;
            if constexpr (MODE == 2) v = __builtin_amdgcn_raw_buffer_load_b128(rs, (s0 + p) * 16, 0, 16 /* sc1 */);
            else v = __builtin_amdgcn_raw_buffer_load_b128(rs, (s0 + p) * 16, 0, 0);
            const unsigned int w = word_of(r, s0 + p);
            if constexpr (MODE != 3) bad += (v[0] != w) + (v[1] != (w ^ 1u)) + (v[2] != (w ^ 2u)) + (v[3] != (w ^ 3u));
            else bad += v[0] == 0x12345678u;
        }
        const unsigned long long t2 = wall_clock64();
        if (tid == 0) { t_sync[(r - 1) * G + b] = t1 - t0; t_read[(r - 1) * G + b] = t2 - t1; }
        arrive_and_wait(static_cast<unsigned int>(G) * (2 * r));
    }
    if (bad) atomicAdd(stale, bad);
}

template <int MODE> void run(const char* label, uint4* x, unsigned int* counter, unsigned long long* ts, unsigned long long* tr, unsigned int* stale)
{
    CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(stale, 0, 4)); CHECK(hipMemset(x, 0, XBYTES));
    hipLaunchKernelGGL(gridsync_kernel<MODE>, dim3(G), dim3(NT), 0, 0, x, counter, ts, tr, stale);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs(ROUNDS * G), hr(ROUNDS * G);
    unsigned int hstale = 0;
    CHECK(hipMemcpy(hs.data(), ts, hs.size() * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hr.data(), tr, hr.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&hstale, stale, 4, hipMemcpyDeviceToHost));
    // per round: the slowest workgroup (what a fused kernel would wait for); then the median over rounds 4..
    std::vector<double> smax, rmax, rmed;
    for (int r = 4; r < ROUNDS; ++r) {
        smax.push_back(*std::max_element(hs.begin() + r * G, hs.begin() + (r + 1) * G) * 0.01);
        std::vector<unsigned long long> q(hr.begin() + r * G, hr.begin() + (r + 1) * G);
        std::sort(q.begin(), q.end());
        rmax.push_back(q.back() * 0.01); rmed.push_back(q[G / 2] * 0.01);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-58s stale words %10u   write+sync (slowest workgroup) %6.2f us   slab read median %6.2f / slowest %6.2f us\n", label, hstale, med(smax), med(rmed), med(rmax));
}

int main()
{
    uint4* x; unsigned int *counter, *stale; unsigned long long *ts, *tr;
    CHECK(hipMalloc(&x, XBYTES)); CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&stale, 4));
    CHECK(hipMalloc(&ts, ROUNDS * G * 8)); CHECK(hipMalloc(&tr, ROUNDS * G * 8));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("sc1 stores, counter, plain loads (no acquire)", x, counter, ts, tr, stale);
        run<1>("sc1 stores, counter, agent acquire fence, plain loads", x, counter, ts, tr, stale);
        run<2>("sc1 stores, counter, sc1 loads", x, counter, ts, tr, stale);
        run<3>("no stores: the grid-wide arrival alone (slab unchanged: L2-hot read)", x, counter, ts, tr, stale);
    }
    return 0;
}
