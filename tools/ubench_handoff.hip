// Micro-benchmark: what a pairwise split-K hand-off costs on gfx950.  256 workgroups in pairs; the producer of a pair publishes a 64 KiB int32
// partial tile (128 x 128: 256 threads x 16 x 16 bytes) with a release at agent scope and raises a flag; its partner spins on the flag, acquires
// and reads the 64 KiB back (sums it).  Reported: time from the producer's first store to the consumer's last load, from the common 100 MHz
// clock (s_memtime / wall_clock64), median over the pairs and rounds, for partners on the SAME XCD (blockIdx ^ 8) and on DIFFERENT XCDs
// (blockIdx ^ 1: consecutive workgroups go round-robin over the 8 XCDs), and for the consumer alone re-reading data it did not have to wait for.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_handoff.hip -o tools/ubench_handoff
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROUNDS = 16, TILE_BYTES = 65536;

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// MODE 0: write-through (sc1) 16-byte stores, drained, relaxed agent-scope flag; the partner polls the flag and reads with sc1 loads - the
//         protocol of gemm_sk.hip (no L2 write-back / invalidate instructions at all)
// MODE 1: ordinary stores + release fence (buffer_wbl2) / acquire fence (buffer_inv) around the flag
template <int MODE>
__global__ __launch_bounds__(256) void handoff_kernel(uint4* ws, unsigned int* flags, unsigned long long* t_pub, unsigned long long* result, int xor_mask,
                                                      unsigned int* sink)
{
    const int b = blockIdx.x, partner = b ^ xor_mask, pair = b < partner ? b : partner;
    const bool producer = b < partner;
    uint4* tile = ws + static_cast<size_t>(pair) * (TILE_BYTES / 16);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile, 0, TILE_BYTES, 0x00020000);
    unsigned int acc = 0;
    for (int r = 1; r <= ROUNDS; ++r) {
        if (producer) {
            const unsigned long long t0 = wall_clock64();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const u32x4v v = {static_cast<unsigned int>(r), static_cast<unsigned int>(b), static_cast<unsigned int>(i), threadIdx.x};
                if constexpr (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (i * 256 + threadIdx.x) * 16, 0, 16 /* sc1 */);
                else tile[i * 256 + threadIdx.x] = make_uint4(v[0], v[1], v[2], v[3]);
            }
            if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_store(t_pub + pair, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flags + pair, static_cast<unsigned int>(r), MODE == 0 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(flags + 512 + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != static_cast<unsigned int>(r)) __builtin_amdgcn_s_sleep(4);
            }
            __syncthreads();
        } else {
            if (threadIdx.x == 0) while (__hip_atomic_load(flags + pair, MODE == 0 ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != static_cast<unsigned int>(r)) __builtin_amdgcn_s_sleep(1);
            __syncthreads();
            if constexpr (MODE == 1) __threadfence();
            u32x4v v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (MODE == 0) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (i * 256 + threadIdx.x) * 16, 0, 16 /* sc1 */);
                else { const uint4 q = tile[i * 256 + threadIdx.x]; v[i] = u32x4v{q.x, q.y, q.z, q.w}; }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += v[i][0] + v[i][3];
            if (acc == 0x7fffffffu) *sink = acc;                            // (the sum is needed: the loads must have landed)
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned long long t1 = wall_clock64();
                const unsigned long long tp = __hip_atomic_load(reinterpret_cast<unsigned long long*>(t_pub) + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                result[(static_cast<size_t>(pair) * ROUNDS + (r - 1)) * 2] = t1 - tp;
                result[(static_cast<size_t>(pair) * ROUNDS + (r - 1)) * 2 + 1] = t1;
            }
            u32x4v w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (i * 256 + threadIdx.x) * 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += w[i][1];
            if (acc == 0x7ffffffeu) *sink = acc;
            __syncthreads();
            if (threadIdx.x == 0) {
                result[(static_cast<size_t>(pair) * ROUNDS + (r - 1)) * 2 + 1] = wall_clock64() - result[(static_cast<size_t>(pair) * ROUNDS + (r - 1)) * 2 + 1];
                __hip_atomic_store(flags + 512 + pair, static_cast<unsigned int>(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (acc == 0x12345u) *sink = acc;
}

int main()
{
    uint4* ws; unsigned int *flags, *sink; unsigned long long *t_pub, *result;
    CHECK(hipMalloc(&ws, 256 * TILE_BYTES)); CHECK(hipMalloc(&flags, 1024 * 4)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&t_pub, 256 * 8)); CHECK(hipMalloc(&result, 256 * ROUNDS * 16));
    for (int mode = 0; mode < 2; ++mode)
    for (int mask : {8, 1, 128}) {
        CHECK(hipMemset(flags, 0, 1024 * 4)); CHECK(hipMemset(result, 0, 256 * ROUNDS * 16));
        if (mode == 0) hipLaunchKernelGGL(handoff_kernel<0>, dim3(256), dim3(256), 0, 0, ws, flags, t_pub, result, mask, sink);
        else           hipLaunchKernelGGL(handoff_kernel<1>, dim3(256), dim3(256), 0, 0, ws, flags, t_pub, result, mask, sink);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(256 * ROUNDS * 2);
        CHECK(hipMemcpy(h.data(), result, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> a, c;
        for (int p = 0; p < 256; ++p)
            for (int r = 2; r < ROUNDS; ++r) {                              // (first rounds: cold code and TLB)
                const unsigned long long v = h[(static_cast<size_t>(p) * ROUNDS + r) * 2], w = h[(static_cast<size_t>(p) * ROUNDS + r) * 2 + 1];
                if (v) { a.push_back(v / 100.0); c.push_back(w / 100.0); }
            }
        std::sort(a.begin(), a.end()); std::sort(c.begin(), c.end());
        printf("%s, partner = blockIdx ^ %3d (%s): publish 64 KiB + flag + acquire + read back: median %.2f us  p10 %.2f  p90 %.2f   | warm re-read of the 64 KiB alone: median %.2f us   (%zu samples)\n",
               mode == 0 ? "sc1 stores / loads" : "plain + fences    ", mask, mask == 8 ? "same XCD" : (mask == 1 ? "neighbouring XCDs" : "same XCD, far"), a[a.size() / 2], a[a.size() / 10], a[a.size() * 9 / 10], c[c.size() / 2], a.size());
    }
    return 0;
}
