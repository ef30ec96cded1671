// Micro-benchmark (round 6): can a compute unit pull a weight image into the 256 MB memory-side cache WITHOUT using its
// vector memory path?  Rounds 3-4 measured every vector form (one dword per 128-byte line from the GEMM's own loader waves, from a side
// stream, from a daemon kernel): all of them take their lines through the CU's L1 - the path the GEMM's k loop is bound by - and lost.
// The scalar data cache is a different path (SQ -> scalar cache -> L2 -> fabric): an s_load_dword of one dword per 128-byte line makes the L2
// fetch the whole line from HBM (which allocates it in the memory-side cache on the way), returns 64 bytes to the scalar cache and nothing to the
// L1.  A wave can keep 15 of them in flight (lgkmcnt is 4 bits; returns are out of order, which a prefetch does not care about).
//   (1) rate: bytes of lines touched per second by W waves per CU on 256 CUs walking 45 MB that nobody has read for > 256 MB of other reads;
//   (2) effect: the time of a full vector read of the same 45 MB right behind the touch pass, against the same read cold and warm.
// Forms: 0 = s_load_dword per 128-byte line, 1 = s_load_dwordx2 at byte 124 of every second line (8 bytes straddling two lines: two lines per
// request, if the scalar cache splits it), 2 = vector global_load_dword, one lane per line (the form of rounds 3-4, for reference).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_sprefetch.hip -o tools/ubench_sprefetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every wave walks a contiguous slice: bytes / (gridDim.x * waves) each, 128-byte lines
template <int FORM>
__global__ __launch_bounds__(512) void touch_kernel(const uint8_t* base, size_t bytes, int* sink)
{
    extern __shared__ uint8_t lds_pad[];                       // (160 KiB requested at launch: one workgroup per CU, like the GEMM)
    const int waves = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t nwaves = static_cast<size_t>(gridDim.x) * waves, wid = static_cast<size_t>(blockIdx.x) * waves + wave;
    const size_t lines = bytes >> 7, per = (lines + nwaves - 1) / nwaves;
    size_t l0 = wid * per, l1 = l0 + per < lines ? l0 + per : lines;
    if (l0 >= l1) return;
    if constexpr (FORM == 2) {
        int acc = 0;
        for (size_t l = l0 + lane; l < l1; l += 64) acc += *reinterpret_cast<const int*>(base + (l << 7));
        if (acc == 0x12345678) sink[0] = acc;
    } else {
        const uint8_t* p = base + (l0 << 7);
        size_t n = l1 - l0;
        // 16 requests per statement into ONE junk register pair (s[98:99], named as clobbers: returns may land at any time until the final wait)
        for (; n >= 32; n -= 32, p += 32 * 128) {
            if constexpr (FORM == 0) {
                asm volatile("s_load_dword s98, %0, 0x0\n\ts_load_dword s98, %0, 0x80\n\ts_load_dword s98, %0, 0x100\n\ts_load_dword s98, %0, 0x180\n\t"
                             "s_load_dword s98, %0, 0x200\n\ts_load_dword s98, %0, 0x280\n\ts_load_dword s98, %0, 0x300\n\ts_load_dword s98, %0, 0x380\n\t"
                             "s_load_dword s98, %0, 0x400\n\ts_load_dword s98, %0, 0x480\n\ts_load_dword s98, %0, 0x500\n\ts_load_dword s98, %0, 0x580\n\t"
                             "s_load_dword s98, %0, 0x600\n\ts_load_dword s98, %0, 0x680\n\ts_load_dword s98, %0, 0x700\n\ts_load_dword s98, %0, 0x780\n\t"
                             "s_load_dword s98, %0, 0x800\n\ts_load_dword s98, %0, 0x880\n\ts_load_dword s98, %0, 0x900\n\ts_load_dword s98, %0, 0x980\n\t"
                             "s_load_dword s98, %0, 0xa00\n\ts_load_dword s98, %0, 0xa80\n\ts_load_dword s98, %0, 0xb00\n\ts_load_dword s98, %0, 0xb80\n\t"
                             "s_load_dword s98, %0, 0xc00\n\ts_load_dword s98, %0, 0xc80\n\ts_load_dword s98, %0, 0xd00\n\ts_load_dword s98, %0, 0xd80\n\t"
                             "s_load_dword s98, %0, 0xe00\n\ts_load_dword s98, %0, 0xe80\n\ts_load_dword s98, %0, 0xf00\n\ts_load_dword s98, %0, 0xf80"
                             :: "s"(p) : "s98", "s99", "memory");
            } else {
                asm volatile("s_load_dwordx2 s[98:99], %0, 0x7c\n\ts_load_dwordx2 s[98:99], %0, 0x17c\n\ts_load_dwordx2 s[98:99], %0, 0x27c\n\ts_load_dwordx2 s[98:99], %0, 0x37c\n\t"
                             "s_load_dwordx2 s[98:99], %0, 0x47c\n\ts_load_dwordx2 s[98:99], %0, 0x57c\n\ts_load_dwordx2 s[98:99], %0, 0x67c\n\ts_load_dwordx2 s[98:99], %0, 0x77c\n\t"
                             "s_load_dwordx2 s[98:99], %0, 0x87c\n\ts_load_dwordx2 s[98:99], %0, 0x97c\n\ts_load_dwordx2 s[98:99], %0, 0xa7c\n\ts_load_dwordx2 s[98:99], %0, 0xb7c\n\t"
                             "s_load_dwordx2 s[98:99], %0, 0xc7c\n\ts_load_dwordx2 s[98:99], %0, 0xd7c\n\ts_load_dwordx2 s[98:99], %0, 0xe7c\n\ts_load_dwordx2 s[98:99], %0, 0xf7c"
                             :: "s"(p) : "s98", "s99", "memory");
            }
        }
        for (; n > 0; --n, p += 128) asm volatile("s_load_dword s98, %0, 0x0" :: "s"(p) : "s98", "s99", "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "s98", "s99", "memory");
    }
    (void)lds_pad; (void)lane;
}

// a plain, wide read of the same bytes (dwordx4 per lane, grid-strided): how fast do they arrive now?
__global__ __launch_bounds__(256) void read_kernel(const uint8_t* base, size_t bytes, int* sink)
{
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const size_t n16 = bytes >> 4, stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    i32x4 acc = {0, 0, 0, 0};
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const i32x4 v = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(base) + i);
        acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) sink[0] = acc[0];
}

// a LATENCY-bound read: one wave per workgroup, every load's address depends on the previous load's data (the buffer holds 0x01 bytes): time per
// iteration = the round trip to wherever the line lives (HBM / memory-side cache / L2).  The wide read above runs at the same 4.4 TB/s from HBM
// and from the memory-side cache - it cannot tell whether a touch pass left the image there; this one can.
__global__ __launch_bounds__(64) void chase_kernel(const uint8_t* base, size_t bytes, int iters, int* sink)
{
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const size_t n16 = bytes >> 4, stride = n16 / static_cast<size_t>(iters);
    size_t i = (static_cast<size_t>(blockIdx.x) * 64 + threadIdx.x) * 8 % stride;     // (lanes 128 bytes apart: 64 distinct lines per load)
    int acc = 0;
    for (int t = 0; t < iters; ++t) {
        const i32x4 v = *(reinterpret_cast<const i32x4*>(base) + i);
        acc += v[0];
        i += stride + static_cast<size_t>(v[1] - 0x01010101);
    }
    if (acc == 0x12345678) sink[0] = acc;
}

int main(int argc, char**)
{
    const size_t img = 45088768, copies = 10, total = img * copies;    // 11008 x 4096 int8 = the metric layer's image; 10 of them = 451 MB > 256 MB
    uint8_t* buf; int* sink;
    CHECK(hipMalloc(&buf, total)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, total)); CHECK(hipMemset(sink, 0, 64));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(touch_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(touch_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(touch_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto timed = [&](auto&& launch) { CHECK(hipEventRecord(e0, st)); launch(); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f; };
    int rot = 0;
    auto next_cold = [&]() { rot = (rot + 1) % copies; return buf + rot * img; };   // a slice last read 9 slices (406 MB) ago
    auto reader = [&](const uint8_t* p) { return timed([&] { hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, st, p, img, sink); }); };
    if (argc > 1) {
        // counter mode (rocprofv3 --pmc FETCH_SIZE: one short run): three launches of each kernel over cold slices - how many bytes does the L2 ask the
        // fabric for per touched line (the full read of 45.1 MB calibrates the counter's unit)
        for (int i = 0; i < 3; ++i) {
            reader(next_cold());
            hipLaunchKernelGGL(touch_kernel<0>, dim3(256), dim3(128), 160 * 1024, st, next_cold(), img, sink);
            hipLaunchKernelGGL(touch_kernel<1>, dim3(256), dim3(128), 160 * 1024, st, next_cold(), img, sink);
            hipLaunchKernelGGL(touch_kernel<2>, dim3(256), dim3(128), 160 * 1024, st, next_cold(), img, sink);
        }
        CHECK(hipDeviceSynchronize());
        printf("counter mode: 3 x (read_kernel, touch_kernel<0>, <1>, <2>) over 45.1 MB each\n");
        return 0;
    }
    // warm-up + the two reference reads
    for (int i = 0; i < 12; ++i) reader(next_cold());
    std::vector<float> cold, warm;
    for (int i = 0; i < 10; ++i) { const uint8_t* p = next_cold(); cold.push_back(reader(p)); warm.push_back(reader(p)); }
    printf("full vector read of one 45.1 MB image: cold (451 MB in rotation) %.2f us = %.2f TB/s, again right behind it %.2f us = %.2f TB/s\n",
           med(cold), img / med(cold) * 1e-6, med(warm), img / med(warm) * 1e-6);
    {
        // where do the lines live after each kind of pass?  chase: 256 waves x 32 dependent loads over the slice
        auto chase = [&](const uint8_t* p) { return timed([&] { hipLaunchKernelGGL(chase_kernel, dim3(256), dim3(64), 0, st, p, img, 32, sink); }) / 32.f; };
        std::vector<float> c_cold, c_warm, c_s, c_v, c_s2;
        for (int i = 0; i < 9; ++i) {
            const uint8_t* p = next_cold(); c_cold.push_back(chase(p));
            p = next_cold(); reader(p); c_warm.push_back(chase(p));
            p = next_cold(); hipLaunchKernelGGL(touch_kernel<0>, dim3(256), dim3(128), 160 * 1024, st, p, img, sink); c_s.push_back(chase(p));
            p = next_cold(); hipLaunchKernelGGL(touch_kernel<2>, dim3(256), dim3(128), 160 * 1024, st, p, img, sink); c_v.push_back(chase(p));
            p = next_cold(); hipLaunchKernelGGL(touch_kernel<1>, dim3(256), dim3(128), 160 * 1024, st, p, img, sink); c_s2.push_back(chase(p));
        }
        printf("dependent-load round trip (us per load, 256 waves x 64 lines): cold slice %.3f | behind a full read %.3f | behind the s_load_dword pass %.3f | behind the s_load_dwordx2 pass %.3f | behind the vector touch pass %.3f\n",
               med(c_cold), med(c_warm), med(c_s), med(c_s2), med(c_v));
    }
    const char* names[3] = {"s_load_dword per 128-byte line", "s_load_dwordx2 straddling two lines", "global_load_dword, one lane per line"};
    for (int form = 0; form < 3; ++form)
        for (int waves : {1, 2, 4, 8}) {
            for (int grid : {256, 232}) {
                std::vector<float> tt, tr;
                for (int i = 0; i < 9; ++i) {
                    const uint8_t* p = next_cold();
                    tt.push_back(timed([&] {
                        if (form == 0) hipLaunchKernelGGL(touch_kernel<0>, dim3(grid), dim3(64 * waves), 160 * 1024, st, p, img, sink);
                        if (form == 1) hipLaunchKernelGGL(touch_kernel<1>, dim3(grid), dim3(64 * waves), 160 * 1024, st, p, img, sink);
                        if (form == 2) hipLaunchKernelGGL(touch_kernel<2>, dim3(grid), dim3(64 * waves), 160 * 1024, st, p, img, sink);
                    }));
                    tr.push_back(reader(p));
                }
                const float t = med(tt), r = med(tr);
                printf("%-40s %d waves x %3d workgroups: touch pass %7.2f us = %5.2f TB/s of lines; full read behind it %.2f us = %.2f TB/s\n", names[form], waves, grid, t, img / t * 1e-6, r, img / r * 1e-6);
            }
        }
    CHECK(hipDeviceSynchronize());
    return 0;
}
