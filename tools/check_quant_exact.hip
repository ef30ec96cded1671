// Exhaustive check of quant_exact (mixq_amd/csrc/common.h) against the division form it replaces,
//     q = clamp(rint(RN_fp32(x / s)), +-QMAX),
// over ALL pairs (x, s) of finite fp16 values with s > 0: 63488 x 31743 ~ 2 x 10^9 pairs per bit width.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/check_quant_exact.hip -o tools/check_quant_exact
#include "../mixq_amd/csrc/common.h"
#include <stdio.h>

template <int BIT>
__device__ __forceinline__ int quant_div(float x, float s) {
    constexpr float QMAX = static_cast<float>((1 << (BIT - 1)) - 1);
    float q = (s > 0.f) ? rintf(__fdiv_rn(x, s)) : 0.f;
    q = fminf(fmaxf(q, -QMAX), QMAX);
    return static_cast<int>(q);
}

template <int BIT>
__global__ void check(unsigned long long* bad, unsigned* first)
{
    const unsigned sb = blockIdx.x + 1;                       // s bits 0x0001 .. 0x7bff (positive finite, incl. denormals)
    const float s = h2f(static_cast<uint16_t>(sb));
    const float rs = __fdiv_rn(1.0f, s);
    unsigned long long n = 0;
    for (unsigned xb = threadIdx.x; xb < 65536u; xb += blockDim.x) {
        if ((xb & 0x7c00u) == 0x7c00u) continue;              // inf / nan
        const float x = h2f(static_cast<uint16_t>(xb));
        const int a = quant_exact<BIT>(x, s, rs), b = quant_div<BIT>(x, s);
        if (a != b) { ++n; atomicMin(first, (sb << 16) | xb); }
    }
    if (n) atomicAdd(bad, n);
}

int main()
{
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    for (int bit : {8, 4}) {
        unsigned long long z = 0; unsigned f = 0xffffffffu;
        hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice); hipMemcpy(first, &f, 4, hipMemcpyHostToDevice);
        if (bit == 8) hipLaunchKernelGGL(check<8>, dim3(0x7bff), dim3(256), 0, 0, bad, first);
        else          hipLaunchKernelGGL(check<4>, dim3(0x7bff), dim3(256), 0, 0, bad, first);
        hipDeviceSynchronize();
        hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
        printf("bit %d: %llu mismatches over all finite (x, s > 0) fp16 pairs", bit, z);
        if (z) printf("; first at s=0x%04x x=0x%04x", f >> 16, f & 0xffff);
        printf("\n");
    }
    return 0;
}
