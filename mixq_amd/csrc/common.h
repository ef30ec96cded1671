// Shared device/host helpers for the MixQ gfx950 kernels.  CDNA4 only: wave = 64 lanes, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/mixq_hip.h"

typedef int      i32x4  __attribute__((ext_vector_type(4)));
typedef int      i32x16 __attribute__((ext_vector_type(16)));
typedef float    f32x4  __attribute__((ext_vector_type(4)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2  __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_u __attribute__((aligned(2)));   // fp16-aligned access to 4 halves

#define MIXQ_WAVE 64

static inline hipStream_t mixq_stream(mixq_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipGetLastError() after a launch; maps to the int the C ABI returns.
static inline int mixq_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MIXQ_OK : static_cast<int>(e);
}

__device__ __forceinline__ float h2f(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ uint16_t f2h(float f) {   // round-to-nearest-even
    return __half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_id_uniform() {
    return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
}
