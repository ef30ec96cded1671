// Shared device/host helpers for the MixQ gfx950 kernels.  CDNA4 only: wave = 64 lanes, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include "../../include/mixq_hip.h"

typedef int      i32x4  __attribute__((ext_vector_type(4)));
typedef int      i32x2  __attribute__((ext_vector_type(2)));
typedef int      i32x6  __attribute__((ext_vector_type(6)));
typedef int      i32x8  __attribute__((ext_vector_type(8)));
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

// Tuning state (a forced tile configuration, a forced launch geometry) is kept PER DEVICE: one slot per HIP device, indexed by the
// device that is current in the calling thread - a process that drives several GPUs ("one process, per-device streams") can force
// a configuration on one of them without touching launches on the others.
inline int mixq_cur_dev() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d & 63;
}
struct MixqDevInt {
    std::atomic<int> v[64];
    explicit MixqDevInt(int init) { for (auto& x : v) x.store(init, std::memory_order_relaxed); }
    int get() const { return v[mixq_cur_dev()].load(std::memory_order_relaxed); }
    void set(int x) { v[mixq_cur_dev()].store(x, std::memory_order_relaxed); }
};

// hipGetLastError() after a launch; maps to the int the C ABI returns.
static inline int mixq_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MIXQ_OK : static_cast<int>(e);
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device): hipFuncSetAttribute is per device, and one process may drive
// several GPUs (north_star: "one process with per-device streams" is a legal variant), so the "already done" note is too.
inline int mixq_ensure_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, uint64_t> done;        // kernel -> bit mask of devices (< 64) already configured
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return MIXQ_ENODEV; }
    const uint64_t bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lock(mu);
    uint64_t& m = done[kernel];
    if (m & bit) return MIXQ_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e != hipSuccess) return static_cast<int>(e);
    m |= bit;
    return MIXQ_OK;
}

__device__ __forceinline__ float h2f(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ uint16_t f2h(float f) {   // round-to-nearest-even
    return __half_as_ushort(__float2half_rn(f));
}
// Maximum over the 64 lanes of a NON-NEGATIVE value, returned in every lane.  DPP moves inside the 16-lane rows (quad
// swaps, half mirror, mirror: 4 VALU) and four v_readlane for the rows, instead of six ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {      // lanes without a source read 0: neutral for a max of values >= 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));          // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));          // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));         // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));         // row_mirror: every lane of a row now holds the row maximum
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// x_scale = fp16(amax / qmax) (include/mixq_hip.h: fp32 divide, RNE to fp16) WITHOUT the division: for every finite fp16 maximum and both
// qmax (127, 7), fp16(RN(amax * RN(1 / qmax))) is the same fp16 value (checked exhaustively: tests/test_oracle_golden.py) - one multiply
// instead of a ~10-instruction IEEE division on every row's critical path (maximum -> scale -> quantise).
__device__ __forceinline__ uint16_t mixq_row_scale(float amax, float qmax) { return f2h(amax * (1.0f / qmax)); }
// 1 / s for the per-value products of quant_exact / quant8_exact below: the IEEE division.  (v_rcp_f32, 1 ulp, passes the exhaustive self-test
// too - the exactness argument below has the slack: t within 3.5e-5 of x / s, needed < 6e-5 - and changed no timing in round 5: not taken.)
__device__ __forceinline__ float mixq_rcp_scale(float s) { return s > 0.f ? __fdiv_rn(1.0f, s) : 0.f; }

// q = clamp(rint(x / s), +-QMAX) with the fp32 IEEE quotient rounded half-to-even (the convention of the oracle's
// find_row_scale), WITHOUT a division per element.  x is an fp16 value, s an fp16 scale (both exact in fp32), rs = 1/s.
// Why it is exact: where it matters |x/s| <= 128, and x/s = (a/b) 2^E with 11-bit a, b lies either exactly on a half-integer or at
// least 1/(4b) > 6e-5 away from one, while t = x * rs is within 2e-5 of x/s - so rint(t) is the correctly rounded quotient
// unless x/s is an exact tie; the tie is detected from the exact residual e = x - q0 s (|q0| s has <= 18 significant bits) and
// resolved to even.  The same argument shows rint(RN(x/s)) == rint_half_even(x/s), i.e. this equals the division form bit
// for bit.
template <int BIT>
__device__ __forceinline__ int quant_exact(float x, float s, float rs) {
    constexpr float QMAX = static_cast<float>((1 << (BIT - 1)) - 1);
    float t = x * rs;
    t = fminf(fmaxf(t, -2.f * QMAX), 2.f * QMAX);     // far-out values (denormal scales) clamp anyway; keeps q0 s exact
    float q0 = rintf(t);
    const float e = fmaf(-q0, s, x);
    if (fabsf(e) * 2.f == s && (static_cast<int>(q0) & 1)) q0 += copysignf(1.f, e);
    q0 = fminf(fmaxf(q0, -QMAX), QMAX);
    return (s > 0.f) ? static_cast<int>(q0) : 0;
}
// The same quotient for the EIGHT values of a 16-byte chunk, at a third of the instructions (the quantise passes spend 0.7 us of a 5 us
// launch at K = 4096, 2.2 us at K = 11008, on this arithmetic: profiles/r05_quant_probe.txt).  Per value: t = x rs clamped to +-QMAX FIRST
// (the bounds are integers: clamp and round commute, and the tie rule below cannot carry a clamped value past them), u = t + 1.5 2^23 - the
// add IS the round-half-even to an integer, and u's low mantissa bits are the two's-complement result: no rint, no float -> int
// conversion, no "& 0xff" - q0 = u - 1.5 2^23 (exact), e = x - q0 s (exact, as above), and the exact-tie test |e| == s / 2.  Ties are rare
// (x / s must be a half-integer exactly: ~1e-5 per value on real activations), so their fix-up - move an ODD q0 to its even neighbour on
// e's side, never past +-QMAX - sits behind ONE wave-uniform branch for the chunk.  Returns the bit patterns of u: byte 0 is the int8
// value (low nibble: the int4 value), quant8_int() the integer.  Equal to quant_exact bit for bit: mixq_selftest_quant_exact checks both
// against the division form over every finite fp16 pair.
constexpr float MIXQ_RMAGIC = 12582912.f;                                     // 1.5 * 2^23 = 0x4B400000
__device__ __forceinline__ int quant8_int(uint32_t ubits) { return static_cast<int>(ubits & 0x7fffffu) - 0x400000; }
template <int BIT>
__device__ __forceinline__ void quant8_exact(const uint4& v, float s, float rs, uint32_t (&ub)[8]) {
    constexpr float QMAX = static_cast<float>((1 << (BIT - 1)) - 1);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    const float hs = 0.5f * s;
    float u[8], e[8];
    bool tie = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = h2f(static_cast<uint16_t>((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu)));
        const float t = __builtin_amdgcn_fmed3f(x * rs, -QMAX, QMAX);
        u[j] = t + MIXQ_RMAGIC;
        e[j] = fmaf(-(u[j] - MIXQ_RMAGIC), s, x);
        tie |= fabsf(e[j]) == hs;
    }
    if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {                            // (wave-uniform; s == 0: x == 0 here, u is even, nothing moves)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (fabsf(e[j]) == hs && (__float_as_uint(u[j]) & 1u))
                u[j] = __builtin_amdgcn_fmed3f(u[j] + copysignf(1.f, e[j]), MIXQ_RMAGIC - QMAX, MIXQ_RMAGIC + QMAX);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) ub[j] = __float_as_uint(u[j]);
}
// bytes 0 of four such patterns as one dword (v_perm_b32: two selects and an or instead of masks and shifts)
__device__ __forceinline__ uint32_t quant8_pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_amdgcn_perm(b, a, 0x0c0c0400u) | __builtin_amdgcn_perm(d, c, 0x04000c0cu);
}
// Byte address of byte `kb` of row `row` in a packed operand ([KB/64][rows16/16] blocks of 16 rows x 64 bytes = 1 KiB):
//   MIXQ_FMT_P16X64: row r of a block stores its four 16-byte chunks c at r*64 + (c ^ (-(r>>2) & 3))*16: a row's 64 bytes stay
//                    contiguous (the quantise kernels write them as such), and BOTH fragment shapes read the LDS image without bank
//                    conflicts: 32x32x32 (row l&31, chunk 2s + (l>>5): gemm.hip) and 16x16x64 (row l&15, chunk l>>4: gemm_wreg.hip);
//                    the round-1 swizzle (r>>2)&3 served only the first (tests/test_pack_properties.py checks both)
//   MIXQ_FMT_F16X64: chunk-major "fragment order" c*256 + r*16 - lane l of a wave owns bytes [16 l, 16 l + 16) of a block,
//                    which is row l&15, k-chunk l>>4: exactly one v_mfma_i32_16x16x64_i8 operand (gemm_wreg.hip)
__device__ __forceinline__ size_t packed_offset(int fmt, int row, int kb, int rows16) {
    const int r = row & 15, c = (kb & 63) >> 4;
    const size_t blk = (static_cast<size_t>(kb >> 6) * (rows16 >> 4) + (row >> 4)) * 1024;
    return blk + (fmt == MIXQ_FMT_F16X64 ? c * 256 + r * 16 : r * 64 + ((c ^ ((0 - (r >> 2)) & 3)) << 4)) + (kb & 15);
}
// ---- MIXQ_FMT_F6X128 (include/mixq_hip.h): int4 values as FP6 E3M2 codes ------------------------------------------------------
// code of a two's-complement nibble (-8 .. 7): sign bit 0x20, then the magnitude's code (3-bit exponent of bias 3, 2-bit mantissa):
// 0 1 2 3 4 5 6 7 8 = 0x00 0x0c 0x10 0x12 0x14 0x15 0x16 0x17 0x18
__device__ __forceinline__ uint32_t f6_code_of_int(int q) {                 // q in [-8, 8]
    const uint32_t mag = static_cast<uint32_t>(q < 0 ? -q : q);
    const uint32_t c = mag >= 4u ? mag + 16u : (0x12100C00u >> (mag * 8u)) & 0xffu;     // 4 .. 8: exponent 5 / 6, mantissa = the low bits
    return c | (q < 0 ? 0x20u : 0u);
}
__device__ __forceinline__ uint32_t f6_code_of_nibble(uint32_t nib) {
    return f6_code_of_int(static_cast<int>(nib << 28) >> 28);
}
__device__ __forceinline__ uint32_t f6_nibble_of_code(uint32_t code) {      // inverse (codes that are no integer of [-8, 8] map to 0)
    const uint32_t c = code & 0x1fu;
    const uint32_t mag = static_cast<uint32_t>(((c & 16u) ? 0x876540302ull : 0x1000000000000ull) >> ((c & 15u) * 4)) & 0xfu;
    return (code & 0x20u) ? (16u - mag) & 0xfu : mag & 0xfu;
}
// byte address of the block of (row, element k) and the lane that owns the element's 32-element group
__device__ __forceinline__ size_t f6_block_offset(int row, int k, int rows16) {
    return (static_cast<size_t>(k >> 7) * (rows16 >> 4) + (row >> 4)) * 1536;
}
__device__ __forceinline__ int f6_group(int k) { return (k & 127) >> 5; }
// where the two pieces of lane fragment (row, g = 32-element group of the block) live inside its block, by format:
//   F6X128 (fragment order): 16 bytes at 16 lane, 8 bytes at 1024 + 8 lane, lane = 16 g + row % 16
//   R6X128 (row-major):      a row's 96 bytes of the block are ONE run at 96 r: its four 16-byte pieces (16 g), then its four 8-byte pieces
//                            (64 + 8 g), r = row % 16 - what a quantise kernel, which owns one row, writes as whole cache-line segments
__device__ __forceinline__ void f6_pieces(int fmt, uint8_t* blk, int row, int g, uint8_t*& A, uint8_t*& B) {
    const int r = row & 15;
    if (fmt == MIXQ_FMT_R6X128) { A = blk + r * 96 + g * 16; B = blk + r * 96 + 64 + g * 8; }
    else                        { A = blk + (g * 16 + r) * 16; B = blk + 1024 + (g * 16 + r) * 8; }
}
// eight consecutive codes (elements 8 c8 .. 8 c8 + 7 of a fragment's 32, c8 = 0..3) = 48 bits at byte 6 c8 of the 24-byte fragment
__device__ __forceinline__ void f6_store8(int fmt, uint8_t* blk, int row, int g, int c8, const uint32_t (&code)[8]) {
    const uint32_t lo = code[0] | (code[1] << 6) | (code[2] << 12) | (code[3] << 18) | (code[4] << 24) | (code[5] << 30);
    const uint32_t hi = (code[5] >> 2) | (code[6] << 4) | (code[7] << 10);                   // 16 bits
    uint8_t *A, *B;
    f6_pieces(fmt, blk, row, g, A, B);
    if (c8 == 0)      { *reinterpret_cast<uint32_t*>(A) = lo;      *reinterpret_cast<uint16_t*>(A + 4) = static_cast<uint16_t>(hi); }
    else if (c8 == 1) { *reinterpret_cast<uint16_t*>(A + 6) = static_cast<uint16_t>(lo); *reinterpret_cast<uint32_t*>(A + 8) = (lo >> 16) | (hi << 16); }
    else if (c8 == 2) { *reinterpret_cast<uint32_t*>(A + 12) = lo; *reinterpret_cast<uint16_t*>(B) = static_cast<uint16_t>(hi); }
    else              { *reinterpret_cast<uint16_t*>(B + 2) = static_cast<uint16_t>(lo); *reinterpret_cast<uint32_t*>(B + 4) = (lo >> 16) | (hi << 16); }
}
// sixteen consecutive codes (elements 16 h .. 16 h + 15 of a lane's 32, h = 0 / 1) = 96 bits (w0 w1 w2) at byte 12 h of the fragment: one
// 12-byte store, or 4 + 8 bytes across the two pieces - what the quantisers write (two adjacent 8-element chunks at a time)
__device__ __forceinline__ void f6_store_words(int fmt, uint8_t* blk, int row, int g, int h, uint32_t w0, uint32_t w1, uint32_t w2) {
    uint8_t *A, *B;
    f6_pieces(fmt, blk, row, g, A, B);
    if (h == 0) {
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
        *reinterpret_cast<u32x3*>(A) = u32x3{w0, w1, w2};
    } else {
        *reinterpret_cast<uint32_t*>(A + 12) = w0;
        *reinterpret_cast<uint2*>(B) = make_uint2(w1, w2);
    }
}
// eight codes as 48 bits (lo: 32, hi: 16)
__device__ __forceinline__ void f6_pack8(const uint32_t* c, uint32_t& lo, uint32_t& hi) {
    lo = c[0] | (c[1] << 6) | (c[2] << 12) | (c[3] << 18) | (c[4] << 24) | (c[5] << 30);
    hi = (c[5] >> 2) | (c[6] << 4) | (c[7] << 10);
}
__device__ __forceinline__ void f6_store16(int fmt, uint8_t* blk, int row, int g, int h, const uint32_t (&c)[16]) {
    uint32_t lo0, hi0, lo1, hi1;
    f6_pack8(c, lo0, hi0);
    f6_pack8(c + 8, lo1, hi1);
    f6_store_words(fmt, blk, row, g, h, lo0, hi0 | (lo1 << 16), (lo1 >> 16) | (hi1 << 16));
}
// The KEPT OUTLIER MAP of a frozen layer (include/mixq_hip.h, mixq_quant_fused_masked): [W = ceil(K / 32) words: bit c set <=> column c is
// an outlier column][1 word: the number of columns marked][pad to a multiple of 4 words][K uint16 AND-masks: keep[c] = 0x0000 for an
// outlier column, 0xffff for every other column].  The AND-masks are what lets a quantise pass take its outlier values OUT OF THE ROW IT
// HOLDS IN REGISTERS at almost no instruction cost (the passes are issue-bound: profiles/r05_quant_pmc.txt): a thread reads the 16 bytes of
// masks of each of its 16-byte chunks beside the chunk (the table is a few KB and cache-resident, the same for every row); a chunk that
// holds a marked column - a handful per wave - goes to an LDS image of the row AS A WHOLE (one ds_write_b128) and is zeroed by four
// v_and; behind the barrier of the row maximum lane j reads element ind[j] of that image for x_out[row][j].  No `ind[j]` ->
// `x[row][ind[j]]` chain of two dependent memory round trips behind the row load, no scattered 2-byte stores, no per-element branches.
__device__ __forceinline__ const uint4* kept_mask_table(const uint32_t* map, int K) {
    return reinterpret_cast<const uint4*>(map + ((((K + 31) >> 5) + 1 + 3) & ~3));
}
// One chunk against its eight AND-masks: a chunk with a marked element goes to the LDS row image (chunk c at halves [8 c, 8 c + 8)) and
// continues with those elements zeroed.  Returns true when the chunk held any.
__device__ __forceinline__ bool kept_apply8(uint4& v, const uint4& mk, uint16_t* rowimg, int c) {
    if ((mk.x & mk.y & mk.z & mk.w) == 0xffffffffu) return false;
    reinterpret_cast<uint4*>(rowimg)[c] = v;
    v = make_uint4(v.x & mk.x, v.y & mk.y, v.z & mk.z, v.w & mk.w);
    return true;
}

__device__ __forceinline__ int wave_id_uniform() {
    return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
}
// SiLU of every fused epilogue.  The product carries no "contract" flag: after inlining, `silu(z) + bias` must stay a multiplication and
// an addition in EVERY kernel form - left free, the compiler fuses them into one FMA in some instantiations and not in others, and two
// routes that compute the same expression differ in the last fp16 bit of one element in 10^5 (found by the joint gate / up launch against
// the two launches it replaces).  The oracle rounds the product, then adds (oracle/mixq_oracle.c).
__device__ __forceinline__ float mixq_silu(float v) {
#pragma clang fp contract(off)
    const float s = v * __builtin_amdgcn_rcpf(1.f + __expf(-v));              // 1-ulp rcp: below fp16 resolution
    return s;
}
