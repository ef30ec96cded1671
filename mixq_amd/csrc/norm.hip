// SURVEY.md §8f row 1 — the step in front of the path: FasterTransformer-style RMSNorm, optionally fused with the
// (i)+(ii) work of the NEXT MixQ linear (extract + zero its known outlier columns, per-token scale, quantise), so the
// activation is read from HBM once and the linear starts from `cache.q_xcache / x_scale / activation_outliers`.
// Replaces mixlib.layernorm_forward_cuda(x, w, out, eps) and mixlib.layernorm_forward_cuda_extract_outliers[_int4](x, w,
// out, eps, ind, x_scale) -> (X_out, q_x)   (call sites /root/reference/mixquant/modules/fused/norm.py:21-33).
//
// Arithmetic contract (fixed by decision - the reference kernel is not in /root/reference - and restated bit-exactly
// by oracle/mixq_oracle.c:orc_rmsnorm*):
//   ss      = sum_k x[k]^2 in fp32 with fmaf, in THIS order: thread t of 256 accumulates its 16-byte chunks t, t+256, ...
//             element by element; 64-lane butterfly (xor 32,16,8,4,2,1); then (w0 + w1) + (w2 + w3) over the 4 waves
//   inv_rms = 1 / sqrt(ss / K + eps)                (IEEE fp32 divide and sqrt)
//   y[k]    = fp16( (float(x[k]) * inv_rms) * float(w[k]) )
//   fused form: X_out[:, j] = y[:, ind[j]]; y[:, ind] = 0 in `out`; x_scale / q from the zeroed y exactly as
//   mixq_find_row_scale does.
#include "common.h"

namespace {

constexpr int NT = 256;
typedef unsigned short us2n_t __attribute__((ext_vector_type(2)));

// y = fp16( fp32(fp32(x * inv) * w) ): every product is rounded to fp32 before the next step.  Written with the _rn
// intrinsics and an opaque barrier so the compiler cannot fold the last multiply and the conversion into one
// v_fma_mixlo_f16 (a single rounding of the exact product, which differs from the contract in ~1 of 30k elements).
__device__ __forceinline__ uint16_t norm1(uint16_t x, float inv, uint16_t w) {
    float t = __fmul_rn(__fmul_rn(h2f(x), inv), h2f(w));
    asm volatile("" : "+v"(t));
    return f2h(t);
}

__device__ __forceinline__ float block_sum_ordered(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int BIT, int NCH, bool QUANT>
__global__ __launch_bounds__(NT) void rmsnorm_kernel(            // (parameter order: the 14 dwords in front of the first requests first - quant.hip)
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint32_t* __restrict__ col_mask, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ ind, int ldx, int K, int n_cap, float eps,
    uint16_t* __restrict__ out, int ldout, uint16_t* __restrict__ x_scale,
    void* __restrict__ q, uint16_t* __restrict__ x_out, int ldxo, int32_t* __restrict__ flag, float thr_scale, int rows16, int fmt)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];     // column bitmask, 8 floats, (16-byte aligned) [K] fp16: the row's LDS image (kept route)
    const int row = blockIdx.x, tid = threadIdx.x;
    const int mask_words = (K + 31) >> 5;
    float* red = reinterpret_cast<float*>(smem + mask_words);
    const uint16_t* xr = x + static_cast<size_t>(row) * ldx;
    const int nchunk = K >> 3;

    uint4 keep[NCH], wk[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT;
        if (c < nchunk) {
            keep[i] = reinterpret_cast<const uint4*>(xr)[c];
            wk[i] = reinterpret_cast<const uint4*>(w)[c];
        }
    }
    // col_mask (mixq_rmsnorm_quant_fused_masked): the next layer's KEPT OUTLIER MAP (bits, count, per-column AND-masks: common.h).  A
    // chunk's eight masks are 16 bytes of it, requested here with the row, the live count and this lane's column ind[tid]: the normalised
    // outlier values are then taken out of the registers that hold the row (a marked chunk goes to the row's LDS image as a whole) -
    // unmasked they cost three dependent round trips (count, ind[j], x[ind[j]]) and two barriers around the LDS mask in front of the row maximum.
    const uint4* pv = (QUANT && col_mask) ? kept_mask_table(col_mask, K) : nullptr;
    uint4 pg[NCH];
    int nd0 = n_cap, mcount = 0, gi = 0;
    if (pv) {
        if (tid < n_cap) gi = ind[tid];
        mcount = static_cast<int>(col_mask[mask_words]);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NT;
            pg[i] = c < nchunk ? pv[c] : make_uint4(~0u, ~0u, ~0u, ~0u);
        }
        if (n_dev) nd0 = *n_dev;
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT;
        if (c < nchunk) {
            const uint32_t d[4] = {keep[i].x, keep[i].y, keep[i].z, keep[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = h2f(static_cast<uint16_t>(d[e] & 0xffffu)), hi = h2f(static_cast<uint16_t>(d[e] >> 16));
                ss = __fmaf_rn(lo, lo, ss);
                ss = __fmaf_rn(hi, hi, ss);
            }
        }
    }
    const float total = block_sum_ordered(ss, red);
    // sqrtf, not __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS the latter is the 1-ulp native v_sqrt_f32, while
    // sqrtf and '/' are IEEE-correct under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt
    const float inv = 1.0f / sqrtf(total / static_cast<float>(K) + eps);

    int n = 0;
    bool have_out = false, kept = false;
    uint16_t* rowimg = reinterpret_cast<uint16_t*>(smem + ((mask_words + 8 + 3) & ~3));   // kept route: marked chunks of the normalised row
    if constexpr (QUANT) {
        n = n_cap;
        if (pv) n = nd0 < n_cap ? nd0 : n_cap;
        else if (n_dev) { const int nd = *n_dev; n = nd < n_cap ? nd : n_cap; }
        have_out = (n > 0) && ind != nullptr;
        // (a kept map built for another count than the live one - lowered in device memory behind the host's back - is not used: quant.hip)
        kept = pv != nullptr && have_out && mcount == n;
        if (have_out && !kept) {
            for (int i = tid; i < mask_words; i += NT) smem[i] = 0u;
            __syncthreads();
            for (int j = tid; j < n; j += NT) {
                const int c = ind[j];
                const uint16_t yv = norm1(xr[c], inv, w[c]);
                if (x_out) x_out[static_cast<size_t>(row) * ldxo + j] = yv;
                atomicOr(&smem[c >> 5], 1u << (c & 31));
            }
        }
        if (x_out && !kept) for (int j = (have_out ? n : 0) + tid; j < ldxo; j += NT) x_out[static_cast<size_t>(row) * ldxo + j] = 0;
        if (have_out && !kept) __syncthreads();
    }

    uint32_t amax_acc = 0u;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT;
        if (c < nchunk) {
            const uint32_t d[4] = {keep[i].x, keep[i].y, keep[i].z, keep[i].w};
            const uint32_t g[4] = {wk[i].x, wk[i].y, wk[i].z, wk[i].w};
            const uint32_t m8 = (!kept && have_out) ? ((smem[c >> 2] >> ((c & 3) * 8)) & 0xffu) : 0u;
            uint32_t y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t lo = norm1(static_cast<uint16_t>(d[e] & 0xffffu), inv, static_cast<uint16_t>(g[e] & 0xffffu));
                uint32_t hi = norm1(static_cast<uint16_t>(d[e] >> 16), inv, static_cast<uint16_t>(g[e] >> 16));
                if (m8 & (1u << (2 * e)))     lo = 0;
                if (m8 & (1u << (2 * e + 1))) hi = 0;
                y[e] = lo | (hi << 16);
            }
            keep[i] = make_uint4(y[0], y[1], y[2], y[3]);
            if (kept) (void)kept_apply8(keep[i], pg[i], rowimg, c);        // a chunk with normalised outlier values -> the row's LDS image; zeroed in the chunk
            // running maximum of |y| as packed fp16 bit patterns (finite halves: |a| < |b| <=> (a & 0x7fff) < (b & 0x7fff) as unsigned): one
            // v_and + one v_pk_max_u16 per pair instead of two conversions and two maxima (the pass is issue- and latency-bound)
            const uint32_t z[4] = {keep[i].x, keep[i].y, keep[i].z, keep[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                amax_acc = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2n_t, amax_acc), __builtin_bit_cast(us2n_t, z[e] & 0x7fff7fffu)));
            reinterpret_cast<uint4*>(out + static_cast<size_t>(row) * ldout)[c] = keep[i];
        }
    }
    if constexpr (!QUANT) return;

    float amax = h2f(static_cast<uint16_t>((amax_acc & 0xffffu) > (amax_acc >> 16) ? (amax_acc & 0xffffu) : (amax_acc >> 16)));
    amax = block_max(amax, red + 4);                                      // (its barrier also orders the row's LDS image)
    if (kept && x_out) for (int j = tid; j < ldxo; j += NT) x_out[static_cast<size_t>(row) * ldxo + j] = j < n ? rowimg[j == tid ? gi : ind[j]] : static_cast<uint16_t>(0);
    constexpr float QMAX = static_cast<float>((1 << (BIT - 1)) - 1);
    const uint16_t sh = mixq_row_scale(amax, QMAX);
    const float s = h2f(sh);
    const float rs = mixq_rcp_scale(s);
    if (tid == 0) {
        x_scale[row] = sh;
        if (flag && s > thr_scale) atomicOr(flag, 1);
    }
    char* qb = static_cast<char*>(q);
    if constexpr (BIT == 4) {
        if (fmt == MIXQ_FMT_F6X128 || fmt == MIXQ_FMT_R6X128) {
            // FP6 codes leave in PAIRS of adjacent chunks (12 bytes: whole dwords).  The chunk -> thread map stays as it is - the order
            // of the sum of squares depends on it - so the odd chunk's 48 bits travel to its even neighbour's lane.
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + i * NT;
                const bool in = c < nchunk;                          // (K % 128 == 0: both chunks of a pair are in, or neither)
                uint32_t lo = 0, hi = 0;
                if (in) {
                    uint32_t ub[8], code[8];
                    quant8_exact<4>(keep[i], s, rs, ub);
#pragma unroll
                    for (int e = 0; e < 8; ++e) code[e] = f6_code_of_int(quant8_int(ub[e]));
                    f6_pack8(code, lo, hi);
                }
                const uint32_t plo = __shfl_xor(lo, 1), phi = __shfl_xor(hi, 1);
                if (in && !(c & 1)) {
                    const int k = c * 8;
                    f6_store_words(fmt, reinterpret_cast<uint8_t*>(qb) + f6_block_offset(row, k, rows16), row, f6_group(k), (k & 31) >> 4,
                                   lo, hi | (plo << 16), (plo >> 16) | (phi << 16));
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT;
        if (c < nchunk) {
            uint32_t ub[8];
            quant8_exact<BIT>(keep[i], s, rs, ub);
            if constexpr (BIT == 8) {
                uint2 o;
                o.x = quant8_pack4(ub[0], ub[1], ub[2], ub[3]);
                o.y = quant8_pack4(ub[4], ub[5], ub[6], ub[7]);
                const size_t off = fmt ? packed_offset(fmt, row, c * 8, rows16) : static_cast<size_t>(row) * K + c * 8;
                *reinterpret_cast<uint2*>(qb + off) = o;
            } else {
                uint32_t o = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) o |= static_cast<uint32_t>((ub[2 * e] & 0xfu) | ((ub[2 * e + 1] & 0xfu) << 4)) << (8 * e);
                const size_t off = fmt ? packed_offset(fmt, row, c * 4, rows16) : static_cast<size_t>(row) * (K >> 1) + c * 4;
                *reinterpret_cast<uint32_t*>(qb + off) = o;
            }
        }
    }
}

template <int BIT, bool QUANT>
int launch_norm(const uint16_t* x, int ldx, const uint16_t* w, float eps, uint16_t* out, int ldout, const int32_t* ind, int n,
                const int32_t* n_dev, uint16_t* x_scale, void* q, uint16_t* x_out, int ldxo, int32_t* flag, int M, int K,
                float thr, int qfmt, hipStream_t st, const uint32_t* col_mask = nullptr)
{
    const size_t shm = ((static_cast<size_t>((K + 31) >> 5) + 8 + 3) & ~static_cast<size_t>(3)) * sizeof(uint32_t) + (col_mask ? static_cast<size_t>(K) * 2 : 0);
    const int nchunk = K >> 3;
    const int rows16 = qfmt ? ((M + 15) & ~15) : 0;
    dim3 g(M), b(NT);
#define MIXQ_NLAUNCH(NCH) hipLaunchKernelGGL((rmsnorm_kernel<BIT, NCH, QUANT>), g, b, shm, st, x, w, col_mask, n_dev, ind, ldx, K, n, eps, out, ldout, x_scale, q, x_out, ldxo, flag, thr, rows16, qfmt)
    if      (nchunk <= 2 * NT)  MIXQ_NLAUNCH(2);
    else if (nchunk <= 4 * NT)  MIXQ_NLAUNCH(4);
    else if (nchunk <= 8 * NT)  MIXQ_NLAUNCH(8);
    else if (nchunk <= 16 * NT) MIXQ_NLAUNCH(16);
    else return MIXQ_ESHAPE;
#undef MIXQ_NLAUNCH
    return mixq_launch_status();
}

inline float fp16_round(float v) { return static_cast<float>(static_cast<_Float16>(v)); }

}  // namespace

extern "C" int mixq_rmsnorm(const uint16_t* x, const uint16_t* weight, uint16_t* out, int M, int K, int ldx, int ldout, float eps,
                            mixq_stream_t stream)
{
    if (M < 0 || K <= 0 || (M > 0 && (!x || !weight || !out))) return MIXQ_EINVAL;
    if ((K & 7) || (ldx & 7) || (ldout & 7) || ldx < K || ldout < K) return MIXQ_ESHAPE;
    if (M == 0) return MIXQ_OK;
    return launch_norm<8, false>(x, ldx, weight, eps, out, ldout, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, M, K,
                                 0.f, 0, mixq_stream(stream));
}

static int rmsnorm_quant_common(const uint16_t* x, const uint16_t* weight, uint16_t* out, const int32_t* ind, int n,
                                const int32_t* n_dev, const uint32_t* col_mask, uint16_t* x_scale, void* q, uint16_t* x_out, int32_t* flag, int M,
                                int K, int ldx, int ldout, int ldxo, float eps, int bit, float sigma, int qfmt,
                                mixq_stream_t stream)
{
    if (M < 0 || K <= 0 || n < 0 || (M > 0 && (!x || !weight || !out || !x_scale || !q))) return MIXQ_EINVAL;
    if (bit != 8 && bit != 4) return MIXQ_EINVAL;
    if (qfmt != MIXQ_FMT_PLAIN && qfmt != MIXQ_FMT_P16X64 && qfmt != MIXQ_FMT_F16X64 && !((qfmt == MIXQ_FMT_F6X128 || qfmt == MIXQ_FMT_R6X128) && bit == 4)) return MIXQ_EINVAL;
    if (n > 0 && (!ind || !x_out || ldxo < n)) return MIXQ_EINVAL;
    if ((K & 7) || (ldx & 7) || (ldout & 7) || ldx < K || ldout < K || (bit == 4 && (K & 15))) return MIXQ_ESHAPE;
    if (qfmt != MIXQ_FMT_PLAIN && (bit == 8 ? K : K / 2) % 64) return MIXQ_ESHAPE;
    if (M == 0) return MIXQ_OK;
    const float qmax = static_cast<float>((1 << (bit - 1)) - 1);
    const float thr = fp16_round(fp16_round(sigma) / qmax);
    uint16_t* xo = (n > 0) ? x_out : nullptr;
    if (K > 30000) col_mask = nullptr;                  // (as mixq_quant_fused_masked: the kept route keeps an fp16 image of the row in LDS, inside the default 64 KB)
    if (bit == 8)
        return launch_norm<8, true>(x, ldx, weight, eps, out, ldout, ind, n, n_dev, x_scale, q, xo, ldxo, flag, M, K, thr, qfmt,
                                    mixq_stream(stream), col_mask);
    return launch_norm<4, true>(x, ldx, weight, eps, out, ldout, ind, n, n_dev, x_scale, q, xo, ldxo, flag, M, K, thr, qfmt,
                                mixq_stream(stream), col_mask);
}

extern "C" int mixq_rmsnorm_quant_fused(const uint16_t* x, const uint16_t* weight, uint16_t* out, const int32_t* ind, int n,
                                        const int32_t* n_dev, uint16_t* x_scale, void* q, uint16_t* x_out, int32_t* flag, int M,
                                        int K, int ldx, int ldout, int ldxo, float eps, int bit, float sigma, int qfmt,
                                        mixq_stream_t stream)
{
    return rmsnorm_quant_common(x, weight, out, ind, n, n_dev, nullptr, x_scale, q, x_out, flag, M, K, ldx, ldout, ldxo, eps, bit, sigma, qfmt, stream);
}
// ... for a next layer whose prediction is frozen: its kept outlier map (as mixq_quant_fused_masked)
extern "C" int mixq_rmsnorm_quant_fused_masked(const uint16_t* x, const uint16_t* weight, uint16_t* out, const int32_t* ind, int n,
                                               const int32_t* n_dev, const uint32_t* col_mask, int map_words, uint16_t* x_scale, void* q, uint16_t* x_out,
                                               int32_t* flag, int M, int K, int ldx, int ldout, int ldxo, float eps, int bit, float sigma,
                                               int qfmt, mixq_stream_t stream)
{
    if (n > 0 && (!col_mask || map_words < mixq_kept_map_words(K) || (reinterpret_cast<uintptr_t>(col_mask) & 15))) return MIXQ_EINVAL;   // (see mixq_quant_fused_masked)
    return rmsnorm_quant_common(x, weight, out, ind, n, n_dev, n > 0 ? col_mask : nullptr, x_scale, q, x_out, flag, M, K, ldx, ldout, ldxo, eps, bit,
                                sigma, qfmt, stream);
}
