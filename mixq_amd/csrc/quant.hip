// Memory-bound activation kernels of the MixQ path for gfx950:
//   (i)  per-token absmax -> fp16 scale, misprediction flag, outlier-column detection + compaction
//   (ii) fused outlier extraction (dense fp16 side matrix, in-place zeroing) + symmetric int8 / int4 quantise
// plus the cold weight-column dequantisation used when new outlier columns are appended.
//
// These are HBM-bound byte kernels: one 256-thread workgroup per activation row, 16-byte coalesced loads,
// the row kept in registers between the absmax pass and the quantise pass (one trip over HBM), wave-shuffle +
// LDS reductions.  They follow the call sites /root/reference/mixquant/modules/linear.py:157-161,187-226 and
// the conventions fixed in include/mixq_hip.h.
#include "common.h"

namespace {

constexpr int QT = 256;                 // threads per row workgroup

__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    return v;
}

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
// Running max of |x| as packed fp16 bit patterns (for finite halves |a| < |b| <=> (a & 0x7fff) < (b & 0x7fff) as unsigned):
// 8 halves with the outlier columns (bits of m8) forced to zero cost one v_and + one v_pk_max_u16 per pair.  The masked
// vector is written back for the quantise pass; amax_finish turns the accumulator into the float maximum.
__device__ __forceinline__ uint32_t amax8_masked(uint4& v, uint32_t m8, uint32_t acc) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (m8) {                                              // rare: a chunk that holds outlier columns
            if (m8 & (1u << (2 * i)))     w[i] &= 0xffff0000u;
            if (m8 & (1u << (2 * i + 1))) w[i] &= 0x0000ffffu;
        }
        acc = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2_t, acc), __builtin_bit_cast(us2_t, w[i] & 0x7fff7fffu)));
    }
    v = make_uint4(w[0], w[1], w[2], w[3]);
    return acc;
}
__device__ __forceinline__ float amax_finish(uint32_t acc) {
    const uint32_t lo = acc & 0xffffu, hi = acc >> 16;
    return h2f(static_cast<uint16_t>(lo > hi ? lo : hi));
}

// FP6 codes of a chunk's 8 quantised values (MIXQ_FMT_F6X128)
__device__ __forceinline__ void quant_codes8(const uint4& v, float s, float rs, uint32_t* code) {
    uint32_t ub[8];
    quant8_exact<4>(v, s, rs, ub);
#pragma unroll
    for (int i = 0; i < 8; ++i) code[i] = f6_code_of_int(quant8_int(ub[i]));
}

// q: plain -> row base pointer, packed (fmt != 0) -> matrix base pointer
template <int BIT>
__device__ __forceinline__ void quant_store8(const uint4& v, float s, float rs, void* q, int chunk, int row, int rows16, int fmt) {
    uint32_t ub[8];
    quant8_exact<BIT>(v, s, rs, ub);
    if constexpr (BIT == 8) {
        uint2 o;
        o.x = quant8_pack4(ub[0], ub[1], ub[2], ub[3]);
        o.y = quant8_pack4(ub[4], ub[5], ub[6], ub[7]);
        if (fmt) *reinterpret_cast<uint2*>(static_cast<char*>(q) + packed_offset(fmt, row, chunk * 8, rows16)) = o;
        else     reinterpret_cast<uint2*>(q)[chunk] = o;
    } else if (fmt == MIXQ_FMT_F6X128 || fmt == MIXQ_FMT_R6X128) {   // FP6 codes (include/mixq_hip.h): chunk = elements 8 chunk .. + 7 of the row
        uint32_t code[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) code[i] = f6_code_of_int(quant8_int(ub[i]));
        const int k = chunk * 8;
        f6_store8(fmt, static_cast<uint8_t*>(q) + f6_block_offset(row, k, rows16), row, f6_group(k), (k & 31) >> 3, code);
    } else {   // nibble pack: low nibble = even column (linear.py:14-18)
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o |= static_cast<uint32_t>((ub[2 * i] & 0xfu) | ((ub[2 * i + 1] & 0xfu) << 4)) << (8 * i);
        if (fmt) *reinterpret_cast<uint32_t*>(static_cast<char*>(q) + packed_offset(fmt, row, chunk * 4, rows16)) = o;
        else     reinterpret_cast<uint32_t*>(q)[chunk] = o;
    }
}

// One workgroup per row.  NCH = number of 16-byte chunks each thread keeps in registers (row <= NCH*256*8
// elements stays on chip between the two passes); NCH == 0 re-reads the row (L2-resident) for any K.
template <int BIT, int NCH>
__global__ __launch_bounds__(QT) void quant_rows_kernel(
    uint16_t* __restrict__ x, int ldx, const int32_t* __restrict__ ind, int n_cap, const int32_t* __restrict__ n_dev,
    uint16_t* __restrict__ x_scale, void* __restrict__ q, uint16_t* __restrict__ x_out, int ldo,
    int32_t* __restrict__ flag, int K, float thr_scale, int rows16, int fmt)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // [K/32 (+1)] column bitmask, 4 floats, row copy
    const int row = blockIdx.x, tid = threadIdx.x;
    const int mask_words = (K + 31) >> 5;
    float* red = reinterpret_cast<float*>(smem + mask_words);
    uint16_t* xr = x + static_cast<size_t>(row) * ldx;

    // The row is requested FIRST (16 bytes per lane, all chunks in flight) so the outlier bookkeeping below - index load,
    // gather, in-place zeroing, LDS bitmask - overlaps the one HBM round trip instead of adding a second one.
    const int nchunk = K >> 3;
    const uint4* xv = reinterpret_cast<const uint4*>(xr);
    uint4 keep[NCH > 0 ? NCH : 1];
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * QT;
            if (c < nchunk) keep[i] = xv[c];
        }
    }

    // The first outlier index of this thread is requested speculatively (entries >= n are ignored) together with the
    // device-side count, so row, count and indices share ONE memory round trip; the gather then reads the row from an LDS
    // copy instead of going back to global memory for x[row][c].
    int c0 = 0;
    if (ind != nullptr && tid < n_cap) c0 = ind[tid];
    int n = n_cap;
    if (n_dev) { int nd = *n_dev; n = nd < n_cap ? nd : n_cap; }
    const bool have_out = (n > 0) && ind != nullptr;
    if (have_out) {
        for (int i = tid; i < mask_words; i += QT) smem[i] = 0u;
        if constexpr (NCH > 0 && NCH <= 8) {      // up to 32 KB of LDS for the row copy
            uint4* stage = reinterpret_cast<uint4*>(smem + ((mask_words + 4 + 3) & ~3));
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + i * QT;
                if (c < nchunk) stage[c] = keep[i];
            }
        }
        __syncthreads();
        const uint16_t* srow = reinterpret_cast<const uint16_t*>(smem + ((mask_words + 4 + 3) & ~3));
        for (int j = tid; j < n; j += QT) {
            const int c = (j == tid) ? c0 : ind[j];
            const uint16_t v = (NCH > 0 && NCH <= 8) ? srow[c] : xr[c];
            if (x_out) x_out[static_cast<size_t>(row) * ldo + j] = v;
            xr[c] = 0;                                             // reference zeroes the caller's tensor in place
            atomicOr(&smem[c >> 5], 1u << (c & 31));
        }
    }
    if (x_out) for (int j = (have_out ? n : 0) + tid; j < ldo; j += QT) x_out[static_cast<size_t>(row) * ldo + j] = 0;
    if (have_out) __syncthreads();

    uint32_t amax_acc = 0u;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * QT;
            if (c < nchunk) {
                uint32_t m8 = have_out ? ((smem[c >> 2] >> ((c & 3) * 8)) & 0xffu) : 0u;
                amax_acc = amax8_masked(keep[i], m8, amax_acc);    // whatever the load saw in a zeroed column is masked
            }
        }
    } else {
        for (int c = tid; c < nchunk; c += QT) {
            uint4 v = xv[c];
            uint32_t m8 = have_out ? ((smem[c >> 2] >> ((c & 3) * 8)) & 0xffu) : 0u;
            amax_acc = amax8_masked(v, m8, amax_acc);
        }
    }
    const float amax = block_max_256(amax_finish(amax_acc), red);
    constexpr float QMAX = static_cast<float>((1 << (BIT - 1)) - 1);
    const uint16_t sh = mixq_row_scale(amax, QMAX);
    const float s = h2f(sh);
    const float rs = mixq_rcp_scale(s);      // one division per row; quant_exact needs none per element
    if (tid == 0) {
        x_scale[row] = sh;
        if (flag && s > thr_scale) atomicOr(flag, 1);
    }
    void* qrow = fmt ? q : static_cast<void*>(static_cast<char*>(q) + static_cast<size_t>(row) * (BIT == 8 ? K : (K >> 1)));
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * QT;
            if (c < nchunk) quant_store8<BIT>(keep[i], s, rs, qrow, c, row, rows16, fmt);
        }
    } else {
        for (int c = tid; c < nchunk; c += QT) {
            uint4 v = xv[c];
            uint32_t m8 = have_out ? ((smem[c >> 2] >> ((c & 3) * 8)) & 0xffu) : 0u;
            (void)amax8_masked(v, m8, 0u);
            quant_store8<BIT>(v, s, rs, qrow, c, row, rows16, fmt);
        }
    }
}

// Second form of the same pass (round 2): TPR threads per row, RPB rows per workgroup.  What changed against the kernel
// above, which needed 5.9 us for 6.3 MB (13 % of the HBM rate): the outlier values are gathered straight from global
// memory while the row loads are in flight (no LDS copy of the row, no second pass over it), the column bitmask is built
// once per workgroup and shared by its RPB rows, and with TPR == 64 a row lives in ONE wave, so the absmax needs no LDS
// round trip and no barrier at all when the layer has no outlier columns.  Results are bit-identical to the first form.
template <int BIT, int TPR, int RPB, int NCH, bool KEPT>
// (parameter order: what a thread needs before it can request its row - 12 dwords - comes FIRST: built with -amdgpu-kernarg-preload-count
// those arrive in SGPRs with the wave, and no scalar load stands in front of the row request; the rest is waited for behind it)
// KEPT: the caller handed over the layer's kept outlier map (col_mask != nullptr, n_cap > 0) - the instantiation holds only that route (and
// its slow form for a map that does not describe the live count); !KEPT: the route that builds the column mask in LDS.
__global__ __launch_bounds__(TPR * RPB) void quant_rows2_kernel(
    uint16_t* __restrict__ x, const uint32_t* __restrict__ col_mask, const int32_t* __restrict__ n_dev, const int32_t* __restrict__ ind,
    int ldx, int M, int K, int n_cap,
    uint16_t* __restrict__ x_scale, void* __restrict__ q, uint16_t* __restrict__ x_out, int ldo,
    int32_t* __restrict__ flag, float thr_scale, int rows16, int fmt, int dbg)
{
    constexpr int NT = TPR * RPB, WPR = TPR / 64;
    // !KEPT: [K/32] column bitmask, RPB * WPR floats.  KEPT: RPB * WPR floats (padded to 16 bytes), [RPB][K] fp16: the rows' LDS images
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x, rw = RPB == 1 ? 0 : tid / TPR, t = tid - rw * TPR;
    const int row = blockIdx.x * RPB + rw;
    const bool valid = RPB == 1 || row < M;
    const int mask_words = (K + 31) >> 5;
    float* red = reinterpret_cast<float*>(KEPT ? smem : smem + mask_words);
    uint16_t* rowimg = reinterpret_cast<uint16_t*>(smem + ((RPB * WPR + 3) & ~3)) + static_cast<size_t>(rw) * K;   // (KEPT)
    uint16_t* xr = x + static_cast<size_t>(valid ? row : 0) * ldx;
    const int nchunk = K >> 3;
    const uint4* xv = reinterpret_cast<const uint4*>(xr);

    // Chunk i of this thread: t + i TPR (consecutive threads, consecutive 16 bytes) - except for the FP6 output format, where a thread
    // owns PAIRS of adjacent chunks (2 (t + (i / 2) TPR) + i % 2; NCH even): 16 codes are 12 bytes, one or two whole-dword stores,
    // while the 6 bytes of a single chunk would be a dword and a halfword store at a 6-byte stride.
    const bool pairs = BIT == 4 && NCH >= 2 && (fmt == MIXQ_FMT_F6X128 || fmt == MIXQ_FMT_R6X128);
    auto chunk = [&](int i) { return pairs ? 2 * (t + (i >> 1) * TPR) + (i & 1) : t + i * TPR; };
    uint4 keep[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = chunk(i);
        keep[i] = (valid && c < nchunk) ? xv[c] : make_uint4(0, 0, 0, 0);
    }
    // col_mask (mixq_quant_fused_masked): the caller's KEPT OUTLIER MAP (a frozen layer's never changes; layout: common.h).  A chunk's eight
    // AND-masks are 16 bytes of it, requested here beside the row, with the count it was built for and this lane's column ind[t]:
    // everything the pass needs then arrives in ONE memory round trip - the row maximum waits for neither the device-resident count, nor
    // `ind`, nor the two barriers around building a mask in LDS, and the outlier values are taken from the registers that hold the row (no
    // second, dependent round trip for x[row][ind[j]]).
    uint4 mk[KEPT ? NCH : 1];
    int mcount = 0, gi = 0;
    if constexpr (KEPT) {
        const uint4* mv = kept_mask_table(col_mask, K);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = chunk(i);
            mk[i] = c < nchunk ? mv[c] : make_uint4(~0u, ~0u, ~0u, ~0u);
#ifdef MIXQ_TUNING
            if (dbg & 64) mk[i] = make_uint4(~0u, ~0u, ~0u, ~0u);          // timing probe: masks loaded, nothing extracted
#endif
        }
        mcount = static_cast<int>(col_mask[mask_words]);
        if (t < n_cap) gi = ind[t];
    }
    int n = n_cap;
    if (n_dev) { const int nd = *n_dev; n = nd < n_cap ? nd : n_cap; }
    const bool have_out = (n > 0) && ind != nullptr && !(dbg & 4);
    // (the kept map must describe the LIVE count, which device code may have lowered behind the host's back: the word behind the bits
    // says how many columns it marks - requested with the row - and a map built for another count is not used: the slow form below)
    const bool kept = KEPT && have_out && mcount == n;
    // Without a kept map the gather of the outlier values (ind[j] -> x[row][ind[j]]) is a second, dependent memory round trip.  Only the
    // column bitmask has to exist before the row can be processed; the gathered values are REQUESTED here (up to GQ per thread, the rest
    // in the tail loop) and only consumed - stored to x_out, their column zeroed in x - after the quantised row has been written,
    // so that round trip runs beside the absmax / quantise work instead of in front of it.
    constexpr int GQ = 2;
    int gcol[GQ] = {-1, -1};
    uint16_t gval[GQ] = {0, 0};
    if constexpr (!KEPT) {
        if (have_out) {
            for (int i = tid; i < mask_words; i += NT) smem[i] = 0u;
#pragma unroll
            for (int g = 0; g < GQ; ++g) { const int j = t + g * TPR; if (j < n) gcol[g] = ind[j]; }
            __syncthreads();
            if (rw == 0) {                                     // one row's threads build the mask for the whole workgroup
#pragma unroll
                for (int g = 0; g < GQ; ++g) if (gcol[g] >= 0) atomicOr(&smem[gcol[g] >> 5], 1u << (gcol[g] & 31));
                for (int j = t + GQ * TPR; j < n; j += TPR) { const int c = ind[j]; atomicOr(&smem[c >> 5], 1u << (c & 31)); }
            }
            if (valid) {
#pragma unroll
                for (int g = 0; g < GQ; ++g) if (gcol[g] >= 0 && !(dbg & 2)) gval[g] = xr[gcol[g]];
            }
            __syncthreads();
        }
    } else if (have_out && !kept) {
        // slow form (a map for another count than the live one): the whole row through its LDS image, lane j takes and zeroes column ind[j]
#pragma unroll
        for (int i = 0; i < NCH; ++i) { const int c = chunk(i); if (c < nchunk) reinterpret_cast<uint4*>(rowimg)[c] = keep[i]; }
        __syncthreads();
        for (int j = t; j < n; j += TPR) {
            const int c = ind[j];
            const uint16_t v = rowimg[c];
            rowimg[c] = 0;
            if (valid) { if (x_out) x_out[static_cast<size_t>(row) * ldo + j] = v; xr[c] = 0; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; ++i) { const int c = chunk(i); if (c < nchunk) keep[i] = reinterpret_cast<const uint4*>(rowimg)[c]; }
    }

    uint32_t amax_acc = 0u;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = chunk(i);
        if (c < nchunk) {
            if constexpr (KEPT) {
                // a chunk with marked elements -> the LDS image of the row, zeroed in the registers; it goes back to x with its outlier columns
                // zeroed (the reference zeroes the caller's tensor in place; the other six or seven halves are rewritten unchanged)
                if (kept && kept_apply8(keep[i], mk[i], rowimg, c) && valid && !(dbg & 1)) reinterpret_cast<uint4*>(xr)[c] = keep[i];
                amax_acc = amax8_masked(keep[i], 0u, amax_acc);
            } else {
                const uint32_t m8 = have_out ? ((smem[c >> 2] >> ((c & 3) * 8)) & 0xffu) : 0u;
                amax_acc = amax8_masked(keep[i], m8, amax_acc);
            }
        }
    }
    float amax = wave_max(amax_finish(amax_acc));
    if constexpr (WPR > 1) {
        if ((tid & 63) == 0) red[tid >> 6] = amax;
        __syncthreads();
        amax = red[rw * WPR];
#pragma unroll
        for (int w = 1; w < WPR; ++w) amax = fmaxf(amax, red[rw * WPR + w]);
    }                                                      // (WPR == 1: the row lives in ONE wave, whose LDS operations are ordered - the image needs no barrier)
    constexpr float QMAX = static_cast<float>((1 << (BIT - 1)) - 1);
    const uint16_t sh = mixq_row_scale(amax, QMAX);
    const float s = h2f(sh);
    const float rs = mixq_rcp_scale(s);
    if (!valid) return;
    if (t == 0) {
        x_scale[row] = sh;
        if (flag && s > thr_scale) atomicOr(flag, 1);
    }
    if constexpr (KEPT) {
        // the x_out row, requested from the image as soon as the barrier of the maximum has ordered it: its LDS round trip and stores run
        // under the quantise arithmetic instead of behind the last quantised store
        if (x_out && !(dbg & 8)) {
            uint16_t* orow = x_out + static_cast<size_t>(row) * ldo;
            if (kept) {                                             // x_out[row][j] = element ind[j] of the row's image; zeros behind the live columns
                for (int j = t; j < ldo; j += TPR) orow[j] = j < n ? rowimg[j == t ? gi : ind[j]] : static_cast<uint16_t>(0);
            } else {                                                // (the slow form stored the live columns already)
                for (int j = (have_out ? n : 0) + t; j < ldo; j += TPR) orow[j] = 0;
            }
        }
    }
    void* qrow = fmt ? q : static_cast<void*>(static_cast<char*>(q) + static_cast<size_t>(row) * (BIT == 8 ? K : (K >> 1)));
    if constexpr (BIT == 4 && NCH >= 2) {
        if (pairs) {                                               // (K % 128 == 0: a pair is inside the row or entirely outside)
#pragma unroll
            for (int i = 0; i < NCH; i += 2) {
                const int c = chunk(i);
                if (c < nchunk) {
                    uint32_t code[16];
                    quant_codes8(keep[i], s, rs, code);
                    quant_codes8(keep[i + 1], s, rs, code + 8);
                    const int k = c * 8;
#ifdef MIXQ_TUNING
                    if (dbg & 16) {                                  // timing probe (tools build): the same bytes to a row-contiguous place
                        uint32_t lo0, hi0, lo1, hi1;
                        f6_pack8(code, lo0, hi0); f6_pack8(code + 8, lo1, hi1);
                        typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
                        *reinterpret_cast<u32x3*>(static_cast<uint8_t*>(q) + static_cast<size_t>(row) * (K * 3 / 4) + (c >> 1) * 12) = u32x3{lo0, hi0 | (lo1 << 16), (lo1 >> 16) | (hi1 << 16)};
                    } else
#endif
                    f6_store16(fmt, static_cast<uint8_t*>(q) + f6_block_offset(row, k, rows16), row, f6_group(k), (k & 31) >> 4, code);
                }
            }
        }
    }
    // ONE chunk per thread (NCH = 1) and an FP6 format: the odd chunk's 48 bits travel to its even neighbour's lane, which stores the pair
    // (12 bytes, whole dwords) - half the per-thread quantise arithmetic of the two-chunks-per-thread form above
    const bool xchg = BIT == 4 && NCH == 1 && (fmt == MIXQ_FMT_F6X128 || fmt == MIXQ_FMT_R6X128);
    if constexpr (BIT == 4 && NCH == 1) {
        if (xchg) {
            const int c = t;
            const bool in = c < nchunk;                                // (K % 128 == 0: both chunks of a pair are in, or neither)
            uint32_t lo = 0, hi = 0;
            if (in) { uint32_t code[8]; quant_codes8(keep[0], s, rs, code); f6_pack8(code, lo, hi); }
            const uint32_t plo = __shfl_xor(lo, 1), phi = __shfl_xor(hi, 1);
            if (in && !(c & 1)) {
                const int k = c * 8;
                f6_store_words(fmt, static_cast<uint8_t*>(q) + f6_block_offset(row, k, rows16), row, f6_group(k), (k & 31) >> 4,
                               lo, hi | (plo << 16), (plo >> 16) | (phi << 16));
            }
        }
    }
    if (!pairs && !xchg) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = t + i * TPR;
#ifdef MIXQ_TUNING
            if (BIT == 8 && (dbg & 32)) {                               // timing probe (tools build): no quantise arithmetic - garbage bytes to the same place
                if (c < nchunk) {
                    const uint2 o = make_uint2(keep[i].x ^ keep[i].y ^ __float_as_uint(rs), keep[i].z ^ keep[i].w);
                    if (fmt) *reinterpret_cast<uint2*>(static_cast<char*>(qrow) + packed_offset(fmt, row, c * 8, rows16)) = o;
                    else     reinterpret_cast<uint2*>(qrow)[c] = o;
                }
                continue;
            }
#endif
            if (c < nchunk) quant_store8<BIT>(keep[i], s, rs, qrow, c, row, rows16, fmt);
        }
    }
    if constexpr (KEPT) return;                                     // (x_out left in front of the quantise arithmetic)
    if (have_out) {
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            if (gcol[g] >= 0) {
                if (x_out && !(dbg & 8)) x_out[static_cast<size_t>(row) * ldo + t + g * TPR] = gval[g];
                if (!(dbg & 1)) xr[gcol[g]] = 0;                                   // reference zeroes the caller's tensor in place (same thread read it: ordered)
            }
        }
        for (int j = t + GQ * TPR; j < n; j += TPR) {              // more than GQ * TPR outlier columns: the plain dependent form
            const int c = ind[j];
            const uint16_t v = xr[c];
            if (x_out) x_out[static_cast<size_t>(row) * ldo + j] = v;
            xr[c] = 0;
        }
    }
    if (x_out && !(dbg & 8)) for (int j = (have_out ? n : 0) + t; j < ldo; j += TPR) x_out[static_cast<size_t>(row) * ldo + j] = 0;
}

// One-pass form for a row whose masked |x| maximum is already KNOWN (the GEMM that produced x left it in row_amax, as fp16 bit
// patterns maximised with atomics - mixq_gemm_i8_fused_amax): no reduction, no column-mask build (col_mask is the layer's kept outlier
// map, whose bit words the producer used), one barrier that only orders "everybody has read the maximum" before it is cleared for the next forward.
// Same bytes out as quant_rows2_kernel on the same row: the scale is fp16(amax / qmax) either way.
template <int BIT, int TPR, int NCH>
__global__ __launch_bounds__(TPR) void quant_known_kernel(       // (parameter order: see quant_rows2_kernel)
    uint16_t* __restrict__ x, const uint32_t* __restrict__ col_mask, uint32_t* __restrict__ row_amax, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ ind, int ldx, int K, int n_cap, int ldo,
    uint16_t* __restrict__ x_scale, void* __restrict__ q, uint16_t* __restrict__ x_out, int32_t* __restrict__ flag, float thr_scale, int rows16, int fmt)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t kimg[];         // [K]: the row's LDS image (kept-map route; K <= 30000)
    const int row = blockIdx.x, t = threadIdx.x;
    uint16_t* xr = x + static_cast<size_t>(row) * ldx;
    const int nchunk = K >> 3, mask_words = (K + 31) >> 5;
    const uint4* xv = reinterpret_cast<const uint4*>(xr);
    const uint4* pv = (col_mask && K <= 30000) ? kept_mask_table(col_mask, K) : nullptr;     // (col_mask is the layer's kept outlier map: bits, count, AND-masks)
    const int mcount = col_mask ? static_cast<int>(col_mask[mask_words]) : 0;
    const int gi = (ind && t < n_cap) ? ind[t] : 0;                         // lane j's column: x_out[row][j] = element ind[j] of the row's image
    uint4 keep[NCH], pg[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = t + i * TPR;
        const bool in = c < nchunk;
        keep[i] = in ? xv[c] : make_uint4(0, 0, 0, 0);
        pg[i] = (in && pv) ? pv[c] : make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    int n = n_cap;
    if (n_dev) { const int nd = *n_dev; n = nd < n_cap ? nd : n_cap; }
    const bool have_out = (n > 0) && ind != nullptr;
    const bool kept = have_out && pv != nullptr && mcount == n;           // (a map built for another live count: the dependent gather below)
    // without a usable map the outlier values of this row are requested now and consumed (x_out, in-place zero) after the quantised row has been written
    constexpr int GQ = 2;
    int gcol[GQ] = {-1, -1};
    uint16_t gval[GQ] = {0, 0};
    if (have_out && !kept) {
#pragma unroll
        for (int g = 0; g < GQ; ++g) { const int j = t + g * TPR; if (j < n) { gcol[g] = ind[j]; gval[g] = xr[gcol[g]]; } }
    }
    const uint32_t abits = row_amax[row] & 0x7fffu;
    if (kept) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = t + i * TPR;
            if (c < nchunk && kept_apply8(keep[i], pg[i], kimg, c)) reinterpret_cast<uint4*>(xr)[c] = keep[i];   // (in-place zeroing of the marked columns)
        }
    }
    __syncthreads();                                                      // every thread of the row has read the maximum (and dropped its outlier values into LDS) ...
    if (t == 0) row_amax[row] = 0u;                                       // ... before it is cleared for the producer's next run
    constexpr float QMAX = static_cast<float>((1 << (BIT - 1)) - 1);
    const uint16_t sh = mixq_row_scale(h2f(static_cast<uint16_t>(abits)), QMAX);
    const float s = h2f(sh);
    const float rs = mixq_rcp_scale(s);
    if (t == 0) {
        x_scale[row] = sh;
        if (flag && s > thr_scale) atomicOr(flag, 1);
    }
    // kept route: the x_out row from the image the barrier has just ordered - its LDS round trip and stores run under the quantise arithmetic
    if (kept && x_out) for (int j = t; j < ldo; j += TPR) x_out[static_cast<size_t>(row) * ldo + j] = j < n ? kimg[j == t ? gi : ind[j]] : static_cast<uint16_t>(0);
    void* qrow = fmt ? q : static_cast<void*>(static_cast<char*>(q) + static_cast<size_t>(row) * (BIT == 8 ? K : (K >> 1)));
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = t + i * TPR;
        if (c < nchunk) {
            // (without a usable map: zero the outlier columns of the chunk, whatever the load saw there, from the map's bit words)
            if (!kept && col_mask) (void)amax8_masked(keep[i], (col_mask[c >> 2] >> ((c & 3) * 8)) & 0xffu, 0u);
            quant_store8<BIT>(keep[i], s, rs, qrow, c, row, rows16, fmt);
        }
    }
    if (kept) return;                                                     // (x_out left in front of the quantise arithmetic)
    if (have_out) {
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            if (gcol[g] >= 0) {
                if (x_out) x_out[static_cast<size_t>(row) * ldo + t + g * TPR] = gval[g];
                xr[gcol[g]] = 0;                                           // the reference zeroes the caller's tensor in place
            }
        }
        for (int j = t + GQ * TPR; j < n; j += TPR) {
            const int c = ind[j];
            const uint16_t v = xr[c];
            if (x_out) x_out[static_cast<size_t>(row) * ldo + j] = v;
            xr[c] = 0;
        }
    }
    if (x_out) for (int j = (have_out ? n : 0) + t; j < ldo; j += TPR) x_out[static_cast<size_t>(row) * ldo + j] = 0;
}

__global__ __launch_bounds__(QT) void extract_kernel(uint16_t* __restrict__ x, int ldx, const int32_t* __restrict__ ind,
                                                     int n, uint16_t* __restrict__ x_out, int ldo)
{
    const int row = blockIdx.x;
    uint16_t* xr = x + static_cast<size_t>(row) * ldx;
    uint16_t* orow = x_out + static_cast<size_t>(row) * ldo;
    for (int j = threadIdx.x; j < ldo; j += QT) {
        uint16_t v = 0;
        if (j < n) { const int c = ind[j]; v = xr[c]; xr[c] = 0; }
        orow[j] = v;
    }
}

// Column-wise OR of |x| > thr over a slab of rows; 8 columns (16 bytes) per thread.
constexpr int DET_ROWS = 16;
__global__ __launch_bounds__(QT) void detect_cols_kernel(const uint16_t* __restrict__ x, int ldx, int M, int K,
                                                         float thr, uint8_t* __restrict__ colflags)
{
    const int c8 = blockIdx.x * QT + threadIdx.x;
    if (c8 * 8 >= K) return;
    const int r0 = blockIdx.y * DET_ROWS;
    const int r1 = (r0 + DET_ROWS < M) ? r0 + DET_ROWS : M;
    uint32_t hit = 0;
    for (int r = r0; r < r1; ++r) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + static_cast<size_t>(r) * ldx + c8 * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (fabsf(h2f(static_cast<uint16_t>(w[i] & 0xffffu))) > thr) hit |= 1u << (2 * i);
            if (fabsf(h2f(static_cast<uint16_t>(w[i] >> 16))) > thr)     hit |= 1u << (2 * i + 1);
        }
    }
    if (hit) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (hit & (1u << e)) colflags[c8 * 8 + e] = 1;
    }
}

// Single workgroup: ascending compaction of the set flags (sorted unique ids, as torch.unique returns).
constexpr int CT = 1024;
__global__ __launch_bounds__(CT) void compact_cols_kernel(const uint8_t* __restrict__ colflags, int K,
                                                          int32_t* __restrict__ ind_out, int32_t* __restrict__ count)
{
    __shared__ int part[CT];
    const int tid = threadIdx.x;
    const int per = (K + CT - 1) / CT;
    const int c0 = tid * per, c1 = (c0 + per < K) ? c0 + per : K;
    int cnt = 0;
    for (int c = c0; c < c1; ++c) cnt += colflags[c] ? 1 : 0;
    part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < CT; off <<= 1) {          // Hillis-Steele inclusive scan
        int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int pos = part[tid] - cnt;
    for (int c = c0; c < c1; ++c) if (colflags[c]) ind_out[pos++] = c;
    if (tid == CT - 1) *count = part[CT - 1];
}

template <int BIT>
__global__ __launch_bounds__(QT) void dequant_cols_kernel(const uint8_t* __restrict__ w, const uint16_t* __restrict__ scale_col,
                                                          const int32_t* __restrict__ ind, int n, uint16_t* __restrict__ out,
                                                          int N, int K, int ldo)
{
    const long long t = static_cast<long long>(blockIdx.x) * QT + threadIdx.x;
    if (t >= static_cast<long long>(N) * n) return;
    const int r = static_cast<int>(t / n), j = static_cast<int>(t % n);
    const int c = ind[j];
    int v;
    if constexpr (BIT == 8) {
        v = static_cast<int8_t>(w[static_cast<size_t>(r) * K + c]);
    } else {
        const uint8_t b = w[static_cast<size_t>(r) * (K >> 1) + (c >> 1)];
        const int nib = (c & 1) ? (b >> 4) : (b & 0xf);
        v = nib >= 8 ? nib - 16 : nib;
    }
    float f = static_cast<float>(v);
    if (scale_col) f = h2f(f2h(f)) * h2f(scale_col[r]);     // fp16 * fp16 -> one rounding, as torch's half multiply
    out[static_cast<size_t>(r) * ldo + j] = f2h(f);
}

// Re-tile a plain [R,KB] byte matrix into a packed format (UNPACK: the inverse); one 16-byte chunk per thread in the
// order of the packed image, rows >= R are zero-filled when packing and skipped when unpacking.
template <bool UNPACK>
__global__ __launch_bounds__(QT) void repack_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int R, int KB,
                                                    int rows16, int fmt)
{
    const long long t = static_cast<long long>(blockIdx.x) * QT + threadIdx.x;
    const long long total = static_cast<long long>(rows16) * (KB >> 4);
    if (t >= total) return;
    const int per_kb = rows16 * 4;
    const int kb = static_cast<int>(t / per_kb), rem = static_cast<int>(t % per_kb);
    const int rb = rem >> 6;
    int r, c;
    if (fmt == MIXQ_FMT_F16X64) { c = (rem >> 4) & 3; r = rem & 15; }
    else                        { r = (rem >> 2) & 15; c = (rem & 3) ^ ((0 - (r >> 2)) & 3); }
    const int row = rb * 16 + r;
    const size_t plain = static_cast<size_t>(row) * KB + kb * 64 + c * 16;
    if constexpr (UNPACK) {
        if (row < R) *reinterpret_cast<uint4*>(dst + plain) = reinterpret_cast<const uint4*>(src)[t];
    } else {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < R) v = *reinterpret_cast<const uint4*>(src + plain);
        reinterpret_cast<uint4*>(dst)[t] = v;
    }
}

// The same for MIXQ_FMT_F6X128: one thread per lane fragment (row, 32-element group): 16 bytes of nibbles <-> 24 bytes of FP6 codes.
template <bool UNPACK>
__global__ __launch_bounds__(QT) void repack_f6_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int R, int K, int rows16, int fmt)
{
    const long long t = static_cast<long long>(blockIdx.x) * QT + threadIdx.x;
    const long long total = static_cast<long long>(rows16) * (K >> 5);
    if (t >= total) return;
    const int lane = static_cast<int>(t & 63);
    const long long b = t >> 6;                                   // block index: kb128 * (rows16 / 16) + rb
    const int nrb = rows16 >> 4, kb = static_cast<int>(b / nrb), rb = static_cast<int>(b - static_cast<long long>(kb) * nrb);
    const int row = rb * 16 + (lane & 15), k0 = kb * 128 + (lane >> 4) * 32;
    const size_t blk = static_cast<size_t>(b) * 1536;
    const size_t plain = static_cast<size_t>(row) * (K >> 1) + (k0 >> 1);
    if constexpr (UNPACK) {
        if (row >= R) return;
        uint8_t *pa, *pb;
        f6_pieces(fmt, const_cast<uint8_t*>(src) + blk, row, lane >> 4, pa, pb);
        const uint4 a = *reinterpret_cast<const uint4*>(pa);
        const uint2 c = *reinterpret_cast<const uint2*>(pb);
        const uint32_t w[6] = {a.x, a.y, a.z, a.w, c.x, c.y};
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int bit = 6 * e, wi = bit >> 5, sh = bit & 31;
            uint32_t code = w[wi] >> sh;
            if (sh > 26) code |= w[wi + 1] << (32 - sh);
            o[e >> 3] |= f6_nibble_of_code(code & 0x3fu) << (4 * (e & 7));
        }
        *reinterpret_cast<uint4*>(dst + plain) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < R) v = *reinterpret_cast<const uint4*>(src + plain);
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t w[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const uint32_t code = f6_code_of_nibble((in[e >> 3] >> (4 * (e & 7))) & 0xfu);
            const int bit = 6 * e, wi = bit >> 5, sh = bit & 31;
            w[wi] |= code << sh;
            if (sh > 26) w[wi + 1] |= code >> (32 - sh);
        }
        uint8_t *pa, *pb;
        f6_pieces(fmt, dst + blk, row, lane >> 4, pa, pb);
        *reinterpret_cast<uint4*>(pa) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint2*>(pb) = make_uint2(w[4], w[5]);
    }
}

// Launch geometry of the extract + scale + quantise pass: 0 = the round-1 kernel (256 threads, one row per workgroup), 1.. =
// quant_rows2_kernel as (threads per row, rows per workgroup); -1 = choose by shape.  Tuning / test knob: mixq_quant_set_config.
MixqDevInt g_quant_cfg(-1);             // per device (common.h)
#ifdef MIXQ_TUNING
int g_quant_dbg = 0;                    // timing probes (bits: 1 no in-place zeroing, 2 no gather loads, 4 no outlier handling, 8 no x_out stores, 32 no quantise arithmetic, 64 positions loaded but nothing extracted): tools build only
#else
constexpr int g_quant_dbg = 0;
#endif
constexpr int NUM_QUANT_CFGS = 10;

template <int BIT, int TPR, int RPB>
int launch_quant_rows2(uint16_t* x, int ldx, const int32_t* ind, int n, const int32_t* n_dev, uint16_t* x_scale, void* q,
                       uint16_t* x_out, int ldo, int32_t* flag, int M, int K, float thr_scale, int qfmt, hipStream_t st,
                       const uint32_t* col_mask)
{
    const int nchunk = K >> 3;
    // kept route: the reduction floats (padded to 16 bytes) and one fp16 image per row; otherwise the column bitmask and the floats
    if (static_cast<size_t>(RPB) * K * 2 > 60000) col_mask = nullptr;       // (the images stay inside the default 64 KB of dynamic LDS: beyond, the in-kernel mask)
    const size_t shm = col_mask ? static_cast<size_t>((RPB * (TPR / 64) + 3) & ~3) * sizeof(uint32_t) + static_cast<size_t>(RPB) * K * 2
                                : (static_cast<size_t>((K + 31) >> 5) + RPB * (TPR / 64)) * sizeof(uint32_t);
    const int rows16 = qfmt ? ((M + 15) & ~15) : 0;
    dim3 g((M + RPB - 1) / RPB), b(TPR * RPB);
#define MIXQ_QLAUNCH2K(NCH, KEPT) hipLaunchKernelGGL((quant_rows2_kernel<BIT, TPR, RPB, NCH, KEPT>), g, b, shm, st, x, col_mask, n_dev, ind, ldx, M, K, n, x_scale, q, x_out, ldo, flag, thr_scale, rows16, qfmt, g_quant_dbg)
#define MIXQ_QLAUNCH2(NCH) do { if (col_mask) MIXQ_QLAUNCH2K(NCH, true); else MIXQ_QLAUNCH2K(NCH, false); } while (0)
    if      (nchunk <= 1 * TPR)  MIXQ_QLAUNCH2(1);
    else if (nchunk <= 2 * TPR)  MIXQ_QLAUNCH2(2);
    else if (BIT == 8 && nchunk <= 3 * TPR) { if constexpr (BIT == 8) MIXQ_QLAUNCH2(3); }   // (K = 11008 at 512 threads: 2.7 chunks per thread; the FP6 forms pair chunks)
    else if (nchunk <= 4 * TPR)  MIXQ_QLAUNCH2(4);
    else if (nchunk <= 8 * TPR)  MIXQ_QLAUNCH2(8);
    else if (nchunk <= 16 * TPR) MIXQ_QLAUNCH2(16);
    else return -100;                                     // the caller falls back to the any-K kernel
#undef MIXQ_QLAUNCH2K
#undef MIXQ_QLAUNCH2
    return mixq_launch_status();
}

template <int BIT>
int launch_quant_rows(uint16_t* x, int ldx, const int32_t* ind, int n, const int32_t* n_dev, uint16_t* x_scale, void* q,
                      uint16_t* x_out, int ldo, int32_t* flag, int M, int K, float thr_scale, int qfmt, hipStream_t st,
                      const uint32_t* col_mask = nullptr)      // (optional: the kept outlier map of `ind`; the round-1 kernel builds its own mask)
{
    const int nchunk = K >> 3;
    int cfg = g_quant_cfg.get();
    if (cfg < 0) {
        // measured on an MI355X (tools/time_quant.py, profiles/r02_quant_sweep.txt), M = 512, K = 4096, 41 outlier columns, us per
        // launch in a graph: (256,1) 5.3 | (256,2) 5.4 | round-1 kernel 5.4 | (128,2) 6.6 | (128,1) 7.1 | (64,x) 9.2-9.4.  A row in ONE wave
        // (no LDS reduction, no barrier) loses: its 64 elements per lane make the quantise arithmetic, not memory, the critical
        // path.  256 threads per row it is; the gain over round 1 is the gather from global memory instead of an LDS row copy.
        cfg = nchunk >= 512 ? 8 : 6;                     // (512 threads per row from K = 4096 up: 5.3 vs 5.7 us)
        // FP6 codes leave in chunk pairs (2 chunks per thread at least); two rows per workgroup.  R6X128 (row-major blocks) at K = 4096:
        // (256,2) 6.31 | (512,1) 6.25 | (512,2) 6.11 us, at 11008 (512,x) 11.2-11.3 | (256,x) 12.2-12.3 (profiles/r03_w4a4_quant_formats.txt)
        if (qfmt == MIXQ_FMT_F6X128 || qfmt == MIXQ_FMT_R6X128) cfg = nchunk <= 256 ? 5 : (qfmt == MIXQ_FMT_F6X128 && nchunk <= 512 ? 7 : 9);
    }
    int rc = -100;
#define MIXQ_Q2(TPR, RPB) rc = launch_quant_rows2<BIT, TPR, RPB>(x, ldx, ind, n, n_dev, x_scale, q, x_out, ldo, flag, M, K, thr_scale, qfmt, st, col_mask)
    switch (cfg) {
        case 1: MIXQ_Q2(64, 1); break;
        case 2: MIXQ_Q2(64, 2); break;
        case 3: MIXQ_Q2(64, 4); break;
        case 4: MIXQ_Q2(128, 1); break;
        case 5: MIXQ_Q2(128, 2); break;
        case 6: MIXQ_Q2(256, 1); break;
        case 7: MIXQ_Q2(256, 2); break;
        case 8: MIXQ_Q2(512, 1); break;
        case 9: MIXQ_Q2(512, 2); break;
        default: break;
    }
#undef MIXQ_Q2
    if (rc != -100) return rc;
    // round-1 kernel: also the any-K form (NCH == 0 re-reads the row)
    // column bitmask, 4 reduction floats, and (rows kept in registers) a copy of the row for the outlier gather
    const size_t shm = ((static_cast<size_t>((K + 31) >> 5) + 4 + 3) & ~static_cast<size_t>(3)) * sizeof(uint32_t)
                       + (nchunk <= 8 * QT ? static_cast<size_t>(K) * 2 : 0);
    const int rows16 = qfmt ? ((M + 15) & ~15) : 0;
    dim3 g(M), b(QT);
#define MIXQ_QLAUNCH(NCH) hipLaunchKernelGGL((quant_rows_kernel<BIT, NCH>), g, b, shm, st, x, ldx, ind, n, n_dev, x_scale, q, x_out, ldo, flag, K, thr_scale, rows16, qfmt)
    if      (nchunk <= 2 * QT)  MIXQ_QLAUNCH(2);
    else if (nchunk <= 4 * QT)  MIXQ_QLAUNCH(4);
    else if (nchunk <= 8 * QT)  MIXQ_QLAUNCH(8);
    else if (nchunk <= 16 * QT) MIXQ_QLAUNCH(16);
    else                        MIXQ_QLAUNCH(0);
#undef MIXQ_QLAUNCH
    return mixq_launch_status();
}

inline float fp16_round(float v) {                    // host: value of fp16(v) as float (RNE), via the device-agnostic type
    return static_cast<float>(static_cast<_Float16>(v));
}

}  // namespace

// Self-test of quant_exact (common.h): every finite fp16 x against every finite fp16 scale s > 0, compared with the
// division form rint(RN(x / s)) it replaces.  One workgroup per scale value; *mismatches receives the count.
template <int BIT>
__global__ __launch_bounds__(256) void selftest_quant_exact_kernel(unsigned long long* __restrict__ mismatches)
{
    constexpr float QMAX = static_cast<float>((1 << (BIT - 1)) - 1);
    const float s = h2f(static_cast<uint16_t>(blockIdx.x + 1));          // bits 0x0001 .. 0x7bff
    const float rs = mixq_rcp_scale(s);
    unsigned long long bad = 0;
    for (unsigned xb = threadIdx.x; xb < 65536u; xb += 256) {
        if ((xb & 0x7c00u) == 0x7c00u) continue;                          // inf / nan
        const float x = h2f(static_cast<uint16_t>(xb));
        float q = rintf(__fdiv_rn(x, s));
        q = fminf(fmaxf(q, -QMAX), QMAX);
        bad += quant_exact<BIT>(x, s, rs) != static_cast<int>(q);
        // the chunk form (quant8_exact): this value in slot xb % 8 of a chunk whose other slots hold its neighbours' bit patterns (finite ones)
        uint32_t hw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const unsigned o = (xb + 8191u * (j + 1)) & 0xffffu; hw[j] = (o & 0x7c00u) == 0x7c00u ? 0u : o; }
        hw[xb & 7u] = xb;
        const uint4 v = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
        uint32_t ub[8];
        quant8_exact<BIT>(v, s, rs, ub);
        uint32_t got = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) if ((xb & 7u) == static_cast<unsigned>(j)) got = ub[j];
        bad += quant8_int(got) != static_cast<int>(q);
        bad += BIT == 8 && (got & 0xffu) != (static_cast<uint32_t>(static_cast<int>(q)) & 0xffu);
    }
    if (bad) atomicAdd(mismatches, bad);
}

extern "C" int mixq_selftest_quant_exact(unsigned long long* mismatches_dev, int bit, mixq_stream_t stream)
{
    if (!mismatches_dev || (bit != 8 && bit != 4)) return MIXQ_EINVAL;
    if (bit == 8) hipLaunchKernelGGL(selftest_quant_exact_kernel<8>, dim3(0x7bff), dim3(256), 0, mixq_stream(stream), mismatches_dev);
    else          hipLaunchKernelGGL(selftest_quant_exact_kernel<4>, dim3(0x7bff), dim3(256), 0, mixq_stream(stream), mismatches_dev);
    return mixq_launch_status();
}

extern "C" int mixq_find_row_scale(const uint16_t* x, uint16_t* x_scale, void* q, int M, int K, int ldx, int bit, int qfmt,
                                   mixq_stream_t stream)
{
    if (qfmt != MIXQ_FMT_PLAIN && qfmt != MIXQ_FMT_P16X64 && qfmt != MIXQ_FMT_F16X64 && !((qfmt == MIXQ_FMT_F6X128 || qfmt == MIXQ_FMT_R6X128) && bit == 4)) return MIXQ_EINVAL;
    if (qfmt != MIXQ_FMT_PLAIN && (bit == 8 ? K : K / 2) % 64) return MIXQ_ESHAPE;
    if (M < 0 || K <= 0 || (M > 0 && (!x || !x_scale || !q))) return MIXQ_EINVAL;   // empty inputs may carry null pointers
    if (bit != 8 && bit != 4) return MIXQ_EINVAL;
    if ((K & 7) || (ldx & 7) || ldx < K || (bit == 4 && (K & 15))) return MIXQ_ESHAPE;
    if (M == 0) return MIXQ_OK;
    uint16_t* xm = const_cast<uint16_t*>(x);           // not written: n = 0 and x_out = NULL
    if (bit == 8) return launch_quant_rows<8>(xm, ldx, nullptr, 0, nullptr, x_scale, q, nullptr, 0, nullptr, M, K, 0.f, qfmt, mixq_stream(stream));
    return launch_quant_rows<4>(xm, ldx, nullptr, 0, nullptr, x_scale, q, nullptr, 0, nullptr, M, K, 0.f, qfmt, mixq_stream(stream));
}

static int quant_fused_common(uint16_t* x, const int32_t* ind, int n, const int32_t* n_dev, const uint32_t* col_mask, uint16_t* x_scale, void* q,
                              uint16_t* x_out, int32_t* flag, int M, int K, int ldx, int ldo, int bit, float sigma, int qfmt,
                              mixq_stream_t stream)
{
    if (qfmt != MIXQ_FMT_PLAIN && qfmt != MIXQ_FMT_P16X64 && qfmt != MIXQ_FMT_F16X64 && !((qfmt == MIXQ_FMT_F6X128 || qfmt == MIXQ_FMT_R6X128) && bit == 4)) return MIXQ_EINVAL;
    if (qfmt != MIXQ_FMT_PLAIN && (bit == 8 ? K : K / 2) % 64) return MIXQ_ESHAPE;
    if (M < 0 || K <= 0 || n < 0 || (M > 0 && (!x || !x_scale || !q))) return MIXQ_EINVAL;
    if (bit != 8 && bit != 4) return MIXQ_EINVAL;
    if (n > 0 && (!ind || !x_out || ldo < n)) return MIXQ_EINVAL;
    if ((K & 7) || (ldx & 7) || ldx < K || (bit == 4 && (K & 15))) return MIXQ_ESHAPE;
    if (M == 0) return MIXQ_OK;
    const float qmax = static_cast<float>((1 << (bit - 1)) - 1);
    // reference: `x_scale.max() > self.sigma / qmax` with sigma an fp16 [1,1] tensor -> fp16(fp16(sigma)/qmax)
    const float thr = fp16_round(fp16_round(sigma) / qmax);
    uint16_t* xo = (n > 0) ? x_out : nullptr;
    if (K > 32768) col_mask = nullptr;                 // (the kept route keeps an fp16 image of the row in LDS: beyond 64 KB per row, the in-kernel mask)
    if (bit == 8) return launch_quant_rows<8>(x, ldx, ind, n, n_dev, x_scale, q, xo, ldo, flag, M, K, thr, qfmt, mixq_stream(stream), col_mask);
    return launch_quant_rows<4>(x, ldx, ind, n, n_dev, x_scale, q, xo, ldo, flag, M, K, thr, qfmt, mixq_stream(stream), col_mask);
}

extern "C" int mixq_quant_fused(uint16_t* x, const int32_t* ind, int n, const int32_t* n_dev, uint16_t* x_scale, void* q,
                                uint16_t* x_out, int32_t* flag, int M, int K, int ldx, int ldo, int bit, float sigma, int qfmt,
                                mixq_stream_t stream)
{
    return quant_fused_common(x, ind, n, n_dev, nullptr, x_scale, q, x_out, flag, M, K, ldx, ldo, bit, sigma, qfmt, stream);
}
// ... with the caller's kept outlier map of the live `ind` entries (include/mixq_hip.h: bit words, count word, per-column AND-masks): same
// bytes out from ONE memory round trip - no in-kernel mask build in front of the row maximum, no dependent gather behind the row load
// int32 words of the kept outlier map of K columns: bit words, the count word, pad to 4 words, K 16-bit AND-masks padded to whole 16-byte chunks
extern "C" int mixq_kept_map_words(int K) { return K > 0 ? ((((K + 31) >> 5) + 1 + 3) & ~3) + ((K + 7) >> 3) * 4 : 0; }
// (a map in another layout than this library's - round 4's stopped behind the count word - must not reach the kernels: they read the AND-masks
// up to 2 K bytes behind the bit words)
static bool kept_map_ok(const uint32_t* col_mask, int map_words, int K) {
    return col_mask && map_words >= mixq_kept_map_words(K) && (reinterpret_cast<uintptr_t>(col_mask) & 15) == 0;
}
extern "C" int mixq_quant_fused_masked(uint16_t* x, const int32_t* ind, int n, const int32_t* n_dev, const uint32_t* col_mask, int map_words,
                                       uint16_t* x_scale, void* q, uint16_t* x_out, int32_t* flag, int M, int K, int ldx, int ldo,
                                       int bit, float sigma, int qfmt, mixq_stream_t stream)
{
    if (n > 0 && !kept_map_ok(col_mask, map_words, K)) return MIXQ_EINVAL;
    return quant_fused_common(x, ind, n, n_dev, n > 0 ? col_mask : nullptr, x_scale, q, x_out, flag, M, K, ldx, ldo, bit, sigma, qfmt, stream);
}

extern "C" int mixq_quant_known_amax(uint16_t* x, const int32_t* ind, int n, const int32_t* n_dev, uint32_t* row_amax,
                                     const uint32_t* col_mask, int map_words, uint16_t* x_scale, void* q, uint16_t* x_out, int32_t* flag, int M, int K,
                                     int ldx, int ldo, int bit, float sigma, int qfmt, mixq_stream_t stream)
{
    if (col_mask && K > 0 && !kept_map_ok(col_mask, map_words, K)) return MIXQ_EINVAL;
    if (qfmt != MIXQ_FMT_PLAIN && qfmt != MIXQ_FMT_P16X64 && qfmt != MIXQ_FMT_F16X64 && !((qfmt == MIXQ_FMT_F6X128 || qfmt == MIXQ_FMT_R6X128) && bit == 4)) return MIXQ_EINVAL;
    if (qfmt != MIXQ_FMT_PLAIN && (bit == 8 ? K : K / 2) % 64) return MIXQ_ESHAPE;
    if (M < 0 || K <= 0 || n < 0 || (M > 0 && (!x || !x_scale || !q || !row_amax))) return MIXQ_EINVAL;
    if (bit != 8 && bit != 4) return MIXQ_EINVAL;
    if (n > 0 && (!ind || !x_out || ldo < n || !col_mask)) return MIXQ_EINVAL;      // outlier columns need their bit mask
    if ((K & 7) || (ldx & 7) || ldx < K || (bit == 4 && (K & 15))) return MIXQ_ESHAPE;
    if (K > 16 * 512 * 8) return MIXQ_ESHAPE;                                       // a row lives in the registers of one workgroup
    if (M == 0) return MIXQ_OK;
    const float qmax = static_cast<float>((1 << (bit - 1)) - 1);
    const float thr = fp16_round(fp16_round(sigma) / qmax);
    uint16_t* xo = (n > 0) ? x_out : nullptr;
    const int rows16 = qfmt ? ((M + 15) & ~15) : 0, nchunk = K >> 3;
    hipStream_t st = mixq_stream(stream);
#define MIXQ_QK(BITv, TPRv, NCHv) hipLaunchKernelGGL((quant_known_kernel<BITv, TPRv, NCHv>), dim3(M), dim3(TPRv), static_cast<size_t>(K <= 30000 ? K : 0) * 2 + 16, st, x, col_mask, row_amax, n_dev, ind, \
                                                     ldx, K, n, ldo, x_scale, q, xo, flag, thr, rows16, qfmt)
#define MIXQ_QK_BY_SIZE(BITv)                                             \
    if      (nchunk <= 256)      MIXQ_QK(BITv, 256, 1);                   \
    else if (nchunk <= 512)      MIXQ_QK(BITv, 512, 1);                   \
    else if (nchunk <= 2 * 512)  MIXQ_QK(BITv, 512, 2);                   \
    else if (nchunk <= 3 * 512)  MIXQ_QK(BITv, 512, 3);                   \
    else if (nchunk <= 4 * 512)  MIXQ_QK(BITv, 512, 4);                   \
    else if (nchunk <= 8 * 512)  MIXQ_QK(BITv, 512, 8);                   \
    else                         MIXQ_QK(BITv, 512, 16);
    if (bit == 8) { MIXQ_QK_BY_SIZE(8) } else { MIXQ_QK_BY_SIZE(4) }
#undef MIXQ_QK_BY_SIZE
#undef MIXQ_QK
    return mixq_launch_status();
}

extern "C" int mixq_extract_outliers_zero(uint16_t* x, const int32_t* ind, int n, uint16_t* x_out, int M, int K, int ldx,
                                          int ldo, mixq_stream_t stream)
{
    if (M < 0 || K <= 0 || n < 0 || (n > 0 && !ind) || ldo < n || ldx < K) return MIXQ_EINVAL;
    if (M > 0 && ldo > 0 && (!x || !x_out)) return MIXQ_EINVAL;
    if (M == 0 || ldo == 0) return MIXQ_OK;
    hipLaunchKernelGGL(extract_kernel, dim3(M), dim3(QT), 0, mixq_stream(stream), x, ldx, ind, n, x_out, ldo);
    return mixq_launch_status();
}

extern "C" int mixq_detect_outlier_cols(const uint16_t* x, float sigma, uint8_t* colflags, int32_t* ind_out, int32_t* count,
                                        int M, int K, int ldx, mixq_stream_t stream)
{
    if (!x || !colflags || !ind_out || !count || M < 0 || K <= 0) return MIXQ_EINVAL;
    if ((K & 7) || (ldx & 7) || ldx < K) return MIXQ_ESHAPE;
    hipStream_t st = mixq_stream(stream);
    hipError_t e = hipMemsetAsync(colflags, 0, K, st);
    if (e != hipSuccess) return static_cast<int>(e);
    if (M > 0) {
        dim3 g((K / 8 + QT - 1) / QT, (M + DET_ROWS - 1) / DET_ROWS);
        hipLaunchKernelGGL(detect_cols_kernel, g, dim3(QT), 0, st, x, ldx, M, K, fp16_round(sigma), colflags);
        int rc = mixq_launch_status();
        if (rc) return rc;
    }
    hipLaunchKernelGGL(compact_cols_kernel, dim3(1), dim3(CT), 0, st, colflags, K, ind_out, count);
    return mixq_launch_status();
}

extern "C" int mixq_dequant_weight_cols(const void* w, const uint16_t* scale_col, const int32_t* ind, int n, uint16_t* out,
                                        int N, int K, int ldo, int bit, mixq_stream_t stream)
{
    if (!w || !out || N < 0 || K <= 0 || n < 0 || (n > 0 && !ind) || ldo < n) return MIXQ_EINVAL;
    if (bit != 8 && bit != 4) return MIXQ_EINVAL;
    if (N == 0 || n == 0) return MIXQ_OK;
    const long long total = static_cast<long long>(N) * n;
    dim3 g(static_cast<unsigned>((total + QT - 1) / QT));
    const uint8_t* wb = static_cast<const uint8_t*>(w);
    if (bit == 8) hipLaunchKernelGGL(dequant_cols_kernel<8>, g, dim3(QT), 0, mixq_stream(stream), wb, scale_col, ind, n, out, N, K, ldo);
    else          hipLaunchKernelGGL(dequant_cols_kernel<4>, g, dim3(QT), 0, mixq_stream(stream), wb, scale_col, ind, n, out, N, K, ldo);
    return mixq_launch_status();
}

static int repack_common(const void* src, void* dst, int R, int KB, int fmt, bool unpack, mixq_stream_t stream)
{
    if (!src || !dst || R < 0 || KB <= 0) return MIXQ_EINVAL;
    if (fmt != MIXQ_FMT_P16X64 && fmt != MIXQ_FMT_F16X64 && fmt != MIXQ_FMT_F6X128 && fmt != MIXQ_FMT_R6X128) return MIXQ_EINVAL;
    if (KB % 64) return MIXQ_ESHAPE;
    if (R == 0) return MIXQ_OK;
    const int rows16 = (R + 15) & ~15;
    if (fmt == MIXQ_FMT_F6X128 || fmt == MIXQ_FMT_R6X128) {  // KB = K / 2 bytes of nibbles per row of the plain side
        const int K = KB * 2;
        const long long total6 = static_cast<long long>(rows16) * (K >> 5);
        const dim3 g6(static_cast<unsigned>((total6 + QT - 1) / QT));
        if (unpack) hipLaunchKernelGGL(repack_f6_kernel<true>, g6, dim3(QT), 0, mixq_stream(stream), static_cast<const uint8_t*>(src),
                                       static_cast<uint8_t*>(dst), R, K, rows16, fmt);
        else        hipLaunchKernelGGL(repack_f6_kernel<false>, g6, dim3(QT), 0, mixq_stream(stream), static_cast<const uint8_t*>(src),
                                       static_cast<uint8_t*>(dst), R, K, rows16, fmt);
        return mixq_launch_status();
    }
    const long long total = static_cast<long long>(rows16) * (KB >> 4);
    const dim3 g(static_cast<unsigned>((total + QT - 1) / QT));
    if (unpack) hipLaunchKernelGGL(repack_kernel<true>, g, dim3(QT), 0, mixq_stream(stream), static_cast<const uint8_t*>(src),
                                   static_cast<uint8_t*>(dst), R, KB, rows16, fmt);
    else        hipLaunchKernelGGL(repack_kernel<false>, g, dim3(QT), 0, mixq_stream(stream), static_cast<const uint8_t*>(src),
                                   static_cast<uint8_t*>(dst), R, KB, rows16, fmt);
    return mixq_launch_status();
}

extern "C" int mixq_pack_p16x64(const void* src, void* dst, int R, int KB, mixq_stream_t stream)
{
    return repack_common(src, dst, R, KB, MIXQ_FMT_P16X64, false, stream);
}

extern "C" int mixq_pack_operand(const void* src, void* dst, int R, int KB, int fmt, mixq_stream_t stream)
{
    return repack_common(src, dst, R, KB, fmt, false, stream);
}

extern "C" int mixq_unpack_operand(const void* src, void* dst, int R, int KB, int fmt, mixq_stream_t stream)
{
    return repack_common(src, dst, R, KB, fmt, true, stream);
}

extern "C" int mixq_quant_set_config(int cfg)
{
#ifdef MIXQ_TUNING
    if (cfg >= 100) { g_quant_dbg = cfg - 100; return MIXQ_OK; }      // (tools/time_quant.py --probe: results are then wrong on purpose)
#endif
    if (cfg < -1 || cfg >= NUM_QUANT_CFGS) return MIXQ_EINVAL;
    g_quant_cfg.set(cfg);
    return MIXQ_OK;
}
