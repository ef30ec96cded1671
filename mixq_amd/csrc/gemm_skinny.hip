// Small-batch ("decode") form of the fused W8A8 / W4A4 GEMM for gfx950: M <= 32 token rows, both operands in P16x64.
//
// At M <= 32 the GEMM is a weight stream: 2*M MACs per weight byte, i.e. HBM-/L2-bound, and the tiled kernel of gemm.hip
// spends its time on per-k-step pipeline overhead (one barrier per 64 bytes of k for two MFMAs per wave) and cannot put
// more than N/64 workgroups on the chip.  Here instead
//   * one workgroup owns 32 output channels (two adjacent 16-row P16x64 weight blocks: 2 KiB contiguous per k-step) and
//     ALL of K; its 8 waves split K among themselves, so N/32 workgroups x 8 waves stream the weights with every CU busy;
//   * no LDS staging and no barriers in the k loop: a lane's MFMA fragment IS a 16-byte piece of the packed image, so
//     weights and activations go global -> VGPR -> v_mfma_i32_32x32x32_i8 directly (the activations are a few hundred KB
//     and stay in L2), with the loads of the next 4 k-steps in flight behind the MFMAs of the current 4;
//   * the 8 partial 32x32 int32 accumulators are summed through LDS (integer adds: exact, order-free, so the result is
//     bit-identical to the tiled kernel's accumulator), then wave 0 runs the same epilogue arithmetic as gemm.hip:
//     acc * x_scale[m] * scale_col[n], fp16 outlier MFMA tail, addend, SiLU, bias, one rounding to fp16.
// Measured on MI355X with HBM-cold weights (tools/time_decode.py): 4096x4096 6.6 us, 4096->11008 14-15 us (3.0-3.2 TB/s of
// weights incl. ~4 us of launch + reduction + epilogue; the stream itself runs at ~5 TB/s), 1.3-2.5x the tiled kernel.
// A [N/32][K/64] block order (contiguous per-workgroup stream) and 2 workgroups per CU were tried: no difference.
// Reference call sites replaced: mixlib.int8FusedDequantize[Silu] (modules/linear.py:235-283, :321-366) at small batch.
#include "common.h"
#include "gemm_skinny.h"
#include <type_traits>

namespace {

struct SkinnyArgs {
    const uint8_t* qx; const uint8_t* qw;
    const uint16_t* sx; const uint16_t* sw; const uint16_t* xo; const uint16_t* wo;
    const int32_t* n_out_dev; const uint16_t* addend; const uint16_t* bias; uint16_t* y;
    int M, N, KB, ldxo, ldwo, n_out, lda, ldy, act, xrows16, wrows16;
    int wf16;                                            // weights in MIXQ_FMT_F16X64 instead of MIXQ_FMT_P16X64
};

constexpr int SKW = 8;                                   // waves per workgroup = k splits

__device__ __forceinline__ int sk_swz(int r, int c) { return c ^ ((0 - (r >> 2)) & 3); }   // P16X64 chunk swizzle (common.h)
__device__ __forceinline__ float sk_silu(float v) { return mixq_silu(v); }   // (common.h: never contracted with the bias addition)

// UNROLL: k-steps whose loads are in flight together (x2 buffers); MINW: waves per SIMD the register budget must allow.
// I4: both operands nibble-packed int4 (KB = K/2 bytes per row): every 16-byte fragment is expanded in registers to the
// two int8 fragments 16*q of its even / odd columns (as gemm.hip does) and the factor 256 leaves in the epilogue.
template <int UNROLL, int MINW, bool I4>
__global__ __launch_bounds__(SKW * 64, MINW) void gemm_skinny_kernel(const SkinnyArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[SKW * 4096];    // [wave][4 reg groups][64 lanes] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = wave_id_uniform();
    const int n0 = blockIdx.x * 32;
    const int nk = a.KB / 64;
    // this wave's k-steps: an even split of nk over the SKW waves
    const int k_lo = (nk * wave) / SKW, k_hi = (nk * (wave + 1)) / SKW;

    int n_out_dev_v = 0;
    if (a.n_out_dev) n_out_dev_v = *a.n_out_dev;

    // fragment addresses: row r of 16-row block rb, k-step kb: ((kb * rows16/16 + rb) * 1024) + (r & 15) * 64 + chunk * 16
    int wr = n0 + lr; wr = wr < a.N ? wr : a.N - 1;                      // clamped rows are computed and dropped
    int xr = lr < a.M ? lr : a.M - 1;
    // P16X64: row r at r*64, chunk c at (c ^ swizzle)*16;  F16X64: chunk c at c*256, row r at r*16
    const uint8_t* wp = a.qw + static_cast<size_t>(wr >> 4) * 1024 + (wr & 15) * (a.wf16 ? 16 : 64);
    const uint8_t* xp = a.qx + static_cast<size_t>(xr >> 4) * 1024 + (xr & 15) * 64;
    const int wc0 = a.wf16 ? lh * 256 : sk_swz(wr, lh) * 16, wc1 = a.wf16 ? (2 + lh) * 256 : sk_swz(wr, 2 + lh) * 16;   // sub-steps 0 / 1: chunks lh / 2 + lh
    const int xc0 = sk_swz(xr, lh) * 16, xc1 = sk_swz(xr, 2 + lh) * 16;
    const size_t wks = static_cast<size_t>(a.wrows16) * 64, xks = static_cast<size_t>(a.xrows16) * 64;

    i32x16 acc, acc1;                                    // two chains: consecutive MFMAs never wait on each other
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0; acc1[r] = 0; }

    i32x4 wf[2][UNROLL][2], xf[2][UNROLL][2];
    auto load_group = [&](auto p_c, int kb) {            // k-steps kb .. kb+UNROLL-1 (beyond k_hi: clamped, never used)
        constexpr int P = decltype(p_c)::value;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int k = kb + u; k = k < k_hi ? k : k_hi - 1;
            const uint8_t* w = wp + k * wks;
            const uint8_t* x = xp + k * xks;
            wf[P][u][0] = *reinterpret_cast<const i32x4*>(w + wc0);
            wf[P][u][1] = *reinterpret_cast<const i32x4*>(w + wc1);
            xf[P][u][0] = *reinterpret_cast<const i32x4*>(x + xc0);
            xf[P][u][1] = *reinterpret_cast<const i32x4*>(x + xc1);
        }
    };
    auto mma_group = [&](auto p_c, int kb) {
        constexpr int P = decltype(p_c)::value;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (kb + u < k_hi) {                         // wave-uniform
                if constexpr (!I4) {
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[P][u][0], xf[P][u][0], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[P][u][1], xf[P][u][1], acc1, 0, 0, 0);
                } else {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        i32x4 wl, wh, xl, xh;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const uint32_t w = static_cast<uint32_t>(wf[P][u][sb][d]), x = static_cast<uint32_t>(xf[P][u][sb][d]);
                            wl[d] = static_cast<int>((w << 4) & 0xf0f0f0f0u); wh[d] = static_cast<int>(w & 0xf0f0f0f0u);
                            xl[d] = static_cast<int>((x << 4) & 0xf0f0f0f0u); xh[d] = static_cast<int>(x & 0xf0f0f0f0u);
                        }
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wl, xl, acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wh, xh, acc1, 0, 0, 0);
                    }
                }
            }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if (k_lo < k_hi) {
        load_group(P0{}, k_lo);
        for (int kb = k_lo; kb < k_hi; kb += 2 * UNROLL) {
            if (kb + UNROLL < k_hi) load_group(P1{}, kb + UNROLL);
            mma_group(P0{}, kb);
            if (kb + UNROLL < k_hi) {
                if (kb + 2 * UNROLL < k_hi) load_group(P0{}, kb + 2 * UNROLL);
                mma_group(P1{}, kb + UNROLL);
            }
        }
    }

    // ---- cross-wave sum (exact) ----------------------------------------------------------------------------------
    acc += acc1;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<i32x4*>(lds + (wave * 4 + g) * 1024 + lane * 16) = i32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        i32x4 s = *reinterpret_cast<const i32x4*>(lds + g * 1024 + lane * 16);
#pragma unroll
        for (int w = 1; w < SKW; ++w) {
            const i32x4 v = *reinterpret_cast<const i32x4*>(lds + (w * 4 + g) * 1024 + lane * 16);
            s += v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = s[e];
    }

    // ---- epilogue (same arithmetic and order as gemm.hip) ----------------------------------------------------------
    int n_out = a.n_out;
    if (a.n_out_dev) n_out = n_out_dev_v < n_out ? n_out_dev_v : n_out;
    if (!a.xo || !a.wo) n_out = 0;
    const int ksteps = (n_out + 15) >> 4;
    auto unpack4 = [](u32x2 v, float* o) {
        o[0] = h2f(static_cast<uint16_t>(v.x & 0xffffu)); o[1] = h2f(static_cast<uint16_t>(v.x >> 16));
        o[2] = h2f(static_cast<uint16_t>(v.y & 0xffffu)); o[3] = h2f(static_cast<uint16_t>(v.y >> 16));
    };
    const int m = lr;                                                   // this lane's token row
    const int mc = m < a.M ? m : a.M - 1;
    const float sxv = h2f(a.sx[mc]) * (I4 ? (1.f / 256.f) : 1.f);
    const int nloc = 4 * lh;                                            // columns nloc + 8 g + e
    float swv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + nloc + 8 * g;
        unpack4(*reinterpret_cast<const u32x2_u*>(a.sw + (n < a.N ? n : a.N - 4)), swv + 4 * g);
    }
    f32x16 f;
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = static_cast<float>(acc[r]) * sxv * swv[r];

    if (ksteps > 0) {
        const uint16_t* wop = a.wo + static_cast<size_t>(wr) * a.ldwo + lh * 8;
        const uint16_t* xop = a.xo + static_cast<size_t>(mc) * a.ldxo + lh * 8;
        for (int kk = 0; kk < ksteps; ++kk) {
            u32x4 wq = *reinterpret_cast<const u32x4*>(wop + kk * 16);
            u32x4 xq = *reinterpret_cast<const u32x4*>(xop + kk * 16);
            const int kb = kk * 16 + lh * 8;                            // mask columns >= n_out (the pad may hold anything)
            if (kb + 8 > n_out) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t keep = 0;
                    if (kb + 2 * d < n_out)     keep |= 0x0000ffffu;
                    if (kb + 2 * d + 1 < n_out) keep |= 0xffff0000u;
                    wq[d] &= keep; xq[d] &= keep;
                }
            }
            f = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wq), __builtin_bit_cast(f16x8, xq), f, 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + nloc + 8 * g;
        const int nc = n < a.N ? n : a.N - 4;                          // N % 4 == 0: groups are all in or all out
        float v[4] = {f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]};
        float av[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.addend) {
            unpack4(*reinterpret_cast<const u32x2_u*>(a.addend + static_cast<size_t>(mc) * a.lda + nc), av);
            if (a.act != MIXQ_ACT_SILU_MUL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += av[e];
            }
        }
        if (a.act != MIXQ_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = sk_silu(v[e]);
        }
        if (a.bias) {
            float bv[4];
            unpack4(*reinterpret_cast<const u32x2_u*>(a.bias + nc), bv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
        }
        if (a.act == MIXQ_ACT_SILU_MUL) {                          // addend is the multiplier; the bias is already in:
#pragma unroll                                                     // (silu(z) + bias) * up, linear.py:372-373 then mlp.py:61
            for (int e = 0; e < 4; ++e) v[e] *= av[e];
        }
        if (m < a.M && n < a.N) {
            u32x2 o;
            o.x = static_cast<uint32_t>(f2h(v[0])) | (static_cast<uint32_t>(f2h(v[1])) << 16);
            o.y = static_cast<uint32_t>(f2h(v[2])) | (static_cast<uint32_t>(f2h(v[3])) << 16);
            *reinterpret_cast<u32x2_u*>(a.y + static_cast<size_t>(m) * a.ldy + n) = o;
        }
    }
}

}  // namespace

bool mixq_skinny_applies(int bit, int M, int N, int KB, bool x_packed, bool w_packed)
{
    return (bit == 8 || bit == 4) && M >= 1 && M <= 32 && x_packed && w_packed && (KB % 64) == 0 && N >= 4;
}

int mixq_skinny_launch(int bit, const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col, const uint16_t* x_out,
                       int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev, const uint16_t* addend,
                       int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int KB, int act, int wf16, hipStream_t st)
{
    SkinnyArgs a;
    a.qx = static_cast<const uint8_t*>(q_x); a.qw = static_cast<const uint8_t*>(q_w);
    a.sx = x_scale; a.sw = scale_col; a.xo = x_out; a.wo = w_out; a.n_out_dev = n_out_dev; a.addend = addend; a.bias = bias; a.y = y;
    a.M = M; a.N = N; a.KB = KB; a.ldxo = ldxo; a.ldwo = ldwo; a.n_out = n_out; a.lda = lda; a.ldy = ldy; a.act = act;
    a.xrows16 = (M + 15) & ~15; a.wrows16 = (N + 15) & ~15; a.wf16 = wf16;
    if (bit == 8) hipLaunchKernelGGL((gemm_skinny_kernel<4, 2, false>), dim3((N + 31) / 32), dim3(SKW * 64), 0, st, a);
    else          hipLaunchKernelGGL((gemm_skinny_kernel<4, 2, true>), dim3((N + 31) / 32), dim3(SKW * 64), 0, st, a);
    return mixq_launch_status();
}
