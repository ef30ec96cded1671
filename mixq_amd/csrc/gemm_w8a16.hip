// Weight-only W8A16 Linear for gfx950:  Y[M,N] = X[M,K] (fp16) * (Wq[K,N] (int8) * scale_col[N]) + bias
// -- what the reference reaches through EETQ's `w8_a16_gemm(x, q_weight, scale_col)` (modules/linear.py:178-184; GPT-J
// fc_out and the "down_weight_only" variants, utils/module.py:4-12).  EETQ is a third-party CUDA extension that is not
// part of the reference tree and is not version-pinned; its CUTLASS-interleaved weight image is an implementation
// detail of that library.  Here:
//
//  * the int8 weights are re-tiled ONCE (mixq_pack_w8a16) from the checkpoint's [K,N] matrix into the same P16x64
//    tile-major layout the W8A8 GEMM streams (rows = output channels, 64 k-bytes per block row), stored offset-binary
//    (q + 128) so that the int8 -> fp16 conversion in the k loop is two VALU per pair of elements:
//        v_perm_b32   -> halves 0x6400 | u   (= 1024 + u exactly)
//        v_pk_add_f16 -> - 1152              (= u - 128 = q, exact)
//  * X stays the caller's fp16 [M,K] matrix; a k-step (64 elements = 128 bytes per row) of a tile is staged by LDS-DMA
//    in pieces of 8 rows x 128 bytes (whole 128-byte row segments feed a CU ~1.5x faster than 64-byte ones, DESIGN.md
//    section 6) with an 8-chunk XOR swizzle (chunk c of row r at c ^ ((r >> 1) & 7)) applied on the source address,
//  * v_mfma_f32_32x32x16_f16, weights as the A operand (a lane holds 4 consecutive output columns of one token, as in
//    gemm.hip, so the epilogue layout is shared).  One 16-byte LDS read of a weight row feeds two MFMA k-steps; the k
//    order inside a stage is permuted identically for both operands (MFMA step 2t+u, lane half h covers
//    k = 16(2t+h) + 8u .. +8),
//  * dedicated loader waves, NSTAGE-deep ring, one s_barrier per k-step, fp16 tile staged through LDS for the stores:
//    the structure of gemm.hip.
#include "common.h"
#include <type_traits>

namespace {

struct WoArgs {
    const uint16_t* x; const uint8_t* w; const uint16_t* sw; const uint16_t* bias; uint16_t* y;
    int M, N, K, ldx, ldy, tiles_m, tiles_n, wrows16;
};

template <int N> __device__ __forceinline__ void wo_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

__device__ __forceinline__ void wo_glds16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int wo_swz(int r, int c) { return c ^ ((r >> 2) & 3); }       // weights: 4 chunks per row
__device__ __forceinline__ int wo_swz8(int r, int c) { return c ^ ((r >> 1) & 7); }      // tokens: 8 chunks per row
__device__ __forceinline__ void wo_fence() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

// 4 offset-binary bytes -> 4 halves (two dwords)
__device__ __forceinline__ void cvt_u8x4(uint32_t d, uint32_t& o0, uint32_t& o1) {
    const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, d, 0x04010400u);      // (0x64, b1, 0x64, b0)
    const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, d, 0x04030402u);      // (0x64, b3, 0x64, b2)
    const h2_t off = {static_cast<_Float16>(1152.f), static_cast<_Float16>(1152.f)};
    o0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, p0) - off);
    o1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, p1) - off);
}

// ABL (tuning only): 0 normal, 2 no DMA, 3 MFMA only, 4 MFMA + conversions
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int LOADERS, int ABL = 0>
__global__ __launch_bounds__((WAVES_M * WAVES_N + LOADERS) * 64) void gemm_w8a16_kernel(const WoArgs a)
{
    constexpr int CW = WAVES_M * WAVES_N, NT = (CW + LOADERS) * 64;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
    constexpr int W_BYTES = BN * 64, X_BYTES = BM * 128, STAGE_BYTES = W_BYTES + X_BYTES;
    constexpr int WP = BN / 16, XP = BM / 8, TI = WP + XP;                      // 1-KiB DMA pieces per stage
    constexpr int LOADS = (TI + LOADERS - 1) / LOADERS;
    constexpr int LOOK = NSTAGE - 2, NEWER = LOOK - 1;
    constexpr int OPITCH = BN * 2 + 16;
    static_assert(BM % 32 == 0 && BN % 32 == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shapes");
    static_assert(LOOK >= 1 && LOADS * NEWER < 64, "vmcnt range");
    static_assert(BM * OPITCH <= NSTAGE * STAGE_BYTES, "output staging tile must fit in the ring");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int ntiles = a.tiles_m * a.tiles_n;
    int tile;
    {   // XCD-aware remap (block b runs on XCD b % 8): consecutive logical tiles share an L2
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, x = b & 7, s = b >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
    }
    const int tn = tile / a.tiles_m, tm = tile - tn * a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int nk = a.K / 64;

    // ================================================================================================ loader waves
    if (wave >= CW) {
        __builtin_amdgcn_s_setprio(2);
        const int iw = wave - CW;
        const uint8_t* nsrc[LOADS];
        int kstr[LOADS], loff[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            int pw = i * LOADERS + iw;
            if (pw >= TI) pw = iw;                                              // benign duplicate
            if (pw < WP) {                                                      // 16 weight rows: one contiguous KiB
                int rb = (n0 >> 4) + pw;
                rb = rb < (a.wrows16 >> 4) ? rb : (a.wrows16 >> 4) - 1;
                nsrc[i] = a.w + static_cast<size_t>(rb) * 1024 + lane * 16;
                kstr[i] = a.wrows16 * 64;
                loff[i] = pw * 1024;
            } else {                                                            // 8 token rows x 128 bytes (64 halves)
                const int xp = pw - WP, r = xp * 8 + (lane >> 3), pc = lane & 7;
                int row = m0 + r;
                row = row < a.M ? row : a.M - 1;
                nsrc[i] = reinterpret_cast<const uint8_t*>(a.x) + static_cast<size_t>(row) * a.ldx * 2 + wo_swz8(r, pc) * 16;
                kstr[i] = 128;
                loff[i] = W_BYTES + xp * 1024;
            }
        }
        auto stage = [&](int buf) {
            uint8_t* base = lds + buf * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < LOADS; ++i) { if constexpr (ABL == 0) wo_glds16(nsrc[i], base + loff[i]); nsrc[i] += kstr[i]; }
        };
#pragma unroll
        for (int s = 0; s < LOOK; ++s)
            if (s < nk) stage(s);
        if (NEWER < nk) wo_wait_vmcnt<LOADS * NEWER>(); else wo_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int nxt = LOOK % NSTAGE, kt = 0;
        for (; kt + LOOK < nk; ++kt) {
            stage(nxt);
            wo_wait_vmcnt<LOADS * NEWER>();
            __builtin_amdgcn_s_barrier();
            nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
        }
        for (; kt + 1 < nk; ++kt) {
            wo_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                                           // epilogue: ring free
        __builtin_amdgcn_s_barrier();                                           // epilogue: staging tile complete
    }

    // ================================================================================================ consumer waves
    const int wn = wave % WAVES_N, wm = (wave / WAVES_N) % WAVES_M;
    const int lr = lane & 31, lh = lane >> 5;
    f32x16 acc[NI][MI];
    if (wave < CW) {
        int wrow[NI], xrow[MI];
#pragma unroll
        for (int i = 0; i < NI; ++i) wrow[i] = wn * WN + i * 32 + lr;
#pragma unroll
        for (int j = 0; j < MI; ++j) xrow[j] = wm * WM + j * 32 + lr;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        u32x4 wraw[2][NI];                   // raw weight chunk t of the stage (16 k-bytes: MFMA steps 2t and 2t+1)
        u32x4 xf[2][MI];                     // token fragments, by MFMA step parity
        if constexpr (ABL != 0) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int i = 0; i < NI; ++i) { wraw[p][i] = u32x4{(uint32_t)lane, 1u, 2u, 3u}; asm volatile("" : "+v"(wraw[p][i])); }
#pragma unroll
                for (int j = 0; j < MI; ++j) { xf[p][j] = u32x4{(uint32_t)lane, 1u, 2u, 3u}; asm volatile("" : "+v"(xf[p][j])); }
            }
        }
        auto load_w = [&](auto t_c, int buf) {
            constexpr int T = decltype(t_c)::value;
            if constexpr (ABL >= 3) return;
            const uint8_t* wb = lds + buf * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                wraw[T][i] = *reinterpret_cast<const u32x4*>(wb + wrow[i] * 64 + wo_swz(wrow[i], 2 * T + lh) * 16);
        };
        auto load_x = [&](auto s_c, int buf) {
            constexpr int S = decltype(s_c)::value;                             // MFMA step 0..3 of the stage
            if constexpr (ABL >= 3) return;
            const uint8_t* xb = lds + buf * STAGE_BYTES + W_BYTES;
#pragma unroll
            for (int j = 0; j < MI; ++j)        // k = 16 (2t + lh) + 8u .. +8  ->  16-byte chunk 4t + 2 lh + u of the row
                xf[S & 1][j] = *reinterpret_cast<const u32x4*>(xb + xrow[j] * 128 + wo_swz8(xrow[j], 4 * (S >> 1) + 2 * lh + (S & 1)) * 16);
        };
        // The int8 -> fp16 conversion of the weight fragments of MFMA step S+1 is issued in the shadow of the MFMAs of
        // step S (one cvt_u8x4 = 4 VALU behind an MFMA: what a wave can co-issue for free, tools/ubench_mfma_valu.hip);
        // converted sets are double-buffered by step parity.
        f16x8 wc[2][NI];
        auto convert_unit = [&](auto sn_c, int u) {          // unit u of step SN: (weight block u / 2, dword u % 2)
            constexpr int SN = decltype(sn_c)::value, T = SN >> 1, U = SN & 1;
            const int i = u >> 1, d = u & 1;
            uint32_t o0, o1;
            cvt_u8x4(wraw[T][i][2 * U + d], o0, o1);
            u32x4 t = __builtin_bit_cast(u32x4, wc[SN & 1][i]);
            t[2 * d] = o0; t[2 * d + 1] = o1;
            wc[SN & 1][i] = __builtin_bit_cast(f16x8, t);
        };
        auto mma_step = [&](auto s_c, auto convert_next_c) {
            constexpr int S = decltype(s_c)::value, U = S & 1, SN = (S + 1) & 3;
            constexpr bool CONV = decltype(convert_next_c)::value;
            constexpr int NM = NI * MI, NU = 2 * NI;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int i = m / MI, j = m % MI;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[S & 1][i], __builtin_bit_cast(f16x8, xf[U][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (CONV && ABL != 3) {
                    // late slots first: the raw chunk of step S+1 may have been requested at the start of this step
                    constexpr int FIRST = NM > 2 ? 1 : 0;
                    if (m >= FIRST) {
#pragma unroll
                        for (int u = (NU * (m - FIRST)) / (NM - FIRST); u < (NU * (m - FIRST + 1)) / (NM - FIRST); ++u)
                            convert_unit(std::integral_constant<int, SN>{}, u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        using S3 = std::integral_constant<int, 3>;

        __builtin_amdgcn_s_barrier();                                           // stage 0 landed
        wo_fence();
        load_w(S0{}, 0); load_x(S0{}, 0);
#pragma unroll
        for (int u = 0; u < 2 * NI; ++u) convert_unit(S0{}, u);
        int cur = 0;
        // one stage; NEXT: a following stage exists (its barrier sits in front of MFMA step 3).  The last stage is
        // peeled so that the accumulators have a single definition inside the loop (no copies at a control-flow merge).
        auto stage_body = [&](auto next_c) {
            constexpr bool NEXT = decltype(next_c)::value;
            const int cur1 = (cur + 1 == NSTAGE) ? 0 : cur + 1;
            load_x(S1{}, cur); load_w(S1{}, cur);
            mma_step(S0{}, std::true_type{});
            wo_fence();
            load_x(S2{}, cur);
            mma_step(S1{}, std::true_type{});
            wo_fence();
            load_x(S3{}, cur);
            mma_step(S2{}, std::true_type{});
            wo_fence();
            if constexpr (NEXT) {
                __builtin_amdgcn_s_barrier();                                   // stage kt+1 landed and visible
                wo_fence();
                load_w(S0{}, cur1); load_x(S0{}, cur1);
                mma_step(S3{}, std::true_type{});
            } else {
                mma_step(S3{}, std::false_type{});
            }
            wo_fence();
            cur = cur1;
        };
        for (int kt = 0; kt + 1 < nk; ++kt) stage_body(std::true_type{});
        stage_body(std::false_type{});

        // ---- epilogue: y = acc * scale_col[n] (+ bias[n]) -> fp16 tile in LDS ---------------------------------------
        __builtin_amdgcn_s_barrier();                                           // every wave is done reading the ring
        wo_fence();
        const bool has_bias = a.bias != nullptr;
        auto unpack4 = [](u32x2 v, float* o) {
            o[0] = h2f(static_cast<uint16_t>(v.x & 0xffffu)); o[1] = h2f(static_cast<uint16_t>(v.x >> 16));
            o[2] = h2f(static_cast<uint16_t>(v.y & 0xffffu)); o[3] = h2f(static_cast<uint16_t>(v.y >> 16));
        };
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nloc = wn * WN + i * 32 + 4 * lh;
            float swv[16], bv[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + nloc + 8 * g;
                const int nc = n < a.N ? n : a.N - 4;                           // N % 4 == 0: groups are all in or all out
                unpack4(*reinterpret_cast<const u32x2_u*>(a.sw + nc), swv + 4 * g);
                if (has_bias) unpack4(*reinterpret_cast<const u32x2_u*>(a.bias + nc), bv + 4 * g);
            }
#pragma unroll
            for (int j = 0; j < MI; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][4 * g + e] * swv[4 * g + e];
                        if (has_bias) v[e] += bv[4 * g + e];
                    }
                    u32x2 o;
                    o.x = static_cast<uint32_t>(f2h(v[0])) | (static_cast<uint32_t>(f2h(v[1])) << 16);
                    o.y = static_cast<uint32_t>(f2h(v[2])) | (static_cast<uint32_t>(f2h(v[3])) << 16);
                    *reinterpret_cast<u32x2*>(lds + xrow[j] * OPITCH + (nloc + 8 * g) * 2) = o;
                }
            }
        }
        // ds_write is asynchronous and a raw s_barrier does not wait for it: without this the last tile writes of a
        // wave can still be queued when another wave's copy-out read of the same bytes is served (seen on MI355X as
        // a stale 2-row x 16-column patch, a few launches in a hundred)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                           // staging tile complete
        wo_fence();
    }
    // all waves: 8-byte segments (N % 4 == 0 is the only alignment the ABI asks of N and ldy)
    constexpr int CPR = BN * 2 / 8;
    for (int q = tid; q < BM * CPR; q += NT) {
        const int r = q / CPR, c = q - r * CPR;
        const int m = m0 + r, n = n0 + c * 4;
        if (m < a.M && n < a.N) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(lds + r * OPITCH + c * 8);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(a.y + static_cast<size_t>(m) * a.ldy + n));
        }
    }
}

// ---- small-batch (decode) form: M <= 32 ----------------------------------------------------------------------------
// A weight stream, like gemm_skinny.hip: one workgroup per 32 output channels, its 8 waves split K, fragments go
// global -> VGPR -> MFMA (weights: the packed offset-binary image; activations: the caller's fp16 rows, L2-resident), no LDS
// staging, no barrier in the k loop.  The 8 fp32 partial tiles are summed through LDS in wave order (a fixed order: results
// are reproducible run to run), wave 0 applies scale_col and bias.
constexpr int WSK = 8, WUN = 2;
__global__ __launch_bounds__(WSK * 64, 2) void gemm_w8a16_skinny_kernel(const WoArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[WSK * 4096];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = wave_id_uniform();
    const int n0 = blockIdx.x * 32;
    const int nk = a.K / 64;
    const int k_lo = (nk * wave) / WSK, k_hi = (nk * (wave + 1)) / WSK;
    int wr = n0 + lr; wr = wr < a.N ? wr : a.N - 1;
    const int xr = lr < a.M ? lr : a.M - 1;
    const uint8_t* wp = a.w + static_cast<size_t>(wr >> 4) * 1024 + (wr & 15) * 64;
    const uint8_t* xp = reinterpret_cast<const uint8_t*>(a.x) + static_cast<size_t>(xr) * a.ldx * 2 + lh * 32;
    const int wc0 = wo_swz(wr, lh) * 16, wc1 = wo_swz(wr, 2 + lh) * 16;      // chunk 2t + lh: MFMA steps 2t and 2t + 1
    const size_t wks = static_cast<size_t>(a.wrows16) * 64;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    u32x4 wf[2][WUN][2], xf[2][WUN][4];
    auto load_group = [&](auto p_c, int kb) {
        constexpr int P = decltype(p_c)::value;
#pragma unroll
        for (int u = 0; u < WUN; ++u) {
            int k = kb + u; k = k < k_hi ? k : k_hi - 1;
            const uint8_t* w = wp + k * wks;
            const uint8_t* x = xp + static_cast<size_t>(k) * 128;        // k = 16 (2t + lh) + 8u: byte 64 t + 32 lh + 16 u
            wf[P][u][0] = *reinterpret_cast<const u32x4*>(w + wc0);
            wf[P][u][1] = *reinterpret_cast<const u32x4*>(w + wc1);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                xf[P][u][s4] = *reinterpret_cast<const u32x4*>(x + (s4 >> 1) * 64 + (s4 & 1) * 16);
        }
    };
    auto mma_group = [&](auto p_c, int kb) {
        constexpr int P = decltype(p_c)::value;
#pragma unroll
        for (int u = 0; u < WUN; ++u) {
            if (kb + u < k_hi) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    uint32_t o0, o1, o2, o3;
                    cvt_u8x4(wf[P][u][s4 >> 1][2 * (s4 & 1)], o0, o1);
                    cvt_u8x4(wf[P][u][s4 >> 1][2 * (s4 & 1) + 1], o2, o3);
                    const f16x8 wc = __builtin_bit_cast(f16x8, u32x4{o0, o1, o2, o3});
                    if (s4 & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc, __builtin_bit_cast(f16x8, xf[P][u][s4]), acc1, 0, 0, 0);
                    else        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc, __builtin_bit_cast(f16x8, xf[P][u][s4]), acc0, 0, 0, 0);
                }
            }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if (k_lo < k_hi) {
        load_group(P0{}, k_lo);
        for (int kb = k_lo; kb < k_hi; kb += 2 * WUN) {
            if (kb + WUN < k_hi) load_group(P1{}, kb + WUN);
            mma_group(P0{}, kb);
            if (kb + WUN < k_hi) {
                if (kb + 2 * WUN < k_hi) load_group(P0{}, kb + 2 * WUN);
                mma_group(P1{}, kb + WUN);
            }
        }
    }
    f32x16 acc = acc0 + acc1;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(lds + (wave * 4 + g) * 1024 + lane * 16) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 s = *reinterpret_cast<const f32x4*>(lds + g * 1024 + lane * 16);
#pragma unroll
        for (int w = 1; w < WSK; ++w) s += *reinterpret_cast<const f32x4*>(lds + (w * 4 + g) * 1024 + lane * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = s[e];
    }
    auto unpack4 = [](u32x2 v, float* o) {
        o[0] = h2f(static_cast<uint16_t>(v.x & 0xffffu)); o[1] = h2f(static_cast<uint16_t>(v.x >> 16));
        o[2] = h2f(static_cast<uint16_t>(v.y & 0xffffu)); o[3] = h2f(static_cast<uint16_t>(v.y >> 16));
    };
    const int m = lr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + 4 * lh + 8 * g;
        const int nc = n < a.N ? n : a.N - 4;
        float sv[4], bv[4] = {0.f, 0.f, 0.f, 0.f};
        unpack4(*reinterpret_cast<const u32x2_u*>(a.sw + nc), sv);
        if (a.bias) unpack4(*reinterpret_cast<const u32x2_u*>(a.bias + nc), bv);
        if (m < a.M && n < a.N) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[4 * g + e] * sv[e]; if (a.bias) v[e] += bv[e]; }
            u32x2 o;
            o.x = static_cast<uint32_t>(f2h(v[0])) | (static_cast<uint32_t>(f2h(v[1])) << 16);
            o.y = static_cast<uint32_t>(f2h(v[2])) | (static_cast<uint32_t>(f2h(v[3])) << 16);
            *reinterpret_cast<u32x2_u*>(a.y + static_cast<size_t>(m) * a.ldy + n) = o;
        }
    }
}

// One-time re-tiling of the checkpoint's [K,N] int8 matrix into offset-binary P16x64 (rows = output channels).
__global__ __launch_bounds__(256) void pack_w8a16_kernel(const int8_t* __restrict__ qkn, uint8_t* __restrict__ dst, int K, int N, int rows16)
{
    const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    const long long total = static_cast<long long>(rows16) * (K >> 4);
    if (t >= total) return;
    const int per_kb = rows16 * 4;
    const int kb = static_cast<int>(t / per_kb), rem = static_cast<int>(t % per_kb);
    const int rb = rem >> 6, r = (rem >> 2) & 15, pc = rem & 3;
    const int n = rb * 16 + r, c = pc ^ ((r >> 2) & 3);
    const int k0 = kb * 64 + c * 16;
    uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};      // q = 0 for padding rows
    if (n < N) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t v = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = qkn[static_cast<size_t>(k0 + 4 * d + e) * N + n];
                v |= static_cast<uint32_t>((q + 128) & 0xff) << (8 * e);
            }
            w[d] = v;
        }
    }
    reinterpret_cast<uint4*>(dst)[t] = make_uint4(w[0], w[1], w[2], w[3]);
}

struct WoConfig {
    const char* name; int bm, bn, waves, nstage;
    void (*k)(const WoArgs);
};
const WoConfig g_wo[] = {
    {"w8a16_128x192_w2x2_s5_l4", 128, 192, 8, 5, gemm_w8a16_kernel<128, 192, 2, 2, 5, 4>},
    {"w8a16_128x128_w2x2_s4_l4", 128, 128, 8, 4, gemm_w8a16_kernel<128, 128, 2, 2, 4, 4>},
    {"w8a16_64x128_w2x2_s4_l4",  64, 128, 8, 4, gemm_w8a16_kernel<64, 128, 2, 2, 4, 4>},
    {"w8a16_128x192_abl2", 128, 192, 8, 4, gemm_w8a16_kernel<128, 192, 2, 2, 4, 4, 2>},     // tuning: no DMA
    {"w8a16_128x192_abl3", 128, 192, 8, 4, gemm_w8a16_kernel<128, 192, 2, 2, 4, 4, 3>},     // tuning: MFMA only
    {"w8a16_128x192_abl4", 128, 192, 8, 4, gemm_w8a16_kernel<128, 192, 2, 2, 4, 4, 4>},     // tuning: MFMA + conversions
};
constexpr int NUM_WO_PICK = 3;
constexpr int NUM_WO = sizeof(g_wo) / sizeof(g_wo[0]);
int g_wo_forced = -1;

inline int wo_cdiv(int a, int b) { return (a + b - 1) / b; }

int pick_wo(int M, int N) {
    double best = 1e30; int bi = 0;
    for (int c = 0; c < NUM_WO_PICK; ++c) {
        const int tiles = wo_cdiv(M, g_wo[c].bm) * wo_cdiv(N, g_wo[c].bn);
        // time ~ rounds x tile cost: MFMA work grows with bm*bn, the weight stream of a tile with bn
        const double t = wo_cdiv(tiles, 256) * (static_cast<double>(g_wo[c].bm) * g_wo[c].bn + 4096.0 * g_wo[c].bn / 64.0);
        if (t < best * 0.999) { best = t; bi = c; }
    }
    return bi;
}

}  // namespace

extern "C" int mixq_pack_w8a16(const int8_t* q_weight_kn, uint8_t* packed, int K, int N, mixq_stream_t stream)
{
    if (K <= 0 || N <= 0 || !q_weight_kn || !packed) return MIXQ_EINVAL;
    if (K % 64) return MIXQ_ESHAPE;
    const int rows16 = (N + 15) & ~15;
    const long long total = static_cast<long long>(rows16) * (K >> 4);
    hipLaunchKernelGGL(pack_w8a16_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, mixq_stream(stream),
                       q_weight_kn, packed, K, N, rows16);
    return mixq_launch_status();
}

extern "C" int mixq_gemm_w8a16(const uint16_t* x, int ldx, const uint8_t* w_packed, const uint16_t* scale_col, const uint16_t* bias,
                               uint16_t* y, int ldy, int M, int N, int K, mixq_stream_t stream)
{
    if (M < 0 || N < 0 || K <= 0 || (M > 0 && N > 0 && (!x || !w_packed || !scale_col || !y))) return MIXQ_EINVAL;
    if ((K % 64) || (N & 3) || (ldy & 3) || ldy < N || ldx < K || (ldx & 7)) return MIXQ_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 7)) return MIXQ_EINVAL;
    if (M == 0 || N == 0) return MIXQ_OK;
    WoArgs a;
    a.x = x; a.w = w_packed; a.sw = scale_col; a.bias = bias; a.y = y;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldy = ldy; a.tiles_m = a.tiles_n = 0; a.wrows16 = (N + 15) & ~15;
    if ((g_wo_forced < 0 && M <= 32) || g_wo_forced == NUM_WO) {                 // small batch: the weight-stream form
        if (M > 32) return MIXQ_EINVAL;
        hipLaunchKernelGGL(gemm_w8a16_skinny_kernel, dim3((N + 31) / 32), dim3(WSK * 64), 0, mixq_stream(stream), a);
        return mixq_launch_status();
    }
    const int c = (g_wo_forced >= 0 && g_wo_forced < NUM_WO) ? g_wo_forced : pick_wo(M, N);
    const WoConfig& g = g_wo[c];
    a.tiles_m = wo_cdiv(M, g.bm); a.tiles_n = wo_cdiv(N, g.bn);
    const size_t shm = static_cast<size_t>(g.bn * 64 + g.bm * 128) * g.nstage;
    if (int rc = mixq_ensure_dynamic_lds(reinterpret_cast<const void*>(g.k), shm)) return rc;
    hipLaunchKernelGGL(g.k, dim3(a.tiles_m * a.tiles_n), dim3(g.waves * 64), shm, mixq_stream(stream), a);
    return mixq_launch_status();
}

extern "C" int mixq_gemm_w8a16_set_config(int cfg) {
    if (cfg < -1 || cfg > NUM_WO) return MIXQ_EINVAL;          // NUM_WO = the small-batch kernel
    g_wo_forced = cfg;
    return MIXQ_OK;
}
