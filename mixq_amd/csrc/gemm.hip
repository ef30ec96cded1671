// (iii)+(iv): int8 x int8 -> int32 MFMA GEMM for gfx950 with the MixQ epilogue fused in.
//
//   Y[m,n] = fp16( (sum_k Xq[m,k] Wq[n,k]) * sx[m] * sw[n]  +  sum_j Xo[m,j] Wo[n,j]  + addend[m,n] ) -> act -> + bias[n]
//
// Design (MI355X-first):
//  * both operands are K-contiguous ("TN"); tiles are streamed HBM/L2 -> LDS with 16-byte LDS-DMA
//    (global_load_lds_dwordx4, no VGPR round trip) into an NSTAGE-deep ring; ONE raw s_barrier per k-step and a
//    counted s_waitcnt vmcnt(N) keep NSTAGE-2 k-steps of loads in flight across the barrier.
//  * LDS rows are BKB bytes; the 16-byte chunk index is XOR-swizzled with row bits so the ds_read_b128 fragment
//    reads of a 16-lane group hit 16 distinct 16-byte bank slots (swizzle applied on the DMA *source* address and
//    on the read address, the LDS image itself is lane-linear as LDS-DMA requires).
//  * v_mfma_i32_32x32x32_i8 with the WEIGHT rows as the A operand and the ACTIVATION rows as the B operand, so a
//    lane ends up holding 4 consecutive output columns n of one token m: the epilogue needs one x_scale per lane,
//    and stores 8-byte (4 x fp16) pieces of a Y row.
//  * the outlier correction runs as fp16 MFMA tail iterations (v_mfma_f32_32x32x16_f16) on the SAME accumulator
//    registers after they were dequantised to fp32 in place - no M x N side matrix is ever materialised
//    (the reference writes and re-reads one: linear.py:248-256).  The outlier count may live in device memory.
//  * W4A4: CDNA4 has no int4 MFMA; nibbles are expanded in registers to int8 values 16*v (a shift and a mask per
//    dword), the factor 256 is removed exactly in the fp32 epilogue.
//  * the grid is 1-D over output tiles with an XCD-aware remap: the tiles that share a weight panel run on the
//    same XCD (same L2).  The tile shape is picked per problem so the tile count fills the 256 CUs.
#include "common.h"
#include <stdio.h>
#include <string.h>

namespace {

struct GemmArgs {
    const uint8_t* qx;  const uint8_t* qw;           // [M, KB], [N, KB] bytes (KB = K for int8, K/2 for int4)
    const uint16_t* sx; const uint16_t* sw;           // fp16 [M], [N]
    const uint16_t* xo; const uint16_t* wo;           // fp16 [M,ldxo], [N,ldwo] outlier operands (may be null)
    const int32_t* n_out_dev;
    const uint16_t* addend; const uint16_t* bias;
    uint16_t* y; int32_t* y32;
    int M, N, KB;                                     // KB: bytes per operand row
    int ldxo, ldwo, n_out, lda, ldy;
    int act;
    int tiles_m, tiles_n;
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

__device__ __forceinline__ void glds16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// physical 16-byte chunk of logical chunk c in row r (involution in c for fixed r)
template <int BKB> __device__ __forceinline__ int swz(int r, int c) {
    if constexpr (BKB == 64) return c ^ ((r >> 2) & 3);
    else                     return c ^ ((r >> 1) & 7);      // BKB == 128
}

__device__ __forceinline__ float silu(float v) { return v / (1.f + __expf(-v)); }

// BM: activation rows per tile, BN: weight rows per tile, BKB: bytes of K per stage and row, WAVES_M x WAVES_N waves,
// NSTAGE ring depth.  MODE 0: int8 fused epilogue, 1: int4 fused epilogue, 2: int8 raw int32 output.
template <int BM, int BN, int BKB, int WAVES_M, int WAVES_N, int NSTAGE, int MODE>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_kernel(const GemmArgs a)
{
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int CH = BKB / 16;                       // 16-byte chunks per row
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int TI = (BM + BN) * CH / 64;              // 1-KiB wave-instructions per stage (weight rows, then activation rows)
    constexpr int LOADS_LO = TI / NW, EXTRA = TI % NW;   // waves < EXTRA issue one more
    constexpr int LOADS_HI = LOADS_LO + (EXTRA ? 1 : 0);
    constexpr int STAGE_BYTES = (BM + BN) * BKB;
    static_assert((BM + BN) * CH % 64 == 0, "stage must be a whole number of 1-KiB DMA pieces");
    static_assert(BM % 16 == 0 && BN % 16 == 0, "swizzle phase must agree between the two operand regions");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
    static_assert(LOADS_HI * (NSTAGE - 2) < 64, "vmcnt range");
    constexpr bool I4 = (MODE == 1);

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    // ---- tile id: XCD-aware remap (block b runs on XCD b % 8) so consecutive logical tiles share an L2 -----
    const int ntiles = a.tiles_m * a.tiles_n;
    int tile;
    {
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, x = b & 7, s = b >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;          // bijective for any ntiles
    }
    const int tn = tile / a.tiles_m, tm = tile - tn * a.tiles_m;               // m fastest: weight panel shared
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;

    // ---- per-thread DMA source pointers (row clamped so ragged tiles never read out of bounds) -------------
    const uint8_t* src[LOADS_HI];
#pragma unroll
    for (int i = 0; i < LOADS_HI; ++i) {
        int qd = (i * NW + wave) * 64 + lane;
        if (qd >= (BM + BN) * CH) qd = (BM + BN) * CH - 1;             // only for waves >= EXTRA in the last slot (never issued)
        const int rr = qd / CH, pc = qd % CH;
        if (rr < BN) {
            int gr = n0 + rr; gr = gr < a.N ? gr : a.N - 1;
            src[i] = a.qw + static_cast<size_t>(gr) * a.KB + swz<BKB>(rr, pc) * 16;
        } else {
            const int r = rr - BN;
            int gr = m0 + r; gr = gr < a.M ? gr : a.M - 1;
            src[i] = a.qx + static_cast<size_t>(gr) * a.KB + swz<BKB>(r, pc) * 16;
        }
    }
    const bool extra_wave = (EXTRA > 0) && (wave < EXTRA);
    const int nk = a.KB / BKB;

    auto stage = [&](int buf, int kt) {
        uint8_t* base = lds + buf * STAGE_BYTES + wave * 1024;
        const int koff = kt * BKB;
#pragma unroll
        for (int i = 0; i < LOADS_LO; ++i) glds16(src[i] + koff, base + i * NW * 1024);
        if constexpr (EXTRA > 0) {
            if (extra_wave) glds16(src[LOADS_LO] + koff, base + LOADS_LO * NW * 1024);
        }
    };

    i32x16 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // fragment row offsets inside a stage (bytes), constant over k
    const int lr = lane & 31, lh = lane >> 5;
    int wrow[NI], xrow[MI];
#pragma unroll
    for (int i = 0; i < NI; ++i) wrow[i] = wn * WN + i * 32 + lr;
#pragma unroll
    for (int j = 0; j < MI; ++j) xrow[j] = wm * WM + j * 32 + lr;

    auto compute = [&](int buf) {
        const uint8_t* wb = lds + buf * STAGE_BYTES;
        const uint8_t* xb = wb + BN * BKB;
#pragma unroll
        for (int s = 0; s < BKB / 32; ++s) {
            i32x4 wf[NI], xf[MI];
#pragma unroll
            for (int i = 0; i < NI; ++i)
                wf[i] = *reinterpret_cast<const i32x4*>(wb + wrow[i] * BKB + swz<BKB>(wrow[i], s * 2 + lh) * 16);
#pragma unroll
            for (int j = 0; j < MI; ++j)
                xf[j] = *reinterpret_cast<const i32x4*>(xb + xrow[j] * BKB + swz<BKB>(xrow[j], s * 2 + lh) * 16);
            if constexpr (!I4) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], acc[i][j], 0, 0, 0);
            } else {
                // 16 packed bytes = 32 int4: even columns (low nibbles) and odd columns (high nibbles) become two
                // int8 fragments holding 16*v; A and B use the same split so the k pairing is preserved.
                i32x4 wl[NI], wh[NI], xl[MI], xh[MI];
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t v = static_cast<uint32_t>(wf[i][d]);
                        wl[i][d] = static_cast<int>((v << 4) & 0xf0f0f0f0u);
                        wh[i][d] = static_cast<int>(v & 0xf0f0f0f0u);
                    }
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t v = static_cast<uint32_t>(xf[j][d]);
                        xl[j][d] = static_cast<int>((v << 4) & 0xf0f0f0f0u);
                        xh[j][d] = static_cast<int>(v & 0xf0f0f0f0u);
                    }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wl[i], xl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wh[i], xh[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    };

    // ---- main loop: NSTAGE-1 stages in flight, one barrier per k-step --------------------------------------
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage(s, s);
    for (int kt = 0; kt < nk; ++kt) {
        // loads issued after stage kt: min(NSTAGE-2, nk-1-kt) stages
        if (kt + NSTAGE - 2 < nk) {
            if constexpr (EXTRA > 0) {
                if (extra_wave) wait_vmcnt<LOADS_HI * (NSTAGE - 2)>();
                else            wait_vmcnt<LOADS_LO * (NSTAGE - 2)>();
            } else {
                wait_vmcnt<LOADS_LO * (NSTAGE - 2)>();
            }
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (kt + NSTAGE - 1 < nk) stage((kt + NSTAGE - 1) % NSTAGE, kt + NSTAGE - 1);
        compute(kt % NSTAGE);
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int m = m0 + xrow[j];
                const int nb = n0 + wn * WN + i * 32 + 4 * lh;
                if (m < a.M) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nb + 8 * g;
                        if (n + 3 < a.N) {
                            i32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            *reinterpret_cast<i32x4*>(a.y32 + static_cast<size_t>(m) * a.ldy + n) = v;
                        } else {
                            for (int e = 0; e < 4; ++e)
                                if (n + e < a.N) a.y32[static_cast<size_t>(m) * a.ldy + n + e] = acc[i][j][4 * g + e];
                        }
                    }
                }
            }
        return;
    } else {
        int n_out = a.n_out;
        if (a.n_out_dev) { const int nd = *a.n_out_dev; n_out = nd < n_out ? nd : n_out; }
        if (!a.xo || !a.wo) n_out = 0;
        const int ksteps = (n_out + 15) >> 4;
        constexpr float PRE = I4 ? (1.f / 256.f) : 1.f;

        float sxv[MI];
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int m = m0 + xrow[j];
            sxv[j] = (m < a.M) ? h2f(a.sx[m]) * PRE : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nb = n0 + wn * WN + i * 32 + 4 * lh;
            // 16 weight scales this lane needs: n = nb + 8g + e
            float swv[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + 8 * g;
                if (n + 3 < a.N) {
                    const u32x2 p = *reinterpret_cast<const u32x2*>(a.sw + n);
                    swv[4 * g]     = h2f(static_cast<uint16_t>(p.x & 0xffffu));
                    swv[4 * g + 1] = h2f(static_cast<uint16_t>(p.x >> 16));
                    swv[4 * g + 2] = h2f(static_cast<uint16_t>(p.y & 0xffffu));
                    swv[4 * g + 3] = h2f(static_cast<uint16_t>(p.y >> 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) swv[4 * g + e] = (n + e < a.N) ? h2f(a.sw[n + e]) : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                f32x16 f;
#pragma unroll
                for (int r = 0; r < 16; ++r) f[r] = static_cast<float>(acc[i][j][r]) * sxv[j] * swv[r];

                // fp16 outlier tail on the same accumulator registers
                if (ksteps > 0) {
                    int wr = n0 + wrow[i]; wr = wr < a.N ? wr : a.N - 1;
                    int xr = m0 + xrow[j]; xr = xr < a.M ? xr : a.M - 1;
                    const uint16_t* wp = a.wo + static_cast<size_t>(wr) * a.ldwo + lh * 8;
                    const uint16_t* xp = a.xo + static_cast<size_t>(xr) * a.ldxo + lh * 8;
                    for (int kk = 0; kk < ksteps; ++kk) {
                        u32x4 wq = *reinterpret_cast<const u32x4*>(wp + kk * 16);
                        u32x4 xq = *reinterpret_cast<const u32x4*>(xp + kk * 16);
                        const int kb = kk * 16 + lh * 8;               // mask columns >= n_out (pad may hold anything)
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

                const int m = m0 + xrow[j];
                if (m < a.M) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nb + 8 * g;
                        float v[4] = {f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]};
                        const bool full = (n + 3 < a.N);
                        if (a.addend) {
                            const uint16_t* ap = a.addend + static_cast<size_t>(m) * a.lda + n;
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (full || n + e < a.N) v[e] += h2f(ap[e]);
                        }
                        if (a.act == MIXQ_ACT_SILU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = silu(v[e]);
                        }
                        if (a.bias) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (full || n + e < a.N) v[e] += h2f(a.bias[n + e]);
                        }
                        uint16_t* yp = a.y + static_cast<size_t>(m) * a.ldy + n;
                        if (full) {
                            u32x2 o;
                            o.x = static_cast<uint32_t>(f2h(v[0])) | (static_cast<uint32_t>(f2h(v[1])) << 16);
                            o.y = static_cast<uint32_t>(f2h(v[2])) | (static_cast<uint32_t>(f2h(v[3])) << 16);
                            *reinterpret_cast<u32x2*>(yp) = o;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (n + e < a.N) yp[e] = f2h(v[e]);
                        }
                    }
                }
            }
        }
    }
}

// Standalone epilogue for the unfused debug pair (mixlib.dequantizeInt8).
__global__ __launch_bounds__(256) void dequant_kernel(const int32_t* __restrict__ y32, int ldy32, const uint16_t* __restrict__ sx,
                                                      const uint16_t* __restrict__ sw, const uint16_t* __restrict__ addend, int lda,
                                                      const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int ldy,
                                                      int M, int N, int act)
{
    const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= static_cast<long long>(M) * N) return;
    const int m = static_cast<int>(t / N), n = static_cast<int>(t % N);
    float v = static_cast<float>(y32[static_cast<size_t>(m) * ldy32 + n]) * h2f(sx[m]) * h2f(sw[n]);
    if (addend) v += h2f(addend[static_cast<size_t>(m) * lda + n]);
    if (act == MIXQ_ACT_SILU) v = silu(v);
    if (bias) v += h2f(bias[n]);
    y[static_cast<size_t>(m) * ldy + n] = f2h(v);
}

// ---- configuration table ---------------------------------------------------------------------------------------
struct GemmConfig {
    const char* name;
    int bm, bn, bkb, waves, nstage;
    void (*k8)(const GemmArgs);
    void (*k4)(const GemmArgs);
    void (*k32)(const GemmArgs);
};

#define MIXQ_CFG(BM, BN, BKB, WMv, WNv, NS)                                                                   \
    { #BM "x" #BN "x" #BKB "_w" #WMv "x" #WNv "_s" #NS, BM, BN, BKB, (WMv) * (WNv), NS,                     \
      gemm_kernel<BM, BN, BKB, WMv, WNv, NS, 0>, gemm_kernel<BM, BN, BKB, WMv, WNv, NS, 1>,                 \
      gemm_kernel<BM, BN, BKB, WMv, WNv, NS, 2> }

const GemmConfig g_cfgs[] = {
    MIXQ_CFG(256, 128, 64, 4, 2, 3),     // 0: 8 waves, 64x64 wave tile
    MIXQ_CFG(256, 128, 64, 2, 2, 3),     // 1: 4 waves, 128x64 wave tile
    MIXQ_CFG(256, 128, 128, 4, 2, 3),    // 2
    MIXQ_CFG(128, 128, 64, 2, 2, 3),     // 3
    MIXQ_CFG(128, 128, 64, 2, 2, 4),     // 4
    MIXQ_CFG(128, 128, 128, 2, 2, 3),    // 5
    MIXQ_CFG(256, 96, 64, 4, 1, 3),      // 6: 4 waves 64(M) x 96(N)
    MIXQ_CFG(256, 96, 64, 8, 1, 3),      // 7: 8 waves 32 x 96
    MIXQ_CFG(128, 192, 64, 2, 2, 3),     // 8: 4 waves 64 x 96
    MIXQ_CFG(128, 192, 64, 4, 2, 3),     // 9: 8 waves 32 x 96
    MIXQ_CFG(128, 64, 64, 2, 2, 4),      // 10
    MIXQ_CFG(64, 64, 64, 2, 2, 4),       // 11
    MIXQ_CFG(32, 128, 64, 1, 4, 4),      // 12: small-M
    MIXQ_CFG(256, 256, 64, 4, 2, 3),     // 13: 8 waves 64(M) x 128(N)
    MIXQ_CFG(256, 128, 64, 4, 2, 4),     // 14
    MIXQ_CFG(256, 128, 128, 2, 2, 3),    // 15
};
constexpr int NUM_CFGS = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

int g_forced_cfg = -1;
bool g_attr_done[NUM_CFGS][3];

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Shape-aware choice: estimated time ~ rounds over 256 CUs x per-tile work (MFMA-bound) with a small penalty for
// narrow tiles (less operand reuse -> more L2 traffic) and for padding waste.
int pick_config(int M, int N, int KB) {
    double best = 1e30; int bi = 0;
    for (int c = 0; c < NUM_CFGS; ++c) {
        const GemmConfig& g = g_cfgs[c];
        if (KB % g.bkb) continue;
        const int tiles = cdiv(M, g.bm) * cdiv(N, g.bn);
        const int rounds = cdiv(tiles, 256);
        const double tile_work = static_cast<double>(g.bm) * g.bn;                 // MFMA time per k-step
        const double feed = 4096.0 * (1.0 / g.bm + 1.0 / g.bn);                    // L2->LDS bytes/clk at peak
        const double eff = feed > 40.0 ? 40.0 / feed : 1.0;
        const double t = rounds * tile_work / eff;
        if (t < best * 0.999) { best = t; bi = c; }
    }
    return bi;
}

int launch_gemm(GemmArgs& a, int mode, hipStream_t st) {
    int c = g_forced_cfg >= 0 ? g_forced_cfg : pick_config(a.M, a.N, a.KB);
    const GemmConfig* g = &g_cfgs[c];
    if (a.KB % g->bkb) {                                     // forced config incompatible with K: fall back to auto
        c = pick_config(a.M, a.N, a.KB);
        g = &g_cfgs[c];
        if (a.KB % g->bkb) return MIXQ_ESHAPE;
    }
    a.tiles_m = cdiv(a.M, g->bm);
    a.tiles_n = cdiv(a.N, g->bn);
    void (*k)(const GemmArgs) = mode == 0 ? g->k8 : (mode == 1 ? g->k4 : g->k32);
    const size_t shm = static_cast<size_t>(g->bm + g->bn) * g->bkb * g->nstage;
    if (!g_attr_done[c][mode]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(shm));
        if (e != hipSuccess) return static_cast<int>(e);
        g_attr_done[c][mode] = true;
    }
    hipLaunchKernelGGL(k, dim3(a.tiles_m * a.tiles_n), dim3(g->waves * 64), shm, st, a);
    return mixq_launch_status();
}

int gemm_fused_common(const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                      const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                      const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int K,
                      int act, int bit, mixq_stream_t stream)
{
    if (!q_x || !q_w || !x_scale || !scale_col || !y || M < 0 || N < 0 || K <= 0 || n_out < 0) return MIXQ_EINVAL;
    if (act != MIXQ_ACT_NONE && act != MIXQ_ACT_SILU) return MIXQ_EINVAL;
    const int KB = bit == 8 ? K : K / 2;
    if ((KB % 64) || (bit == 4 && (K & 1)) || (N & 3) || (ldy & 3) || ldy < N) return MIXQ_ESHAPE;
    if (addend && lda != 0 && lda < N) return MIXQ_EINVAL;
    if (n_out > 0 && x_out && w_out) {
        const int need = (n_out + 15) & ~15;
        if (ldxo < need || ldwo < need || (ldxo & 7) || (ldwo & 7)) return MIXQ_ESHAPE;
    }
    if (M == 0 || N == 0) return MIXQ_OK;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.qx = static_cast<const uint8_t*>(q_x); a.qw = static_cast<const uint8_t*>(q_w);
    a.sx = x_scale; a.sw = scale_col; a.xo = x_out; a.wo = w_out; a.n_out_dev = n_out_dev;
    a.addend = addend; a.bias = bias; a.y = y;
    a.M = M; a.N = N; a.KB = KB; a.ldxo = ldxo; a.ldwo = ldwo; a.n_out = n_out; a.lda = lda; a.ldy = ldy; a.act = act;
    return launch_gemm(a, bit == 8 ? 0 : 1, mixq_stream(stream));
}

}  // namespace

extern "C" int mixq_gemm_i8_fused(const int8_t* q_x, const int8_t* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                                  const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out,
                                  const int32_t* n_out_dev, const uint16_t* addend, int lda, const uint16_t* bias,
                                  uint16_t* y, int ldy, int M, int N, int K, int act, mixq_stream_t stream)
{
    return gemm_fused_common(q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                             ldy, M, N, K, act, 8, stream);
}

extern "C" int mixq_gemm_i4_fused(const uint8_t* q_x, const uint8_t* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                                  const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out,
                                  const int32_t* n_out_dev, const uint16_t* addend, int lda, const uint16_t* bias,
                                  uint16_t* y, int ldy, int M, int N, int K, int act, mixq_stream_t stream)
{
    return gemm_fused_common(q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                             ldy, M, N, K, act, 4, stream);
}

extern "C" int mixq_gemm_i8(const int8_t* q_x, const int8_t* q_w, int32_t* y32, int ldy, int M, int N, int K,
                            mixq_stream_t stream)
{
    if (!q_x || !q_w || !y32 || M < 0 || N < 0 || K <= 0) return MIXQ_EINVAL;
    if ((K % 64) || (N & 3) || (ldy & 3) || ldy < N) return MIXQ_ESHAPE;
    if (M == 0 || N == 0) return MIXQ_OK;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.qx = reinterpret_cast<const uint8_t*>(q_x); a.qw = reinterpret_cast<const uint8_t*>(q_w);
    a.y32 = y32; a.M = M; a.N = N; a.KB = K; a.ldy = ldy;
    return launch_gemm(a, 2, mixq_stream(stream));
}

extern "C" int mixq_dequant(const int32_t* y32, int ldy32, const uint16_t* x_scale, const uint16_t* scale_col,
                            const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int act,
                            mixq_stream_t stream)
{
    if (!y32 || !x_scale || !scale_col || !y || M < 0 || N < 0 || ldy32 < N || ldy < N) return MIXQ_EINVAL;
    if (act != MIXQ_ACT_NONE && act != MIXQ_ACT_SILU) return MIXQ_EINVAL;
    if (M == 0 || N == 0) return MIXQ_OK;
    const long long total = static_cast<long long>(M) * N;
    hipLaunchKernelGGL(dequant_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, mixq_stream(stream), y32,
                       ldy32, x_scale, scale_col, addend, lda, bias, y, ldy, M, N, act);
    return mixq_launch_status();
}

extern "C" int mixq_gemm_set_config(int cfg) {
    if (cfg < -1 || cfg >= NUM_CFGS) return MIXQ_EINVAL;
    g_forced_cfg = cfg;
    return MIXQ_OK;
}
extern "C" int mixq_gemm_num_configs(void) { return NUM_CFGS; }
extern "C" int mixq_gemm_config_name(int cfg, char* buf, int cap) {
    if (cfg < 0 || cfg >= NUM_CFGS || !buf || cap <= 0) return MIXQ_EINVAL;
    snprintf(buf, cap, "%s", g_cfgs[cfg].name);
    return MIXQ_OK;
}
extern "C" int mixq_gemm_pick_config(int M, int N, int K, int bit) {
    if (M <= 0 || N <= 0 || K <= 0 || (bit != 4 && bit != 8)) return MIXQ_EINVAL;
    return pick_config(M, N, bit == 8 ? K : K / 2);
}

extern "C" int mixq_version(void) { return 1000; }

extern "C" int mixq_device_info(char* buf, int cap) {
    if (!buf || cap <= 0) return MIXQ_EINVAL;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return MIXQ_ENODEV; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { (void)hipGetLastError(); return MIXQ_ENODEV; }
    snprintf(buf, cap, "%s arch=%s cu=%d clock_khz=%d lds_per_block=%zu l2=%d", p.name, p.gcnArchName, p.multiProcessorCount,
             p.clockRate, p.sharedMemPerBlock, p.l2CacheSize);
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? MIXQ_OK : MIXQ_ENODEV;
}
