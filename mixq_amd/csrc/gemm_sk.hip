// Stream-K form of the fused int8/int4 MFMA GEMM (packed P16x64 operands only).
//
// Why: at the metric shape (M = 512, N = 11008 = 2^8 * 43) a 256x128 tiling has 172 tiles for 256 CUs, a 256x256 tiling
// 86; the data-parallel kernel leaves a third of the chip idle and every busy CU is bound by its own LDS-DMA feed rate
// (~50 B/clk).  Here the K loops of all tiles are laid end to end (tiles x ksteps "units") and cut into
// one equal span per CU: every CU streams the same number of k-steps, so both the MFMA work and the operand feed are
// spread over all 256 CUs.  A span that does not start at k = 0 of its tile produces an int32 partial tile; int32
// addition is exact and order-independent, so the result is bit-identical to the data-parallel kernel.
//
// Protocol (placement-independent, cdna_hip_programming.md G16):
//   * logical workgroup L owns units [u0, u1).  Its FIRST segment may be the middle/tail of a tile (k_a > 0): it stores
//     its accumulators, in register order, to workspace slot L with write-through (sc1) 16-byte stores, drains them
//     (s_waitcnt vmcnt(0) in every wave), barriers, and one lane publishes flag[L] = 1 (relaxed, agent scope).
//   * the workgroup that owns the HEAD of a tile (k_a = 0) finishes it: this is always its LAST segment, while the
//     contributors (L+1, L+2, ...) produced their partials as their FIRST segment, so the partials are long there when
//     the head arrives - nobody waits in steady state and there is no circular dependency.  The finisher polls
//     flag[c] with one lane (relaxed) and issues ONE agent-scope acquire, barriers, adds the partial (plain 16-byte
//     loads), resets flag[c] = 0 for the next launch, and runs the fused epilogue.
//   * all G workgroups must be resident at once: G <= number of CUs and one workgroup per CU (LDS > 80 KiB).
// The workspace (G slots of BM*BN int32 + G flags) is provided by the host once, zero-initialised
// (mixq_gemm_set_workspace); launches that use it must be serialised on one stream.
#include "common.h"
#include "gemm_sk.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// Round 5: the kernels of this file are compiled into the TUNING build only (make tuning -> libmixq_hip_tuning.so).  The automatic choice
// never picked them since round 2, and the one place where a stream-K schedule had something to offer - the partial last round of tiles at
// prefill sizes - is covered by the N split of the data-parallel kernels (gemm_wreg.hip: mixq_wr_split; profiles/r05_prefill_sweep.txt).
// Round 6: this file is a source of the tuning build ONLY (csrc/Makefile: TSRCS); the product library does not compile it - the workspace
// registration its callers may still make lives in gemm.hip, the entry points below are inline stubs there (gemm_sk.h).
#ifndef MIXQ_TUNING
#error "gemm_sk.hip belongs to the tuning build (make tuning): the product library does not contain the stream-K kernels"
#endif
namespace {

struct SkArgs {
    const uint8_t* qx;  const uint8_t* qw;
    const uint16_t* sx; const uint16_t* sw;
    const uint16_t* xo; const uint16_t* wo;
    const int32_t* n_out_dev;
    const uint16_t* addend; const uint16_t* bias;
    uint16_t* y;
    int32_t* ws; int32_t* flags;
    int M, N, KB;
    int ldxo, ldwo, n_out, lda, ldy;
    int act;
    int tiles_m, tiles_n;
    int xrows16, wrows16;
    int G, nk, total_units;
    int dbg;                                          // tuning only: bit0 = skip the partial hand-off (wrong results)
};

constexpr int BKB = 64;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void glds16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz(int r, int c) { return c ^ ((0 - (r >> 2)) & 3); }
__device__ __forceinline__ float silu(float v) { return mixq_silu(v); }   // (common.h: never contracted with the bias addition)

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int MODE>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_sk_kernel(const SkArgs a)
{
    constexpr int CW = WAVES_M * WAVES_N;
    constexpr int NT = CW * 64;
    constexpr int CH = BKB / 16;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int TI = (BM + BN) * CH / 64;
    constexpr int LOADS = (TI + CW - 1) / CW;
    constexpr int STAGE_BYTES = (BM + BN) * BKB;
    constexpr int LOOK = NSTAGE - 2;
    constexpr int NEWER = LOOK - 1;
    constexpr bool I4 = (MODE == 1);
    constexpr int OPITCH = BN * 2 + 16;
    constexpr int HALF_ROWS = BM / 2;                  // the fp16 tile is staged and stored in two row halves
    static_assert(BM % 32 == 0 && BN % 16 == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
    static_assert(LOOK >= 1 && LOADS * NEWER < 64, "vmcnt range");
    static_assert(HALF_ROWS * OPITCH <= NSTAGE * STAGE_BYTES, "output staging must fit in the ring");
    static_assert(WAVES_M % 2 == 0, "row halves are split between wave rows");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;
    const int nk = a.nk;

    // logical id: the workgroups of one XCD (b % 8) own consecutive spans, so the contributors of a tile share an L2
    int L = blockIdx.x;
    if ((a.G & 7) == 0) L = (blockIdx.x & 7) * (a.G >> 3) + (blockIdx.x >> 3);
    const int per = a.total_units / a.G, rem = a.total_units % a.G;
    int u = L * per + (L < rem ? L : rem);
    const int uend = u + per + (L < rem ? 1 : 0);

    int wrow[NI], xrow[MI];
#pragma unroll
    for (int i = 0; i < NI; ++i) wrow[i] = wn * WN + i * 32 + lr;
#pragma unroll
    for (int j = 0; j < MI; ++j) xrow[j] = wm * WM + j * 32 + lr;

    const __amdgpu_buffer_rsrc_t ws_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, static_cast<int>(static_cast<long long>(a.G) * BM * BN * 4 > 0x7fffffffLL
                                                                         ? 0x7fffffff
                                                                         : static_cast<long long>(a.G) * BM * BN * 4),
                                          0x00020000);

    while (u < uend) {
        const int tile = u / nk, ka = u - tile * nk;
        const int len = (nk - ka) < (uend - u) ? (nk - ka) : (uend - u);
        const int kb = ka + len;
        const int tn = tile / a.tiles_m, tm = tile - tn * a.tiles_m;
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- DMA piece table (packed operands: one contiguous KiB per piece), starting at k-step ka -------------
        const uint8_t* nsrc[LOADS];
        int kstr[LOADS];
        int piece[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            int pw = i * CW + wave;
            if (pw >= TI) pw = wave;
            piece[i] = pw;
            const int qd = pw * 64 + lane;
            const int rr = qd / CH, pc = qd % CH;
            const bool is_w = rr < BN;
            const int r = is_w ? rr : rr - BN;
            const int row0 = is_w ? n0 : m0, rows16 = is_w ? a.wrows16 : a.xrows16;
            const uint8_t* base = is_w ? a.qw : a.qx;
            int rb = (row0 + r) >> 4; rb = rb < (rows16 >> 4) ? rb : (rows16 >> 4) - 1;
            kstr[i] = rows16 * 64;
            nsrc[i] = base + static_cast<size_t>(rb) * 1024 + (r & 15) * 64 + pc * 16 + static_cast<size_t>(ka) * kstr[i];
        }
        auto stage = [&](int buf) {
            uint8_t* base = lds + buf * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < LOADS; ++i) { glds16(nsrc[i], base + piece[i] * 1024); nsrc[i] += kstr[i]; }
        };

        i32x16 acc[NI][MI];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

        i32x4 wf[2][NI], xf[2][MI];
        auto load_one = [&](int p, int buf, int sb, int idx) {
            const uint8_t* wb = lds + buf * STAGE_BYTES;
            const uint8_t* xb = wb + BN * BKB;
            if (idx == 0 || idx > MI) {
                const int i = idx == 0 ? 0 : idx - MI;
                wf[p][i] = *reinterpret_cast<const i32x4*>(wb + wrow[i] * BKB + swz(wrow[i], sb * 2 + lh) * 16);
            } else {
                const int j = idx - 1;
                xf[p][j] = *reinterpret_cast<const i32x4*>(xb + xrow[j] * BKB + swz(xrow[j], sb * 2 + lh) * 16);
            }
        };
        auto region = [&](auto p_c, auto nd_c, auto ng_c, int rbuf, int rsub, int dma_buf) {
            constexpr int P = decltype(p_c)::value, ND_ = decltype(nd_c)::value, NG_ = decltype(ng_c)::value;
            constexpr int NM = NI * MI, SL = NM > 1 ? NM - 1 : 1;
            uint8_t* dbase = lds + dma_buf * STAGE_BYTES;
            i32x4 wl[NI], wh[NI], xl[MI], xh[MI];
            if constexpr (I4) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t v = static_cast<uint32_t>(wf[P][i][d]);
                        wl[i][d] = static_cast<int>((v << 4) & 0xf0f0f0f0u);
                        wh[i][d] = static_cast<int>(v & 0xf0f0f0f0u);
                    }
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t v = static_cast<uint32_t>(xf[P][j][d]);
                        xl[j][d] = static_cast<int>((v << 4) & 0xf0f0f0f0u);
                        xh[j][d] = static_cast<int>(v & 0xf0f0f0f0u);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int i = m / MI, j = m % MI;
                if constexpr (!I4) {
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[P][i], xf[P][j], acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wl[i], xl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wh[i], xh[j], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (m < SL) {
                    if constexpr (NG_ > 0) {
#pragma unroll
                        for (int g = (NG_ * m) / SL; g < (NG_ * (m + 1)) / SL; ++g) { glds16(nsrc[g], dbase + piece[g] * 1024); nsrc[g] += kstr[g]; }
                    }
#pragma unroll
                    for (int d = (ND_ * m) / SL; d < (ND_ * (m + 1)) / SL; ++d) load_one(1 - P, rbuf, rsub, d);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using IND = std::integral_constant<int, NI + MI>;
        using ING = std::integral_constant<int, LOADS>;

        // ---- software-pipelined k loop over [ka, kb) (same structure as gemm.hip, LOADERS = 0) -------------------
#pragma unroll
        for (int s = 0; s < LOOK; ++s)
            if (s < len) stage(s);
        if (NEWER < len) wait_vmcnt<LOADS * NEWER>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int d = 0; d < NI + MI; ++d) load_one(0, 0, 0, d);
        int cur = 0, nxt = LOOK % NSTAGE;
        auto body = [&](auto issue_c, auto next_c) {
            constexpr bool ISSUE = decltype(issue_c)::value, NEXT = decltype(next_c)::value;
            const int cur1 = (cur + 1 == NSTAGE) ? 0 : cur + 1;
            if constexpr (ISSUE) region(I0{}, IND{}, ING{}, cur, 1, nxt);
            else                 region(I0{}, IND{}, I0{}, cur, 1, nxt);
            if constexpr (NEXT) {
                if constexpr (ISSUE) wait_vmcnt<LOADS * NEWER>();
                else                 wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                region(I1{}, IND{}, I0{}, cur1, 0, nxt);
            } else {
                region(I1{}, I0{}, I0{}, cur1, 0, nxt);
            }
            cur = cur1;
            nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
        };
        int kt = 0;
        for (; kt + LOOK < len; ++kt) body(std::true_type{}, std::true_type{});
        for (; kt + 1 < len; ++kt)    body(std::false_type{}, std::true_type{});
        body(std::false_type{}, std::false_type{});

        // =============================================================================================================
        // end of segment
        // =============================================================================================================
        if (ka > 0 && (a.dbg & 1)) {
        } else if (ka > 0) {
            // ---- contributor: publish the int32 partial (register order: one KiB per store instruction) -------------
            const int slot_off = L * (BM * BN * 4) + wave * (NI * MI * 4096) + lane * 16;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const u32x4 v = {static_cast<uint32_t>(acc[i][j][4 * g]), static_cast<uint32_t>(acc[i][j][4 * g + 1]),
                                         static_cast<uint32_t>(acc[i][j][4 * g + 2]), static_cast<uint32_t>(acc[i][j][4 * g + 3])};
                        __builtin_amdgcn_raw_buffer_store_b128(v, ws_rsrc, slot_off + ((i * MI + j) * 4 + g) * 1024, 0, 16 /* sc1 */);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains its own stores
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.flags + L, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // ---- finisher: request the scales, fold in the contributors' partials, fused epilogue -------------------
            u32x2 swp[NI][4];
            uint16_t sxh[MI];
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int m = m0 + xrow[j];
                sxh[j] = (m < a.M) ? a.sx[m] : static_cast<uint16_t>(0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * WN + i * 32 + 4 * lh + 8 * g;
                    if (n + 3 < a.N) swp[i][g] = *reinterpret_cast<const u32x2*>(a.sw + n);
                    else {
                        uint32_t h[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (n + e < a.N) ? a.sw[n + e] : 0u;
                        swp[i][g] = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
                    }
                }
            int kcov = kb, c = L + 1;
            while (kcov < nk && !(a.dbg & 1)) {
                const int cnt_c = per + (c < rem ? 1 : 0);
                const int seg_c = (nk - kcov) < cnt_c ? (nk - kcov) : cnt_c;
                if (tid == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(a.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 26)) __builtin_trap();      // a stuck producer must not hang the GPU silently
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const int slot_off = c * (BM * BN * 4) + wave * (NI * MI * 4096) + lane * 16;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws_rsrc, slot_off + ((i * MI + j) * 4 + g) * 1024, 0, 16 /* sc1 */);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += static_cast<int>(v[e]);
                        }
                __syncthreads();
                if (tid == 0) __hip_atomic_store(a.flags + c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                kcov += seg_c;
                ++c;
            }

            int n_out = a.n_out;
            if (a.n_out_dev) { const int nd = *a.n_out_dev; n_out = nd < n_out ? nd : n_out; }
            if (!a.xo || !a.wo) n_out = 0;
            const int ksteps = (n_out + 15) >> 4;
            constexpr float PRE = I4 ? (1.f / 256.f) : 1.f;
            const bool staged = ((a.N & 7) == 0) && ((a.ldy & 7) == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
            float sxv[MI];
#pragma unroll
            for (int j = 0; j < MI; ++j) sxv[j] = h2f(sxh[j]) * PRE;

            // two passes over the row halves of the tile: wave rows of half h stage their fp16 results, everyone stores
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                __builtin_amdgcn_s_barrier();                               // ring / previous half no longer read
                if ((wm / (WAVES_M / 2)) == half) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int nloc = wn * WN + i * 32 + 4 * lh;
                        const int nb = n0 + nloc;
                        float swv[16];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            swv[4 * g]     = h2f(static_cast<uint16_t>(swp[i][g].x & 0xffffu));
                            swv[4 * g + 1] = h2f(static_cast<uint16_t>(swp[i][g].x >> 16));
                            swv[4 * g + 2] = h2f(static_cast<uint16_t>(swp[i][g].y & 0xffffu));
                            swv[4 * g + 3] = h2f(static_cast<uint16_t>(swp[i][g].y >> 16));
                        }
#pragma unroll
                        for (int j = 0; j < MI; ++j) {
                            f32x16 f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) f[r] = static_cast<float>(acc[i][j][r]) * sxv[j] * swv[r];
                            if (ksteps > 0) {
                                int wr = n0 + wrow[i]; wr = wr < a.N ? wr : a.N - 1;
                                int xr = m0 + xrow[j]; xr = xr < a.M ? xr : a.M - 1;
                                const uint16_t* wp = a.wo + static_cast<size_t>(wr) * a.ldwo + lh * 8;
                                const uint16_t* xp = a.xo + static_cast<size_t>(xr) * a.ldxo + lh * 8;
                                for (int kk = 0; kk < ksteps; ++kk) {
                                    u32x4 wq = *reinterpret_cast<const u32x4*>(wp + kk * 16);
                                    u32x4 xq = *reinterpret_cast<const u32x4*>(xp + kk * 16);
                                    const int kq = kk * 16 + lh * 8;
                                    if (kq + 8 > n_out) {
#pragma unroll
                                        for (int d = 0; d < 4; ++d) {
                                            uint32_t keep = 0;
                                            if (kq + 2 * d < n_out)     keep |= 0x0000ffffu;
                                            if (kq + 2 * d + 1 < n_out) keep |= 0xffff0000u;
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
                                    u32x2 o;
                                    o.x = static_cast<uint32_t>(f2h(v[0])) | (static_cast<uint32_t>(f2h(v[1])) << 16);
                                    o.y = static_cast<uint32_t>(f2h(v[2])) | (static_cast<uint32_t>(f2h(v[3])) << 16);
                                    if (staged) {
                                        *reinterpret_cast<u32x2*>(lds + (xrow[j] - half * HALF_ROWS) * OPITCH + (nloc + 8 * g) * 2) = o;
                                    } else {
                                        uint16_t* yp = a.y + static_cast<size_t>(m) * a.ldy + n;
                                        if (full) *reinterpret_cast<u32x2*>(yp) = o;
                                        else {
#pragma unroll
                                            for (int e = 0; e < 4; ++e) if (n + e < a.N) yp[e] = f2h(v[e]);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // tile writes performed, not just issued
                __builtin_amdgcn_s_barrier();                               // staging half complete
                if (staged) {
                    constexpr int CPR = BN * 2 / 16;
                    for (int q = tid; q < HALF_ROWS * CPR; q += NT) {
                        const int r = q / CPR, cc = q - r * CPR;
                        const int m = m0 + half * HALF_ROWS + r, n = n0 + cc * 8;
                        if (m < a.M && n < a.N) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * OPITCH + cc * 16);
                            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(a.y + static_cast<size_t>(m) * a.ldy + n));
                        }
                    }
                }
            }
        }
        __syncthreads();            // the ring is reused by the next segment's DMA
        u += len;
    }
}

struct SkConfig {
    const char* name;
    int bm, bn, waves, nstage;
    void (*k8)(const SkArgs);
    void (*k4)(const SkArgs);
};
#define MIXQ_SK(BM, BN, WMv, WNv, NS) \
    { "sk" #BM "x" #BN "_w" #WMv "x" #WNv "_s" #NS, BM, BN, (WMv) * (WNv), NS, gemm_sk_kernel<BM, BN, WMv, WNv, NS, 0>, gemm_sk_kernel<BM, BN, WMv, WNv, NS, 1> }
const SkConfig g_sk[] = {
    MIXQ_SK(256, 128, 4, 2, 5),
    MIXQ_SK(256, 256, 4, 2, 4),
    MIXQ_SK(256, 128, 2, 2, 5),
    MIXQ_SK(128, 128, 2, 2, 5),
};
constexpr int NUM_SK = sizeof(g_sk) / sizeof(g_sk[0]);

}  // namespace

int mixq_sk_num_configs() { return NUM_SK; }
const char* mixq_sk_config_name(int c) { return (c >= 0 && c < NUM_SK) ? g_sk[c].name : ""; }
size_t mixq_sk_workspace_need(int c, int G) { return MIXQ_WS_FLAG_BYTES + static_cast<size_t>(G) * g_sk[c].bm * g_sk[c].bn * 4; }

bool mixq_sk_usable(int c) {
    MixqDevState* d = mixq_dev_state();
    if (c < 0 || c >= NUM_SK || !d || !d->ws) return false;
    const int cus = mixq_num_cus();
    return cus > 0 && mixq_sk_workspace_need(c, cus) <= d->bytes;
}

int mixq_sk_launch(int c, int bit, const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                   const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                   const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int KB, int act,
                   hipStream_t st)
{
    if (!mixq_sk_usable(c)) return MIXQ_EINVAL;
    const SkConfig& g = g_sk[c];
    SkArgs a;
    memset(&a, 0, sizeof(a));
    a.qx = static_cast<const uint8_t*>(q_x); a.qw = static_cast<const uint8_t*>(q_w);
    a.sx = x_scale; a.sw = scale_col; a.xo = x_out; a.wo = w_out; a.n_out_dev = n_out_dev; a.addend = addend; a.bias = bias; a.y = y;
    a.M = M; a.N = N; a.KB = KB; a.ldxo = ldxo; a.ldwo = ldwo; a.n_out = n_out; a.lda = lda; a.ldy = ldy; a.act = act;
    a.tiles_m = (M + g.bm - 1) / g.bm; a.tiles_n = (N + g.bn - 1) / g.bn;
    a.xrows16 = (M + 15) & ~15; a.wrows16 = (N + 15) & ~15;
    a.nk = KB / BKB;
    a.total_units = a.tiles_m * a.tiles_n * a.nk;
    MixqDevState* d = mixq_dev_state();
    const int cus = mixq_num_cus();
    a.G = cus < a.total_units ? cus : a.total_units;
    if (const char* e = getenv("MIXQ_SK_G")) { const int g2 = atoi(e); if (g2 > 0 && g2 <= a.G) a.G = g2; }     // tuning only
    if (const char* e = getenv("MIXQ_SK_DBG")) a.dbg = atoi(e);
    a.flags = static_cast<int32_t*>(d->ws);
    a.ws = reinterpret_cast<int32_t*>(static_cast<char*>(d->ws) + MIXQ_WS_FLAG_BYTES);
    if (static_cast<size_t>(a.G) * 4 > MIXQ_WS_FLAG_BYTES) return MIXQ_EINVAL;
    void (*k)(const SkArgs) = bit == 8 ? g.k8 : g.k4;
    const size_t shm = static_cast<size_t>(g.bm + g.bn) * BKB * g.nstage;
    if (int rc = mixq_ensure_dynamic_lds(reinterpret_cast<const void*>(k), shm)) return rc;
    hipLaunchKernelGGL(k, dim3(a.G), dim3(g.waves * 64), shm, st, a);
    return mixq_launch_status();
}

// enough for every stream-K configuration on a 256-CU part: 256 slots of the largest tile + flags
long long mixq_sk_workspace_bytes() {
    size_t need = 0;
    for (int c = 0; c < NUM_SK; ++c) { const size_t n = mixq_sk_workspace_need(c, 256); if (n > need) need = n; }
    return static_cast<long long>(need);
}
