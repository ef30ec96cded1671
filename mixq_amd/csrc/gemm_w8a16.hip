// Weight-only W8A16 Linear for gfx950:  Y[M,N] = X[M,K] (fp16) * (Wq[K,N] (int8) * scale_col[N]) + bias
// -- what the reference reaches through EETQ's `w8_a16_gemm(x, q_weight, scale_col)` (modules/linear.py:178-184; GPT-J
// fc_out and the "down_weight_only" variants, utils/module.py:4-12).  EETQ is a third-party CUDA extension that is not
// part of the reference tree and is not version-pinned; its CUTLASS-interleaved weight image is an implementation
// detail of that library (mixq_amd/eetq.py converts to and from it).  Here (round 2: the weights-in-registers structure of
// gemm_wreg.hip; round 1 staged both operands through LDS and stopped at 35 % of the fp16 peak, LDS-port-bound):
//
//  * the int8 weights are re-tiled ONCE (mixq_pack_w8a16) from the checkpoint's [K,N] matrix into MIXQ_FMT_F16X64 blocks
//    (16 output channels x 64 k: byte c*256 + r*16 + b = channel r, k = 16 c + b), stored offset-binary (q + 128) so that
//    the int8 -> fp16 conversion in the k loop is two VALU per pair of elements:
//        v_perm_b32   -> halves 0x6400 | u   (= 1024 + u exactly)
//        v_pk_add_f16 -> - 1152              (= u - 128 = q, exact)
//    A block is one contiguous KiB = one global_load_dwordx4 of a wave; the 16 bytes a lane receives are its A-operand bytes
//    for the two v_mfma_f32_16x16x32_f16 of a 64-deep k-step (bytes 8 s .. 8 s + 7 for MFMA s: k = 16 (l>>4) + 8 s ..), in a
//    register ring D k-steps deep.  Weights never touch LDS.
//  * X stays the caller's fp16 [M,K] matrix; a k-step (64 elements = 128 bytes per row) of the tile's 16 MB rows is staged by
//    LDS-DMA in pieces of 8 rows x 128 bytes (whole 128-byte lines) with an 8-chunk XOR swizzle applied on the SOURCE
//    address: chunk q of row t lands at position q ^ ((t >> 1) & 5) of the row's 128 bytes.  With ds_read_b128's lane groups
//    ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... MI355X_MICROARCH.md, LDS) the 16x16x32 B-fragment read (lane = token l&15, chunk
//    2 (l>>4) + s) then hits 16 distinct 16-byte slots per group.  The k order inside a k-step is permuted identically for both
//    operands (MFMA s, lane quarter c covers k = 16 c + 8 s .. +8).
//  * 4 consumer waves side by side along N (each 16 MB tokens x 16 WNB channels), loader wave(s) for the X ring, one s_barrier
//    per k-step (in its middle: stage kt+1 is first needed by the second MFMA step), X fragments refilled in place behind
//    their last MFMA, the conversion of the NEXT MFMA step's weight fragments issued behind the MFMAs of the current one
//    (1 VALU per MFMA at MB = 8: what a wave co-issues for free, tools/ubench_mfma_valu.hip).
#include "common.h"
#include <stdio.h>
#include <type_traits>

namespace {

struct WoArgs {
    const uint16_t* x; const uint8_t* w; const uint16_t* sw; const uint16_t* bias; uint16_t* y;
    int M, N, K, ldx, ldy, tiles_m, tiles_n, wblocks;
};

constexpr int WO_CW = 4;                              // consumer waves, 1 x 4 along N

template <int N> __device__ __forceinline__ void wo_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

__device__ __forceinline__ void wo_glds16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// position of 16-byte chunk q inside the 128 bytes a token row occupies per k-step (t: row inside its 16-row block)
__device__ __forceinline__ int wo_xpos(int t, int q) { return q ^ ((t >> 1) & 5); }

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

template <int I, int N, class F> __device__ __forceinline__ void wo_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); wo_static_for<I + 1, N>(f); }
}

// 4 offset-binary bytes -> 4 halves (two dwords)
__device__ __forceinline__ void cvt_u8x4(uint32_t d, uint32_t& o0, uint32_t& o1) {
    const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, d, 0x04010400u);      // (0x64, b1, 0x64, b0)
    const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, d, 0x04030402u);      // (0x64, b3, 0x64, b2)
    const h2_t off = {static_cast<_Float16>(1152.f), static_cast<_Float16>(1152.f)};
    o0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, p0) - off);
    o1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, p1) - off);
}

// MB: 16-row token blocks per tile (BM = 16 MB), WNB: 16-channel weight blocks per wave (BN = 64 WNB), NSTAGE: X ring depth,
// D: weight register ring depth (k-steps), LOADERS: DMA waves.  ABL (tuning only): 0 normal, 1 no weight loads, 2 no X traffic,
// 3 MFMA only, 4 MFMA + conversions (results of 1-4 are garbage), 5 no prologue ramp in the loader.
template <int MB, int WNB, int NSTAGE, int D, int LOADERS, int ABL = 0>
__global__ __launch_bounds__((WO_CW + LOADERS) * 64) void gemm_w8a16_kernel(const WoArgs a)
{
    constexpr int CW = WO_CW, NT = (CW + LOADERS) * 64;
    constexpr int BM = MB * 16, WN = WNB * 16, BN = CW * WN;
    constexpr int STAGE_BYTES = MB * 2048;               // 16 MB rows x 128 bytes
    constexpr int PIECES = MB * 2, LOADS = PIECES / LOADERS;
    constexpr int LOOK = NSTAGE - 2, NEWER = LOOK - 1;
    constexpr int OPITCH = BN * 2 + 16;
    constexpr bool NOW = ABL == 1 || ABL == 3 || ABL == 4, NOX = ABL >= 2 && ABL <= 4;
    static_assert(LOADERS >= 1 && PIECES % LOADERS == 0, "pieces must divide evenly over the loader waves");
    static_assert(LOOK >= 1 && LOADS * NEWER < 64, "vmcnt range");
    static_assert(NSTAGE * STAGE_BYTES <= 160 * 1024 && BM * OPITCH <= NSTAGE * STAGE_BYTES, "LDS: ring and staging tile");
    static_assert(D >= 2 && D <= 5, "weight ring depth");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int ntiles = a.tiles_m * a.tiles_n;
    int tile;
    {   // XCD-aware remap (block b runs on XCD b % 8): consecutive logical tiles share an L2
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, x = b & 7, s = b >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
    }
    const int tn = tile / a.tiles_m, tm = tile - tn * a.tiles_m;               // m fastest: a weight panel stays on one XCD
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int nk = a.K >> 6;

    // ================================================================================================ loader wave(s)
    if (wave >= CW) {
        __builtin_amdgcn_s_setprio(2);
        const int lw = wave - CW;
        const uint8_t* src[LOADS];
        int dsto[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int p = lw + i * LOADERS;                                      // piece: tile rows 8 p .. 8 p + 7
            const int t = p * 8 + (lane >> 3);
            int row = m0 + t; row = row < a.M ? row : a.M - 1;                   // rows past M: loaded, computed, dropped
            src[i] = reinterpret_cast<const uint8_t*>(a.x) + static_cast<size_t>(row) * a.ldx * 2 + wo_xpos(t & 15, lane & 7) * 16;
            dsto[i] = p * 1024;
        }
        size_t koff = 0;
        auto stage = [&](int slot) {
            if constexpr (!NOX) {
#pragma unroll
                for (int i = 0; i < LOADS; ++i) wo_glds16(src[i] + koff, lds + slot * STAGE_BYTES + dsto[i]);
            }
            koff += 128;
        };
        // ramp (as in gemm_wreg.hip): RP stages up front, then two per k-step until the loader is LOOK stages ahead, instead of every
        // workgroup asking for LOOK x 16 MB KiB at once in front of its first stage
        constexpr int RP = (LOOK > 4 && ABL != 5) ? 3 : LOOK;
        int nxt, kt = 0;
        if (RP < LOOK && nk >= 2 * LOOK) {
#pragma unroll
            for (int s = 0; s < RP; ++s) stage(s);
            wo_wait_vmcnt<LOADS * (RP - 1)>();
            __builtin_amdgcn_s_barrier();                                        // B0: stage 0 landed
            wo_static_for<0, LOOK - RP>([&](auto i_c) {
                constexpr int i = decltype(i_c)::value;
                stage((RP + 2 * i) % NSTAGE);
                stage((RP + 2 * i + 1) % NSTAGE);
                wo_wait_vmcnt<LOADS * (RP + i)>();                               // stage i + 1 landed: RP + i younger stages may be in flight
                __builtin_amdgcn_s_barrier();
            });
            kt = LOOK - RP;
            nxt = (2 * LOOK - RP) % NSTAGE;
        } else {
#pragma unroll
            for (int s = 0; s < LOOK; ++s)
                if (s < nk) stage(s);
            if (NEWER < nk) wo_wait_vmcnt<LOADS * NEWER>(); else wo_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                                        // B0: stage 0 landed
            nxt = LOOK % NSTAGE;
        }
        for (; kt + LOOK < nk; ++kt) {
            stage(nxt);
            wo_wait_vmcnt<LOADS * NEWER>();                                      // stage kt+1 landed
            __builtin_amdgcn_s_barrier();
            nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
        }
        for (; kt + 1 < nk; ++kt) {
            wo_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                                            // epilogue: ring free
        __builtin_amdgcn_s_barrier();                                            // epilogue: staging tile complete
    }

    // ================================================================================================ consumer waves
    const int lm = lane & 15, lq = lane >> 4;
    const int nw0 = n0 + wave * WN;                                              // first channel of this wave (wave < CW)
    if (wave < CW) {
        f32x4 acc[MB][WNB];
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int i = 0; i < WNB; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

        // weight stream: wave-uniform block bases (scalar registers), one lane offset
        const uint8_t* wb[WNB];
#pragma unroll
        for (int i = 0; i < WNB; ++i) {
            int rb = (nw0 >> 4) + i; rb = rb < a.wblocks ? rb : a.wblocks - 1;  // blocks past N: computed and dropped
            wb[i] = a.w + static_cast<size_t>(rb) * 1024;
        }
        const size_t wks = static_cast<size_t>(a.wblocks) * 1024;
        size_t woff = 0;
        const int lane16 = lane * 16;
        // token fragment of MFMA step s: row lm, 16-byte chunk 2 lq + s
        const int xoff0 = lm * 128 + (wo_xpos(lm, 2 * lq) << 4), xoff1 = lm * 128 + (wo_xpos(lm, 2 * lq + 1) << 4);

        // Raw weight ring: NSLOT = D + 1 slots; k-step kt is consumed from slot kt % NSLOT while the loads of k-step kt + D go into
        // the slot k-step kt - 1 freed.  Inline-asm loads on read-write operands with hand-counted waits, for the reasons given in
        // gemm_wreg.hip (the compiler would drain the ring at every loop header, and may copy in-flight registers at a merge).
        constexpr int NSLOT = D + 1;
        i32x4 wq[NSLOT][WNB];
        u32x4 wc[2][WNB];                                  // converted fragments of MFMA step 0 / 1
        u32x4 xf[MB];
#pragma unroll
        for (int d = 0; d < NSLOT; ++d)
#pragma unroll
            for (int i = 0; i < WNB; ++i) {
                wq[d][i] = i32x4{lane, lane, lane, lane};
                if constexpr (ABL >= 1 && ABL <= 4) asm volatile("" : "+v"(wq[d][i]));
            }
        if constexpr (ABL >= 1 && ABL <= 4) {
#pragma unroll
            for (int j = 0; j < MB; ++j) { xf[j] = u32x4{(uint32_t)lane, 1u, 2u, 3u}; asm volatile("" : "+v"(xf[j])); }
#pragma unroll
            for (int i = 0; i < WNB; ++i) { wc[0][i] = wc[1][i] = u32x4{(uint32_t)lane, 1u, 2u, 3u}; asm volatile("" : "+v"(wc[0][i]), "+v"(wc[1][i])); }
        }
        auto wload1 = [&](auto d_c, int i, int cond) {
            constexpr int d = decltype(d_c)::value;
            if constexpr (!NOW) {
                const int cs = __builtin_amdgcn_readfirstlane(cond);
                const uint8_t* src = wb[i] + woff;
                i32x4& dst = wq[d][i];
                const int l16 = lane16;
                asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\tglobal_load_dwordx4 %0, %1, %2\n1:"
                             : "+v"(dst) : "v"(l16), "s"(src), "s"(cs) : "memory", "scc");
            }
        };
        auto wload1_always = [&](auto d_c, int i) {
            constexpr int d = decltype(d_c)::value;
            if constexpr (!NOW) {
                const uint8_t* src = wb[i] + woff;
                i32x4& dst = wq[d][i];
                const int l16 = lane16;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(l16), "s"(src) : "memory");
            }
        };
        auto wwait = [&](auto d_c, auto cnt_c) {
            constexpr int d = decltype(d_c)::value, CNT = decltype(cnt_c)::value;
            if constexpr (!NOW) {
                if constexpr (WNB == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wq[d][0]) : "i"(CNT));
                if constexpr (WNB == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wq[d][0]), "+v"(wq[d][1]) : "i"(CNT));
                if constexpr (WNB == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]) : "i"(CNT));
                if constexpr (WNB == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]), "+v"(wq[d][3]) : "i"(CNT));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto wwait_rt = [&](auto d_c, int younger) {       // run-time count (start and tail of the k loop), selected inside ONE statement
            constexpr int d = decltype(d_c)::value;
            if constexpr (!NOW) {
                const int sel = __builtin_amdgcn_readfirstlane(younger >= D - 1 ? D - 1 : (younger < 0 ? 0 : younger));
#define MIXQ_WO_WAITS                                                                                                           \
                "s_cmp_lt_u32 %[sel], 1\n\ts_cbranch_scc1 10f\n\ts_cmp_lt_u32 %[sel], 2\n\ts_cbranch_scc1 11f\n\t"               \
                "s_cmp_lt_u32 %[sel], 3\n\ts_cbranch_scc1 12f\n\ts_cmp_lt_u32 %[sel], 4\n\ts_cbranch_scc1 13f\n\t"               \
                "s_waitcnt vmcnt(%[c4])\n\ts_branch 19f\n"                                                                       \
                "10:\n\ts_waitcnt vmcnt(0)\n\ts_branch 19f\n11:\n\ts_waitcnt vmcnt(%[c1])\n\ts_branch 19f\n"                      \
                "12:\n\ts_waitcnt vmcnt(%[c2])\n\ts_branch 19f\n13:\n\ts_waitcnt vmcnt(%[c3])\n19:"
#define MIXQ_WO_WAIT_IN [sel] "s"(sel), [c1] "i"(WNB), [c2] "i"(2 * WNB), [c3] "i"(3 * WNB), [c4] "i"(4 * WNB)
                if constexpr (WNB == 1) asm volatile(MIXQ_WO_WAITS : "+v"(wq[d][0]) : MIXQ_WO_WAIT_IN : "scc");
                if constexpr (WNB == 2) asm volatile(MIXQ_WO_WAITS : "+v"(wq[d][0]), "+v"(wq[d][1]) : MIXQ_WO_WAIT_IN : "scc");
                if constexpr (WNB == 3) asm volatile(MIXQ_WO_WAITS : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]) : MIXQ_WO_WAIT_IN : "scc");
                if constexpr (WNB == 4) asm volatile(MIXQ_WO_WAITS : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]), "+v"(wq[d][3]) : MIXQ_WO_WAIT_IN : "scc");
#undef MIXQ_WO_WAITS
#undef MIXQ_WO_WAIT_IN
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto xread = [&](int slot, int j, int xo) {
            if constexpr (!NOX) xf[j] = *reinterpret_cast<const u32x4*>(lds + slot * STAGE_BYTES + j * 2048 + xo);
        };
        // conversion unit u (0 .. 2 WNB - 1) of MFMA step S from ring slot C: raw dword 2 S + (u & 1) of block u >> 1
        auto convert_unit = [&](auto s_c, auto c_c, int u) {
            constexpr int S = decltype(s_c)::value, C = decltype(c_c)::value;
            if constexpr (ABL != 3) {
                const int i = u >> 1, d = u & 1;
                uint32_t o0, o1;
                cvt_u8x4(static_cast<uint32_t>(wq[C][i][2 * S + d]), o0, o1);
                wc[S][i][2 * d] = o0; wc[S][i][2 * d + 1] = o1;
            }
        };
        // MFMA step S of the k-step in ring slot C.  Behind token block j's MFMAs: its fragment is re-read in place (chunk `xo` of
        // stage `rslot`), the conversion units of the NEXT MFMA step (slot CN, step 1 - S) that fall to j, and - step 0 only - the
        // weight loads of k-step kt + D into slot L that fall to j.
        auto mma_step = [&](auto s_c, auto c_c, auto cn_c, auto full_c, int rslot, int xo, int issue) {
            constexpr int S = decltype(s_c)::value, C = decltype(c_c)::value, L = (C + D) % NSLOT;
            constexpr bool FULL = decltype(full_c)::value;
            constexpr int NU = 2 * WNB;
            using LC = std::integral_constant<int, L>;
            using SN = std::integral_constant<int, 1 - S>;
#pragma unroll
            for (int j = 0; j < MB; ++j) {
#pragma unroll
                for (int i = 0; i < WNB; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[S][i]), __builtin_bit_cast(f16x8, xf[j]),
                                                                       acc[j][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                xread(rslot, j, xo);
#pragma unroll
                for (int u = (NU * j) / MB; u < (NU * (j + 1)) / MB; ++u) convert_unit(SN{}, cn_c, u);
                if constexpr (S == 0) {
#pragma unroll
                    for (int i = 0; i < WNB; ++i) {
                        const int pos = (WNB >= MB) ? (i % MB) : ((2 * i + 1) * MB) / (2 * WNB);
                        if (pos == j) { if constexpr (FULL) wload1_always(LC{}, i); else wload1(LC{}, i, issue); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // ---- prologue ------------------------------------------------------------------------------------------------
        auto prologue_w = [&](auto d_c) {
#pragma unroll
            for (int i = 0; i < WNB; ++i) wload1(d_c, i, decltype(d_c)::value < nk ? 1 : 0);
            if (decltype(d_c)::value < nk) woff += wks;
        };
        prologue_w(std::integral_constant<int, 0>{});
        prologue_w(std::integral_constant<int, 1>{});
        if constexpr (D > 2) prologue_w(std::integral_constant<int, 2>{});
        if constexpr (D > 3) prologue_w(std::integral_constant<int, 3>{});
        if constexpr (D > 4) prologue_w(std::integral_constant<int, 4>{});
        __builtin_amdgcn_s_barrier();                                            // B0: stage 0 landed
        __builtin_amdgcn_s_waitcnt(0xC07F);                                      // lgkmcnt(0): kernel arguments have long arrived (see gemm_wreg.hip)
#pragma unroll
        for (int j = 0; j < MB; ++j) xread(0, j, xoff0);
        wwait_rt(std::integral_constant<int, 0>{}, nk - 1);
#pragma unroll
        for (int u = 0; u < 2 * WNB; ++u) convert_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, u);

        // ---- k loop: unrolled by NSLOT so ring slots are compile-time registers ----------------------------------------
        int kt = 0, slot0 = 0;                                                   // X ring slot of stage kt
        auto one = [&](auto c_c, auto full_c) {
            constexpr int C = decltype(c_c)::value, CN = (C + 1) % NSLOT;
            constexpr bool FULL = decltype(full_c)::value;                       // FULL: stage kt+1 and k-step kt+D exist
            using CNc = std::integral_constant<int, CN>;
            const int slot1 = (slot0 + 1 == NSTAGE) ? 0 : slot0 + 1;
            // MFMA step 0: fragments refilled with step 1's chunks of THIS stage; step-1 weight fragments converted from slot C
            mma_step(std::integral_constant<int, 0>{}, c_c, c_c, full_c, slot0, xoff1, FULL ? 1 : (kt + D < nk ? 1 : 0));
            if (FULL || kt + D < nk) woff += wks;
            if constexpr (FULL) {
                wwait(CNc{}, std::integral_constant<int, WNB * (D - 1)>{});      // k-step kt+1's weights; the D-1 younger k-steps stay in flight
                __builtin_amdgcn_s_barrier();                                    // stage kt+1 landed
            } else if (kt + 1 < nk) {
                wwait_rt(CNc{}, nk - 2 - kt);
                __builtin_amdgcn_s_barrier();
            }
            // MFMA step 1: fragments refilled with step 0's chunks of the NEXT stage; step-0 weight fragments of the next k-step
            // converted from slot CN (past the end: harmless reads / conversions of dead data)
            mma_step(std::integral_constant<int, 1>{}, c_c, CNc{}, full_c, slot1, xoff0, 0);
            slot0 = slot1;
            ++kt;
        };
        auto group = [&](auto full_c) {
            one(std::integral_constant<int, 0>{}, full_c);
            one(std::integral_constant<int, 1>{}, full_c);
            one(std::integral_constant<int, 2>{}, full_c);
            if constexpr (NSLOT > 3) one(std::integral_constant<int, 3>{}, full_c);
            if constexpr (NSLOT > 4) one(std::integral_constant<int, 4>{}, full_c);
            if constexpr (NSLOT > 5) one(std::integral_constant<int, 5>{}, full_c);
        };
        while (kt + NSLOT + D <= nk) group(std::true_type{});                    // every k-step of the group has kt + D < nk
        while (kt < nk) {                                                        // fewer than NSLOT + D k-steps, guarded individually
            const int k0 = kt;
            auto tail_one = [&](auto c_c) { if (k0 + decltype(c_c)::value < nk) one(c_c, std::false_type{}); };
            tail_one(std::integral_constant<int, 0>{});
            tail_one(std::integral_constant<int, 1>{});
            tail_one(std::integral_constant<int, 2>{});
            if constexpr (NSLOT > 3) tail_one(std::integral_constant<int, 3>{});
            if constexpr (NSLOT > 4) tail_one(std::integral_constant<int, 4>{});
            if constexpr (NSLOT > 5) tail_one(std::integral_constant<int, 5>{});
        }

        // ---- epilogue: y = acc * scale_col[n] (+ bias[n]) -> fp16 tile in LDS ---------------------------------------
        __builtin_amdgcn_s_waitcnt(0x0F70);                                      // vmcnt(0): nothing of this wave is in flight any more
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // (the harmless past-the-end refills)
        __builtin_amdgcn_s_barrier();                                            // every wave is done reading the ring
        const bool has_bias = a.bias != nullptr;
        auto unpack4 = [](u32x2 v, float* o) {
            o[0] = h2f(static_cast<uint16_t>(v.x & 0xffffu)); o[1] = h2f(static_cast<uint16_t>(v.x >> 16));
            o[2] = h2f(static_cast<uint16_t>(v.y & 0xffffu)); o[3] = h2f(static_cast<uint16_t>(v.y >> 16));
        };
#pragma unroll
        for (int i = 0; i < WNB; ++i) {
            const int nloc = wave * WN + i * 16 + lq * 4, n = n0 + nloc;
            const int nc = n < a.N ? n : a.N - 4;                               // N % 4 == 0: groups are all in or all out
            float swv[4], bv[4];
            unpack4(*reinterpret_cast<const u32x2_u*>(a.sw + nc), swv);
            if (has_bias) unpack4(*reinterpret_cast<const u32x2_u*>(a.bias + nc), bv);
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[j][i][e] * swv[e];
                    if (has_bias) v[e] += bv[e];
                }
                u32x2 o;
                o.x = static_cast<uint32_t>(f2h(v[0])) | (static_cast<uint32_t>(f2h(v[1])) << 16);
                o.y = static_cast<uint32_t>(f2h(v[2])) | (static_cast<uint32_t>(f2h(v[3])) << 16);
                *reinterpret_cast<u32x2*>(lds + (j * 16 + lm) * OPITCH + nloc * 2) = o;
            }
        }
        // ds_write is asynchronous and a raw s_barrier does not wait for it: without this the last tile writes of a
        // wave can still be queued when another wave's copy-out read of the same bytes is served
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                            // staging tile complete
    }
    // all waves (loaders included): every LDS read of a lane before its first store.  16-byte row segments when N, ldy and y allow
    // it, else 8-byte ones (N % 4 == 0 is the only alignment the ABI asks of N and ldy)
    const bool wide = ((a.N & 7) == 0) && ((a.ldy & 7) == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
    if (wide) {
        constexpr int CPR = BN / 8, ITER = (BM * CPR + NT - 1) / NT;
        u32x4 v[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = tid + it * NT, r = q / CPR, c = q - r * CPR;
            if (q < BM * CPR) v[it] = *reinterpret_cast<const u32x4*>(lds + r * OPITCH + c * 16);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = tid + it * NT, r = q / CPR, c = q - r * CPR;
            const int m = m0 + r, n = n0 + c * 8;
            if (q < BM * CPR && m < a.M && n < a.N)
                __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(a.y + static_cast<size_t>(m) * a.ldy + n));
        }
    } else {
        constexpr int CPR = BN / 4;
        for (int q = tid; q < BM * CPR; q += NT) {
            const int r = q / CPR, c = q - r * CPR;
            const int m = m0 + r, n = n0 + c * 4;
            if (m < a.M && n < a.N) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(lds + r * OPITCH + c * 8);
                __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(a.y + static_cast<size_t>(m) * a.ldy + n));
            }
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
    const uint8_t* wp = a.w + static_cast<size_t>(wr >> 4) * 1024 + (wr & 15) * 16;     // F16X64: chunk c of row r at c * 256 + r * 16
    const uint8_t* xp = reinterpret_cast<const uint8_t*>(a.x) + static_cast<size_t>(xr) * a.ldx * 2 + lh * 32;
    const int wc0 = lh * 256, wc1 = (2 + lh) * 256;                           // chunk 2t + lh: MFMA steps 2t and 2t + 1
    const size_t wks = static_cast<size_t>(a.wblocks) * 1024;

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

// One-time re-tiling of the checkpoint's [K,N] int8 matrix into offset-binary MIXQ_FMT_F16X64 (rows = output channels): thread t
// writes the t-th 16-byte piece of the image, piece (kb, rb, c, r) = channel 16 rb + r, k = 64 kb + 16 c .. + 16.
__global__ __launch_bounds__(256) void pack_w8a16_kernel(const int8_t* __restrict__ qkn, uint8_t* __restrict__ dst, int K, int N, int rows16)
{
    const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    const long long total = static_cast<long long>(rows16) * (K >> 4);
    if (t >= total) return;
    const int per_kb = rows16 * 4;
    const int kb = static_cast<int>(t / per_kb), rem = static_cast<int>(t % per_kb);
    const int rb = rem >> 6, c = (rem >> 4) & 3, r = rem & 15;
    const int n = rb * 16 + r;
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
    const char* name; int mb, wnb, nstage, loaders;
    void (*k)(const WoArgs);
};
#define MIXQ_WO(MBv, WNBv, NS, Dv, LD, ABL, TAG) { "w8a16_" TAG, MBv, WNBv, NS, LD, gemm_w8a16_kernel<MBv, WNBv, NS, Dv, LD, ABL> }
const WoConfig g_wo[] = {
    // name = tile (token rows x channels) _ X ring depth _ weight ring depth _ loader waves
    MIXQ_WO(8, 3, 8, 3, 2, 0, "128x192_s8_d3_l2"),     // 0
    MIXQ_WO(8, 2, 8, 4, 2, 0, "128x128_s8_d4_l2"),     // 1
    MIXQ_WO(4, 2, 12, 4, 2, 0, "64x128_s12_d4_l2"),    // 2
    MIXQ_WO(4, 4, 12, 3, 2, 0, "64x256_s12_d3_l2"),    // 3
    MIXQ_WO(4, 3, 12, 4, 2, 0, "64x192_s12_d4_l2"),    // 4
    MIXQ_WO(4, 1, 8, 4, 1, 0, "64x64_s8_d4_l1"),       // 5
    MIXQ_WO(8, 3, 8, 4, 2, 0, "128x192_s8_d4_l2"),     // 6
    MIXQ_WO(8, 3, 6, 3, 1, 0, "128x192_s6_d3_l1"),     // 7
    MIXQ_WO(2, 1, 8, 5, 1, 0, "32x64_s8_d5_l1"),       // 8: small batches of wide layers (see gemm_wreg.hip's 32x64 tiling)
#ifdef MIXQ_TUNING                                     // ablation forms (results are garbage by design): tools build only (make tuning)
    MIXQ_WO(8, 3, 8, 3, 2, 1, "128x192_abl1_noW"),     // tuning: cfg 0 without the weight loads
    MIXQ_WO(8, 3, 8, 3, 2, 2, "128x192_abl2_noX"),     // tuning: cfg 0 without X traffic
    MIXQ_WO(8, 3, 8, 3, 2, 3, "128x192_abl3_mfma"),    // tuning: MFMA + epilogue only
    MIXQ_WO(8, 3, 8, 3, 2, 4, "128x192_abl4_cvt"),     // tuning: MFMA + conversions
    MIXQ_WO(8, 3, 8, 3, 2, 5, "128x192_abl5_noramp"),  // tuning: cfg 0 with all LOOK stages requested at once
#endif
};
constexpr int NUM_WO_PICK = 6;
// M <= 32: the 32 x 64 tiling (one 64-channel weight panel per workgroup, 5 k-steps of weights in flight per wave): 11.7 us against
// 20.6 us for gemm_w8a16_skinny_kernel at 32 x 4096 -> 11008, 11.1 vs 11.5 us at 4096 -> 4096 (profiles/r02_decode.txt)
constexpr int WO_SMALL = 8;
constexpr int NUM_WO = sizeof(g_wo) / sizeof(g_wo[0]);
MixqDevInt g_wo_forced_dev(-1);                      // per device (common.h)

inline int wo_cdiv(int a, int b) { return (a + b - 1) / b; }

// time ~ rounds over the 256 CUs x (k-steps x time per k-step of one tile + fixed); per k-step a wave issues 2 MB WNB MFMAs of 16
// cycles, the fixed part grows with the tile's output bytes
int pick_wo(int M, int N, int K) {
    // few tiles (narrow layer, small batch): 32-row tiles when 64-row ones would leave half the CUs idle (as in gemm_wreg.hip;
    // 64 x 4096 -> 4096: 11.6 vs 13.2 us, 128 x 4096 -> 4096: 12.3 vs 13.7 us)
    if (wo_cdiv(M, 32) * wo_cdiv(N, 64) <= 256 && wo_cdiv(M, 64) * wo_cdiv(N, 64) <= 128) return WO_SMALL;
    double best = 1e30; int bi = 0;
    const int nk = K >> 6;
    for (int c = 0; c < NUM_WO_PICK; ++c) {
        const WoConfig& g = g_wo[c];
        const int tiles = wo_cdiv(M, g.mb * 16) * wo_cdiv(N, g.wnb * 64);
        const double tk = 2.0 * g.mb * g.wnb * 16 / 2000.0 * 1.15 + 0.02, fixed = 3.0 + g.mb * g.wnb * 0.2;      // us
        const double t = wo_cdiv(tiles, 256) * (nk * tk + fixed);
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
    const int g_wo_forced = g_wo_forced_dev.get();
    WoArgs a;
    a.x = x; a.w = w_packed; a.sw = scale_col; a.bias = bias; a.y = y;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldy = ldy; a.tiles_m = a.tiles_n = 0; a.wblocks = (N + 15) >> 4;
    if (g_wo_forced == NUM_WO) {                                                 // the in-workgroup K-split form: explicit only (slower, see WO_SMALL)
        if (M > 32) return MIXQ_EINVAL;
        hipLaunchKernelGGL(gemm_w8a16_skinny_kernel, dim3((N + 31) / 32), dim3(WSK * 64), 0, mixq_stream(stream), a);
        return mixq_launch_status();
    }
    const int c = (g_wo_forced >= 0 && g_wo_forced < NUM_WO) ? g_wo_forced : (M <= 32 ? WO_SMALL : pick_wo(M, N, K));
    const WoConfig& g = g_wo[c];
    const int bm = g.mb * 16, bn = g.wnb * 64;
    a.tiles_m = wo_cdiv(M, bm); a.tiles_n = wo_cdiv(N, bn);
    const size_t ring = static_cast<size_t>(g.nstage) * g.mb * 2048, stg = static_cast<size_t>(bm) * (bn * 2 + 16);
    const size_t shm = ring > stg ? ring : stg;
    if (int rc = mixq_ensure_dynamic_lds(reinterpret_cast<const void*>(g.k), shm)) return rc;
    hipLaunchKernelGGL(g.k, dim3(a.tiles_m * a.tiles_n), dim3((WO_CW + g.loaders) * 64), shm, mixq_stream(stream), a);
    return mixq_launch_status();
}

extern "C" int mixq_gemm_w8a16_set_config(int cfg) {
    if (cfg < -1 || cfg > NUM_WO) return MIXQ_EINVAL;          // NUM_WO = the small-batch kernel
    g_wo_forced_dev.set(cfg);
    return MIXQ_OK;
}
extern "C" int mixq_gemm_w8a16_num_configs(void) { return NUM_WO + 1; }
extern "C" int mixq_gemm_w8a16_config_name(int cfg, char* buf, int cap) {
    if (cfg < 0 || cfg > NUM_WO || !buf || cap <= 0) return MIXQ_EINVAL;
    snprintf(buf, cap, "%s", cfg == NUM_WO ? "w8a16_decode32" : g_wo[cfg].name);
    return MIXQ_OK;
}
