// (iii)+(iv): int8 x int8 -> int32 MFMA GEMM for gfx950 with the MixQ epilogue fused in.
//
//   Y[m,n] = fp16( act( (sum_k Xq[m,k] Wq[n,k]) * sx[m] * sw[n]  +  sum_j Xo[m,j] Wo[n,j]  + addend[m,n] ) + bias[n] )
//
// Design (MI355X-first; every choice below was measured on an MI355X, numbers in DESIGN.md §6):
//  * Operands are K-contiguous.  A k-step moves a (BN + BM) x 64-byte slab HBM/L2 -> LDS with 16-byte LDS-DMA
//    (global_load_lds_dwordx4, no VGPR round trip) into an NSTAGE-deep ring.  The per-CU feed rate, not HBM, is
//    what limits this shape: 64-byte row segments of a plain [N,K] matrix stream at ~54 GB/s per CU, contiguous
//    KiB pieces at ~90-104 GB/s.  So the fast path takes operands in the tile-major "P16x64" layout
//    (include/mixq_hip.h) whose 16-row x 64-byte blocks ARE the LDS image; plain row-major operands (the
//    reference's layout) are still accepted, with the XOR swizzle applied on the DMA source address instead.
//  * LDS rows are 64 bytes; the 16-byte chunk index is XOR-ed with (-(row>>2) & 3) so the ds_read_b128 fragment reads
//    of every 16-lane group hit 16 distinct bank slots (SQ_LDS_BANK_CONFLICT = 0 measured).
//  * Wave specialisation: LOADERS dedicated wave(s) do nothing but issue the DMA and publish stages
//    (counted s_waitcnt vmcnt + the k-step's single s_barrier); the WAVES_M x WAVES_N consumer waves only read
//    fragments and issue MFMAs.  With the DMA issued from the MFMA waves the two costs ADD (an LDS-DMA issue
//    blocks its wave while the texture addresser is busy, and the MFMAs queued behind it wait): 418 ns per k-step
//    against 247 ns MFMA-only and 230 ns DMA-only.  LOADERS = 0 keeps that self-issuing form for comparison.
//  * v_mfma_i32_32x32x32_i8 with the WEIGHT rows as the A operand and the ACTIVATION rows as the B operand, so a
//    lane ends up holding 4 consecutive output columns n of one token m (one x_scale per lane).  Fragment sets
//    are double-buffered in registers: the reads of sub-step s+1 are issued between the MFMAs of sub-step s.
//  * The outlier correction runs as fp16 MFMA tail iterations (v_mfma_f32_32x32x16_f16) on the SAME accumulator
//    registers after they were dequantised to fp32 in place - no M x N side matrix is ever materialised
//    (the reference writes and re-reads one: linear.py:248-256).  The outlier count may live in device memory.
//  * Epilogue through LDS: the fp16 tile is staged in the (now idle) ring and written as whole 16-byte-per-lane
//    row segments; the direct per-lane 8-byte stores touched 32 rows per instruction and cost 5 us of a 30 us
//    kernel.
//  * W4A4: CDNA4 has no int4 MFMA; nibbles are expanded in registers to int8 values 16*v (a shift and a mask per
//    dword), the factor 256 is removed exactly in the fp32 epilogue.
//  * The grid is 1-D over output tiles with an XCD-aware remap: the tiles that share a weight panel run on the
//    same XCD (same L2).  The tile shape is picked per problem so the tile count fills the 256 CUs.
#include "common.h"
#include "gemm_sk.h"
#include "gemm_skinny.h"
#include "gemm_wreg.h"
#include <stdio.h>
#include <string.h>
#include <type_traits>

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
    int x_packed, w_packed;                           // operand stored in the P16x64 tile-major layout
    int xrows16, wrows16;                             // rows rounded up to 16 (packed operands)
    int gm;                                           // M tiles per group of the tile order (as in gemm_wreg.hip)
    int w_f16;                                        // the (packed) weight operand is stored in MIXQ_FMT_F16X64, not P16X64
    unsigned long long* trace;                        // diagnostics: 2 x 8 timestamps per workgroup, or null
};

constexpr int BKB = 64;                               // bytes of K per stage and row

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

__device__ __forceinline__ void glds16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// physical 16-byte chunk of logical chunk c in row r (an involution in c for fixed r)
__device__ __forceinline__ int swz(int r, int c) { return c ^ ((0 - (r >> 2)) & 3); }

__device__ __forceinline__ float silu(float v) { return mixq_silu(v); }   // (common.h: never contracted with the bias addition)

// BM: activation rows per tile, BN: weight rows per tile, WAVES_M x WAVES_N consumer waves, NSTAGE ring depth,
// MODE 0: int8 fused epilogue, 1: int4 fused epilogue, 2: int8 raw int32 output.  LOADERS: dedicated DMA waves
// (0 = the consumer waves issue the DMA themselves).  ABL (tuning only): 0 normal, 1 DMA only, 2 no DMA,
// 3 MFMA only, 5 no epilogue stores.
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int MODE, int LOADERS, int ABL>
__global__ __launch_bounds__((WAVES_M * WAVES_N + LOADERS) * 64) void gemm_kernel(const GemmArgs a)
{
    constexpr int CW = WAVES_M * WAVES_N;              // consumer waves
    constexpr int NT = (CW + LOADERS) * 64;
    constexpr int CH = BKB / 16;                       // 16-byte chunks per row
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int TI = (BM + BN) * CH / 64;            // 1-KiB DMA pieces per stage (weight rows, then activation rows)
    constexpr int IW = LOADERS ? LOADERS : CW;         // waves that issue DMA
    constexpr int LOADS = (TI + IW - 1) / IW;          // pieces per issuing wave and stage (a short wave repeats a piece)
    constexpr int STAGE_BYTES = (BM + BN) * BKB;
    constexpr int LOOK = NSTAGE - 2;                   // stages in flight ahead of the one being computed
    constexpr int NEWER = LOOK - 1;                    // stages younger than the one a wait retires
    constexpr bool I4 = (MODE == 1);
    constexpr int OPITCH = BN * 2 + 16;                // bytes per row of the fp16 output staging tile
    // Outlier-tail operands staged through the idle ring by the loader waves (epilogue): needs loaders and (BM + BN) x 256 bytes, and is
    // compiled into the int4 kernels only.  W4A4 layers carry 128 outlier columns (fp_features_num, linear.py:123-143): as global
    // fragment loads that tail costs 4.4 us at the metric tile, staged 2.6 us.  A short tail (W8A8's 41 columns) is faster from the
    // register ring whose first k-steps are requested before the dequantisation (1.8 vs 2.9 us), and both forms in one kernel cost 98
    // spilled registers - so the int8 kernels keep the ring, the int4 kernels stage (a short int4 tail pays ~1 us for it).
    constexpr bool TAIL_LDS = LOADERS > 0 && MODE == 1 && (BM + BN) * 256 <= NSTAGE * STAGE_BYTES;
    static_assert((BM + BN) * CH % 64 == 0, "stage must be a whole number of 1-KiB DMA pieces");
    static_assert(BM % 16 == 0 && BN % 16 == 0, "tiles are made of 16-row blocks");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
    static_assert(LOOK >= 1 && LOADS * NEWER < 64, "vmcnt range");
    static_assert(MODE == 2 || BM * OPITCH <= NSTAGE * STAGE_BYTES, "output staging tile must fit in the ring");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    // ---- tile id: XCD-aware remap (block b runs on XCD b % 8) so consecutive logical tiles share an L2 -----
    const int ntiles = a.tiles_m * a.tiles_n;
    int tile;
    {
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, x = b & 7, s = b >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;          // bijective for any ntiles
    }
    // groups of gm M tiles, M fastest inside a group, then N (gemm_wreg.hip explains why: at prefill sizes the CUs of an XCD then share
    // a few weight panels AND a few activation slabs instead of one panel and 32 slabs)
    int tm, tn;
    {
        const int per_group = a.gm * a.tiles_n, grp = tile / per_group, first_m = grp * a.gm;
        const int gsz = a.tiles_m - first_m < a.gm ? a.tiles_m - first_m : a.gm;
        const int r = tile - grp * per_group;
        tn = r / gsz; tm = first_m + (r - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int nk = a.KB / BKB;
    auto stamp = [&](int slot) {                         // diagnostics only (mixq_gemm_set_trace; tools build)
#ifdef MIXQ_TUNING
        if (a.trace && tid == 0) {
            a.trace[blockIdx.x * 16 + slot] = wall_clock64();
            a.trace[blockIdx.x * 16 + 8 + slot] = __builtin_readcyclecounter();   // s_memtime
        }
#else
        (void)slot;
#endif
    };
    stamp(0);
    int n_out_dev_v = 0;                                 // requested now, consumed by the epilogue
    if constexpr (MODE != 2) { if (a.n_out_dev) n_out_dev_v = *a.n_out_dev; }

    // ---- DMA piece table of an issuing wave ------------------------------------------------------------------
    // Plain [R,KB] operands: a 1-KiB piece is 16 rows x 64 B with the XOR swizzle on the SOURCE chunk.
    // Packed P16x64 operands: the memory image of a 16-row block IS the (swizzled) LDS image: a piece is one
    // contiguous KiB, a stage is one contiguous run per operand, and the k-step stride is rows16 * 64.
    const uint8_t* nsrc[LOADS];                          // source of the NEXT stage to issue, advanced per stage
    int kstr[LOADS];
    int piece[LOADS];
    auto dma_setup = [&](int iw) {                       // iw: index of this wave among the issuing waves
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            int pw = i * IW + iw;
            if (pw >= TI) pw = iw;                       // benign duplicate of this wave's first piece
            piece[i] = pw;
            const int qd = pw * 64 + lane;
            const int rr = qd / CH, pc = qd % CH;
            const bool is_w = rr < BN;
            const int r = is_w ? rr : rr - BN;
            const int row0 = is_w ? n0 : m0, rows = is_w ? a.N : a.M, rows16 = is_w ? a.wrows16 : a.xrows16;
            const uint8_t* base = is_w ? a.qw : a.qx;
            const bool packed = is_w ? a.w_packed : a.x_packed;
            if (packed) {
                int rb = (row0 + r) >> 4; rb = rb < (rows16 >> 4) ? rb : (rows16 >> 4) - 1;
                // P16X64: the memory image of a block IS the LDS image (row r, physical chunk pc at r*64 + pc*16).  F16X64 (the one
                // weight image an operator keeps, fragment order c*256 + r*16): the same bytes sit at logical chunk c = swz(r, pc) -
                // 16-byte pieces of one KiB block, still whole cache lines per wave instruction - and land in the same LDS image
                if (is_w && a.w_f16) nsrc[i] = base + static_cast<size_t>(rb) * 1024 + swz(r & 15, pc) * 256 + (r & 15) * 16;
                else                 nsrc[i] = base + static_cast<size_t>(rb) * 1024 + (r & 15) * 64 + pc * 16;
                kstr[i] = rows16 * 64;
            } else {
                int gr = row0 + r; gr = gr < rows ? gr : rows - 1;
                nsrc[i] = base + static_cast<size_t>(gr) * a.KB + swz(r, pc) * 16;
                kstr[i] = BKB;
            }
        }
    };
    auto stage = [&](int buf) {                          // issue the next stage into ring slot `buf`
        uint8_t* base = lds + buf * STAGE_BYTES;
        if constexpr (ABL != 2 && ABL != 3) {
#pragma unroll
            for (int i = 0; i < LOADS; ++i) { glds16(nsrc[i], base + piece[i] * 1024); nsrc[i] += kstr[i]; }
        }
    };

    // =================================================================================================================
    // loader wave(s): issue stage kt+LOOK, retire stage kt+1, meet the consumers at the k-step's barrier
    // =================================================================================================================
    if constexpr (LOADERS > 0) {
        if (wave >= CW) {
            __builtin_amdgcn_s_setprio(2);               // the DMA issue must never queue behind MFMA waves
            dma_setup(wave - CW);
#pragma unroll
            for (int s = 0; s < LOOK; ++s)
                if (s < nk) stage(s);
            if (NEWER < nk) wait_vmcnt<LOADS * NEWER>(); else wait_vmcnt<0>();       // stage 0 landed
            __builtin_amdgcn_s_barrier();
            int nxt = LOOK % NSTAGE, kt = 0;
            for (; kt + LOOK < nk; ++kt) {               // steady state: branch-free body
                stage(nxt);
                wait_vmcnt<LOADS * NEWER>();             // stage kt+1 landed, NEWER younger stages still in flight
                if constexpr (ABL != 8) __builtin_amdgcn_s_barrier();
                nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
            }
            for (; kt + 1 < nk; ++kt) {                  // drain
                wait_vmcnt<0>();
                if constexpr (ABL != 8) __builtin_amdgcn_s_barrier();
            }
            if constexpr (MODE != 2) {                   // take part in the epilogue's barriers and its stores
                __builtin_amdgcn_s_barrier();            // every wave is done reading the ring
                if constexpr (TAIL_LDS) {
                    // The fp16 outlier tail's operands through LDS: as global fragment loads they are 32-byte pieces of 2 n-byte rows
                    // (one cache line per lane pair, every tail k-step an exposed round trip: 4.4 us at 128 columns, the W4A4 case);
                    // here the loader waves copy the tile's X_out and W_out rows - up to 128 columns = 256 contiguous bytes per
                    // row and pass - into the idle ring with LDS-DMA (chunk c of row r at position c ^ (r & 15): conflict-free
                    // fragment reads) while the consumers dequantise, and the tail MFMAs run out of LDS.
                    int n_out_l = a.n_out;
                    if (a.n_out_dev) { const int nd = *a.n_out_dev; n_out_l = nd < n_out_l ? nd : n_out_l; }
                    if (!a.xo || !a.wo || ABL == 6) n_out_l = 0;
                    if (n_out_l > 0) {
                        const int kpad_l = (n_out_l + 15) & ~15;
                        const int lw = wave - CW;
                        for (int pass0 = 0; pass0 < n_out_l; pass0 += 128) {
                            if (pass0 > 0) __builtin_amdgcn_s_barrier();           // the previous pass has been consumed
                            for (int q = lw; q < (BM + BN) / 4; q += LOADERS) {
                                const int rr = 4 * q + (lane >> 4);
                                int col = pass0 + (((lane & 15) ^ (rr & 15)) << 3);
                                col = col < kpad_l ? col : 0;                      // past the readable width: any valid address (masked later)
                                const uint16_t* src;
                                if (rr < BM) { int m = m0 + rr; m = m < a.M ? m : a.M - 1; src = a.xo + static_cast<size_t>(m) * a.ldxo + col; }
                                else         { int n = n0 + rr - BM; n = n < a.N ? n : a.N - 1; src = a.wo + static_cast<size_t>(n) * a.ldwo + col; }
                                glds16(reinterpret_cast<const uint8_t*>(src), lds + q * 1024);
                            }
                            wait_vmcnt<0>();
                            __builtin_amdgcn_s_barrier();                          // this pass has landed
                        }
                        __builtin_amdgcn_s_barrier();                              // tail reads done: the staging tile may overwrite them
                    }
                }
                __builtin_amdgcn_s_barrier();            // staging tile complete
            }
        }
    }

    // =================================================================================================================
    // consumer waves
    // =================================================================================================================
    const int wn = wave % WAVES_N, wm = (wave / WAVES_N) % WAVES_M;
    const int lr = lane & 31, lh = lane >> 5;
    int wrow[NI], xrow[MI];
#pragma unroll
    for (int i = 0; i < NI; ++i) wrow[i] = wn * WN + i * 32 + lr;
#pragma unroll
    for (int j = 0; j < MI; ++j) xrow[j] = wm * WM + j * 32 + lr;

    i32x16 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    u32x2 swp[NI][4];                                    // epilogue scales, prefetched near the end of the k loop
    uint16_t sxh[MI];
    if (LOADERS == 0 || wave < CW) {
        if constexpr (LOADERS == 0) dma_setup(wave);

        // Fragment sets are double-buffered in registers: set p holds the operands of one 32-byte sub-step.
        i32x4 wf[2][NI], xf[2][MI];
        if constexpr (ABL != 0 && ABL != 5) {           // ablation builds: give never-loaded sets defined, opaque values
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int i = 0; i < NI; ++i) { wf[p][i] = i32x4{lane, lane, lane, lane}; asm volatile("" : "+v"(wf[p][i])); }
#pragma unroll
                for (int j = 0; j < MI; ++j) { xf[p][j] = i32x4{lane, 1, lane, 1}; asm volatile("" : "+v"(xf[p][j])); }
            }
        }
        // One fragment read, in the order the MFMAs of a sub-step consume them: w0, x0..x(MI-1), w1..w(NI-1).
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
        // One region = the MFMAs of fragment set P with (a) ND_ reads of the other set and (b) NG_ DMA pieces issued
        // in the gaps behind the MFMAs, in exactly this program order (sched_barrier(0) pins it): an MFMA keeps the
        // matrix pipe busy for 32 cycles while the wave issues the next LDS reads (and DMA pieces when LOADERS = 0).
        // W4A4: 16 packed bytes = 32 int4.  Even columns (low nibbles) and odd columns (high nibbles) become two int8
        // fragments holding 16*v (A and B split identically, so the k pairing is preserved).  The expansion of the set
        // read in this region is spread behind this region's MFMAs (3 VALU per dword) so it never stands in front of them.
        i32x4 cwl[2][I4 ? NI : 1], cwh[2][I4 ? NI : 1], cxl[2][I4 ? MI : 1], cxh[2][I4 ? MI : 1];
        if constexpr (I4 && ABL != 0 && ABL != 5) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int i = 0; i < NI; ++i) { cwl[p][i] = cwh[p][i] = i32x4{lane, lane, lane, lane}; asm volatile("" : "+v"(cwl[p][i]), "+v"(cwh[p][i])); }
#pragma unroll
                for (int j = 0; j < MI; ++j) { cxl[p][j] = cxh[p][j] = i32x4{lane, 1, lane, 1}; asm volatile("" : "+v"(cxl[p][j]), "+v"(cxh[p][j])); }
            }
        }
        uint32_t nib = 0xf0f0f0f0u;                          // kept in an SGPR: a literal would double every v_and's size
        if constexpr (I4) asm volatile("s_mov_b32 %0, 0xf0f0f0f0" : "=s"(nib));
        auto convert_dword = [&](int p, int idx, int d) {     // idx in the order of load_one
            if constexpr (I4) {
                if (idx == 0 || idx > MI) {
                    const int i = idx == 0 ? 0 : idx - MI;
                    const uint32_t v = static_cast<uint32_t>(wf[p][i][d]);
                    cwl[p][i][d] = static_cast<int>((v << 4) & nib);
                    cwh[p][i][d] = static_cast<int>(v & nib);
                } else {
                    const int j = idx - 1;
                    const uint32_t v = static_cast<uint32_t>(xf[p][j][d]);
                    cxl[p][j][d] = static_cast<int>((v << 4) & nib);
                    cxh[p][j][d] = static_cast<int>(v & nib);
                }
            }
        };
        auto region = [&](auto p_c, auto nd_c, auto ng_c, int rbuf, int rsub, int dma_buf) {
            constexpr int P = decltype(p_c)::value, ND_ = decltype(nd_c)::value, NG_ = decltype(ng_c)::value;
            constexpr int NM = NI * MI, SL = NM > 1 ? NM - 1 : 1;
            uint8_t* dbase = lds + dma_buf * STAGE_BYTES;
            if constexpr (!I4) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int i = m / MI, j = m % MI;
                    if constexpr (ABL != 1)
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[P][i], xf[P][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m < SL) {
                        if constexpr (ABL != 2 && ABL != 3 && NG_ > 0) {
#pragma unroll
                            for (int g = (NG_ * m) / SL; g < (NG_ * (m + 1)) / SL; ++g) { glds16(nsrc[g], dbase + piece[g] * 1024); nsrc[g] += kstr[g]; }
                        }
                        if constexpr (ABL != 1 && ABL != 3) {
#pragma unroll
                            for (int d = (ND_ * m) / SL; d < (ND_ * (m + 1)) / SL; ++d) load_one(1 - P, rbuf, rsub, d);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // 2*NM MFMA slots: one LDS read behind each of the first ND_ MFMAs, the DMA pieces with them, then the
                // dword-wise expansion of what was just read, spread over the remaining slots (CS = first expansion slot).
                constexpr int NS2 = 2 * NM, CS = (NS2 > 5) ? 4 : NS2 - 1, NCV = ND_ * 4;
#pragma unroll
                for (int mm = 0; mm < NS2; ++mm) {
                    const int m = mm % NM, i = m / MI, j = m % MI;     // all low-nibble MFMAs, then all high-nibble ones:
                    if constexpr (ABL != 1) {                         // consecutive MFMAs never share an accumulator
                        if (mm < NM) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cwl[P][i], cxl[P][j], acc[i][j], 0, 0, 0);
                        else               acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cwh[P][i], cxh[P][j], acc[i][j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ABL != 2 && ABL != 3 && NG_ > 0) {
                        if (mm < NS2 - 1) {
#pragma unroll
                            for (int g = (NG_ * mm) / (NS2 - 1); g < (NG_ * (mm + 1)) / (NS2 - 1); ++g) { glds16(nsrc[g], dbase + piece[g] * 1024); nsrc[g] += kstr[g]; }
                        }
                    }
                    if constexpr (ABL != 1 && ABL != 3) {
                        if (mm < ND_) load_one(1 - P, rbuf, rsub, mm);
                        if (mm >= CS && NCV > 0) {
#pragma unroll
                            for (int g = (NCV * (mm - CS)) / (NS2 - CS); g < (NCV * (mm - CS + 1)) / (NS2 - CS); ++g) convert_dword(1 - P, g >> 2, g & 3);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using IND = std::integral_constant<int, NI + MI>;
        using ING = std::integral_constant<int, LOADERS ? 0 : LOADS>;

        // ---- software-pipelined k loop -------------------------------------------------------------------------
        // LOOK stages in flight.  The DMA of stage kt+LOOK goes into the ring slot last read in iteration kt-2 (every
        // wave finished those reads before the barrier of iteration kt-1), so it needs no barrier of its own.  The
        // single barrier of a k-step sits in front of the first fragment read of the NEXT stage, between two MFMA groups.
        if constexpr (LOADERS == 0) {
#pragma unroll
            for (int s = 0; s < LOOK; ++s)
                if (s < nk) stage(s);
            if (NEWER < nk) wait_vmcnt<LOADS * NEWER>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        stamp(1);
        if constexpr (ABL != 1 && ABL != 3) {
#pragma unroll
            for (int d = 0; d < NI + MI; ++d) load_one(0, 0, 0, d);
            if constexpr (I4) {
#pragma unroll
                for (int g = 0; g < (NI + MI) * 4; ++g) convert_dword(0, g >> 2, g & 3);
            }
        }
        int cur = 0, nxt = LOOK % NSTAGE;                // ring slots of stage kt and stage kt+LOOK
        // one k-step; ISSUE: DMA stage kt+LOOK from this wave, NEXT: a stage kt+1 exists.
        auto body = [&](auto issue_c, auto next_c) {
            constexpr bool ISSUE = decltype(issue_c)::value, NEXT = decltype(next_c)::value;
            const int cur1 = (cur + 1 == NSTAGE) ? 0 : cur + 1;
            if constexpr (ISSUE) region(I0{}, IND{}, ING{}, cur, 1, nxt);     // MFMA(set 0), read set 1, DMA
            else                 region(I0{}, IND{}, I0{}, cur, 1, nxt);
            if constexpr (NEXT) {
                if constexpr (LOADERS == 0) {
                    if constexpr (ISSUE) wait_vmcnt<LOADS * NEWER>();
                    else                 wait_vmcnt<0>();
                }
                if constexpr (ABL != 8) __builtin_amdgcn_s_barrier();
                region(I1{}, IND{}, I0{}, cur1, 0, nxt);                       // MFMA(set 1), first reads of stage kt+1
            } else {
                region(I1{}, I0{}, I0{}, cur1, 0, nxt);
            }
            cur = cur1;
            nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
        };
        int kt = 0;
        if constexpr (LOADERS == 0) {
            for (; kt + LOOK < nk; ++kt) body(std::true_type{}, std::true_type{});
        } else {
            for (; kt + 4 < nk; ++kt) body(std::false_type{}, std::true_type{});
        }
        // The epilogue's scales are requested a few k-steps before the loop ends, so their HBM latency hides under the
        // remaining MFMAs instead of standing between the last MFMA and the first store.
        if constexpr (MODE != 2) {
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int m = m0 + xrow[j];
                sxh[j] = a.sx[m < a.M ? m : a.M - 1];     // rows past M are computed but never stored
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {             // N % 4 == 0: a group of 4 columns is all in or all out
                    const int n = n0 + wn * WN + i * 32 + 4 * lh + 8 * g;
                    swp[i][g] = *reinterpret_cast<const u32x2_u*>(a.sw + (n < a.N ? n : a.N - 4));
                }
        }
        for (; kt + 1 < nk; ++kt) body(std::false_type{}, std::true_type{});
        body(std::false_type{}, std::false_type{});
        stamp(2);
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    if constexpr (MODE == 2) {
        if (wave >= CW) return;
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
        // Rows of Y are written from an fp16 staging tile in LDS when whole 16-byte chunks are addressable.
        const bool staged = ((a.N & 7) == 0) && ((a.ldy & 7) == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
        if (LOADERS == 0 || wave < CW) {
            int n_out = a.n_out;
            if (a.n_out_dev) n_out = n_out_dev_v < n_out ? n_out_dev_v : n_out;
            if (!a.xo || !a.wo) n_out = 0;
            const int ksteps = ABL == 6 ? 0 : (n_out + 15) >> 4;
            constexpr float PRE = I4 ? (1.f / 256.f) : 1.f;

            float sxv[MI];
#pragma unroll
            for (int j = 0; j < MI; ++j) sxv[j] = h2f(sxh[j]) * PRE;
            const bool has_add = ABL != 6 && a.addend != nullptr, has_bias = ABL != 6 && a.bias != nullptr, do_silu = ABL != 6 && a.act != MIXQ_ACT_NONE;
            const bool mul_add = a.act == MIXQ_ACT_SILU_MUL;                   // addend multiplies after the SiLU instead of adding before it
            auto unpack4 = [](u32x2 v, float* o) {
                o[0] = h2f(static_cast<uint16_t>(v.x & 0xffffu)); o[1] = h2f(static_cast<uint16_t>(v.x >> 16));
                o[2] = h2f(static_cast<uint16_t>(v.y & 0xffffu)); o[3] = h2f(static_cast<uint16_t>(v.y >> 16));
            };
            // fp16 outlier tail: the operands of TD k-steps are in flight at once (a ring of TD register sets: k-step 0 is
            // requested before the barrier, 1..TD-1 after the dequantisation, set d again for k-step kk+TD right after its
            // MFMAs), so the tail exposes one memory round trip per TD k-steps instead of one per k-step.  Small tiles
            // have few MFMAs per k-step to hide a round trip behind and registers to spare: deeper ring.
            constexpr int REG_CAP = 512 / ((CW + LOADERS + 3) / 4);        // VGPRs per wave at this workgroup's occupancy
            constexpr int TD_FIT = (REG_CAP - NI * MI * 16 - 100) / ((NI + MI) * 4);
            constexpr int TD = TD_FIT < 2 ? 2 : (TD_FIT > 8 ? 8 : TD_FIT);
            u32x4 xoq[TD][MI], woq[TD][NI];
            auto tail_load = [&](int P, int kk) {            // P: a constant after unrolling
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    int xr = m0 + xrow[j]; xr = xr < a.M ? xr : a.M - 1;
                    xoq[P][j] = *reinterpret_cast<const u32x4*>(a.xo + static_cast<size_t>(xr) * a.ldxo + lh * 8 + kk * 16);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    int wr = n0 + wrow[i]; wr = wr < a.N ? wr : a.N - 1;
                    woq[P][i] = *reinterpret_cast<const u32x4*>(a.wo + static_cast<size_t>(wr) * a.ldwo + lh * 8 + kk * 16);
                }
            };
            if constexpr (!TAIL_LDS) { if (ksteps > 0) tail_load(0, 0); }                      // the rest of the ring: after the dequantisation
            if (LOADERS > 0 || staged) __builtin_amdgcn_s_barrier();          // every wave is done reading the ring
            stamp(6);

            // dequantise in place: f = acc * x_scale[m] * scale_col[n]
            f32x16 fa[NI][MI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float swv[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) unpack4(swp[i][g], swv + 4 * g);
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) fa[i][j][r] = static_cast<float>(acc[i][j][r]) * sxv[j] * swv[r];
            }
            auto tail_mma = [&](int P, int kk) {
                const int kb = kk * 16 + lh * 8;                     // mask columns >= n_out (the pad may hold anything)
                if (kb + 8 > n_out) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        uint32_t keep = 0;
                        if (kb + 2 * d < n_out)     keep |= 0x0000ffffu;
                        if (kb + 2 * d + 1 < n_out) keep |= 0xffff0000u;
#pragma unroll
                        for (int j = 0; j < MI; ++j) xoq[P][j][d] &= keep;
#pragma unroll
                        for (int i = 0; i < NI; ++i) woq[P][i][d] &= keep;
                    }
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
                        fa[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, woq[P][i]),
                                                                          __builtin_bit_cast(f16x8, xoq[P][j]), fa[i][j], 0, 0, 0);
            };
            if constexpr (TAIL_LDS) {
              if (ksteps > 0) {
                auto tail_lds_load = [&](int P, int kl) {    // tail k-step kl (16 columns) of the current pass: chunk 2 kl + lh of a row
#pragma unroll
                    for (int j = 0; j < MI; ++j)
                        xoq[P][j] = *reinterpret_cast<const u32x4*>(lds + xrow[j] * 256 + (((2 * kl + lh) ^ (xrow[j] & 15)) << 4));
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        woq[P][i] = *reinterpret_cast<const u32x4*>(lds + (BM + wrow[i]) * 256 + (((2 * kl + lh) ^ ((BM + wrow[i]) & 15)) << 4));
                };
                for (int pass0 = 0; pass0 < n_out; pass0 += 128) {
                    if (pass0 > 0) __builtin_amdgcn_s_barrier();               // (the loaders refill only after everybody has read the previous pass)
                    __builtin_amdgcn_s_barrier();                              // this pass has landed
                    const int k0 = pass0 >> 4;
                    const int kn = (ksteps - k0) < 8 ? (ksteps - k0) : 8;
                    tail_lds_load(0, 0);
                    for (int kl = 0; kl < kn; kl += 2) {
                        if (kl + 1 < kn) tail_lds_load(1, kl + 1);
                        tail_mma(0, k0 + kl);
                        if (kl + 1 < kn) {
                            if (kl + 2 < kn) tail_lds_load(0, kl + 2);
                            tail_mma(1, k0 + kl + 1);
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                                  // tail reads done: the staging tile may overwrite them
              }
            } else {
#pragma unroll
                for (int d = 1; d < TD; ++d)
                    if (d < ksteps) tail_load(d, d);
                for (int kk0 = 0; kk0 < ksteps; kk0 += TD) {
#pragma unroll
                    for (int d = 0; d < TD; ++d) {
                        const int kk = kk0 + d;
                        if (kk < ksteps) {
                            tail_mma(d, kk);
                            if (kk + TD < ksteps) tail_load(d, kk + TD);
                        }
                    }
                }
            }

            // Optional terms, fp16 rounding and the store.  Specialised on (staged, any optional term) so the common
            // case is straight-line code (a taken branch costs a fetch bubble that nothing hides with one wave per SIMD);
            // columns come in all-in/all-out groups of 4 (N % 4 == 0), and out-of-range rows/columns read clamped
            // addresses and are dropped at the store.
            auto finish_tile = [&](auto staged_c, auto opt_c) {
                constexpr bool ST = decltype(staged_c)::value, OPT = decltype(opt_c)::value;
                // the addend / multiplier of block (i, j+1) is requested while block (i, j) is computed: one exposed round trip
                u32x2 aq[4];
                auto add_load = [&](int i, int j) {
                    const int m = m0 + xrow[j];
                    const uint16_t* ap = a.addend + static_cast<size_t>(m < a.M ? m : a.M - 1) * a.lda;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + wn * WN + i * 32 + 4 * lh + 8 * g;
                        aq[g] = *reinterpret_cast<const u32x2_u*>(ap + (n < a.N ? n : a.N - 4));
                    }
                };
                if (OPT && has_add) add_load(0, 0);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int nloc = wn * WN + i * 32 + 4 * lh;
                    const int nb = n0 + nloc;
                    float bv[16];
                    if (OPT && has_bias) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = nb + 8 * g;
                            unpack4(*reinterpret_cast<const u32x2_u*>(a.bias + (n < a.N ? n : a.N - 4)), bv + 4 * g);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < MI; ++j) {
                        f32x16 f = fa[i][j];
                        const int m = m0 + xrow[j];
                        float avv[16];
                        if (OPT && has_add) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) unpack4(aq[g], avv + 4 * g);
                            if (j + 1 < MI) add_load(i, j + 1);
                            else if (i + 1 < NI) add_load(i + 1, 0);
                            if (!mul_add) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) f[r] += avv[r];
                            }
                        }
                        if (OPT && do_silu) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) f[r] = silu(f[r]);
                        }
                        if (OPT && has_bias) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) f[r] += bv[r];
                        }
                        if (OPT && mul_add) {                  // (silu(z) + bias) * up: linear.py:372-373, then mlp.py:61
#pragma unroll
                            for (int r = 0; r < 16; ++r) f[r] *= avv[r];
                        }
                        if (ABL != 5 || a.act == 77) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                u32x2 o;
                                o.x = static_cast<uint32_t>(f2h(f[4 * g])) | (static_cast<uint32_t>(f2h(f[4 * g + 1])) << 16);
                                o.y = static_cast<uint32_t>(f2h(f[4 * g + 2])) | (static_cast<uint32_t>(f2h(f[4 * g + 3])) << 16);
                                if constexpr (ST) {
                                    *reinterpret_cast<u32x2*>(lds + xrow[j] * OPITCH + (nloc + 8 * g) * 2) = o;
                                } else {
                                    if (m < a.M && nb + 8 * g < a.N)
                                        *reinterpret_cast<u32x2_u*>(a.y + static_cast<size_t>(m) * a.ldy + nb + 8 * g) = o;
                                }
                            }
                        }
                    }
                }
            };
            const bool opt = has_add || has_bias || do_silu;
            if (staged) { if (opt) finish_tile(std::true_type{}, std::true_type{}); else finish_tile(std::true_type{}, std::false_type{}); }
            else        { if (opt) finish_tile(std::false_type{}, std::true_type{}); else finish_tile(std::false_type{}, std::false_type{}); }
            stamp(7);
            // ds_write is asynchronous and a raw s_barrier does not wait for it: the tile writes must have been
            // performed before another wave's copy-out reads are issued
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (LOADERS > 0 || staged) __builtin_amdgcn_s_barrier();          // staging tile complete
            stamp(3);
        }
        if (staged) {
            // all waves (loader included): 16 bytes per lane, 16 consecutive lanes cover a 256-byte row segment
            constexpr int CPR = BN * 2 / 16;                                   // 16-byte chunks per tile row
            for (int q = tid; q < BM * CPR; q += NT) {
                const int r = q / CPR, c = q - r * CPR;
                const int m = m0 + r, n = n0 + c * 8;
                if (m < a.M && n < a.N && (ABL != 5 || a.act == 77)) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * OPITCH + c * 16);
                    // streaming (nt) store: Y is written once and not re-read by this kernel; keeping 11 MB of dirty
                    // lines in the L2s only postpones their write-back to the kernel boundary
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(a.y + static_cast<size_t>(m) * a.ldy + n));
                }
            }
        }
#ifdef MIXQ_TUNING
        if (a.trace) {
            stamp(4);
            wait_vmcnt<0>();
            stamp(5);
        }
#endif
    }
}

// Standalone epilogue of the unfused pair (mixlib.gemm + mixlib.dequantizeInt8[Silu]) - the route the UNCHANGED reference takes
// on gfx950, where torch reports capability major 9 (linear.py:86,234-241).  HBM-bound: 8 outputs per thread, 2 x 16-byte
// int32 loads, one 16-byte fp16 store (VEC), or one element per thread for shapes that do not allow it.
template <bool VEC>
__global__ __launch_bounds__(256) void dequant_kernel(const int32_t* __restrict__ y32, int ldy32, const uint16_t* __restrict__ sx,
                                                      const uint16_t* __restrict__ sw, const uint16_t* __restrict__ addend, int lda,
                                                      const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int ldy,
                                                      int M, int N, int act)
{
    constexpr int E = VEC ? 8 : 1;
    const int per_row = N / E;
    const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= static_cast<long long>(M) * per_row) return;
    const int m = static_cast<int>(t / per_row), n = static_cast<int>(t % per_row) * E;
    const float sxv = h2f(sx[m]);
    float v[E];
    if constexpr (VEC) {
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(y32 + static_cast<size_t>(m) * ldy32 + n);
        const i32x4 a1 = *reinterpret_cast<const i32x4*>(y32 + static_cast<size_t>(m) * ldy32 + n + 4);
        const u32x4 s = *reinterpret_cast<const u32x4*>(sw + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e]     = static_cast<float>(a0[e]) * sxv * h2f(static_cast<uint16_t>(e & 1 ? s[e >> 1] >> 16 : s[e >> 1] & 0xffffu));
            v[4 + e] = static_cast<float>(a1[e]) * sxv * h2f(static_cast<uint16_t>(e & 1 ? s[2 + (e >> 1)] >> 16 : s[2 + (e >> 1)] & 0xffffu));
        }
        if (addend) {
            const u32x4 ad = *reinterpret_cast<const u32x4*>(addend + static_cast<size_t>(m) * lda + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += h2f(static_cast<uint16_t>(e & 1 ? ad[e >> 1] >> 16 : ad[e >> 1] & 0xffffu));
        }
    } else {
        v[0] = static_cast<float>(y32[static_cast<size_t>(m) * ldy32 + n]) * sxv * h2f(sw[n]);
        if (addend) v[0] += h2f(addend[static_cast<size_t>(m) * lda + n]);
    }
    if (act == MIXQ_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = silu(v[e]);
    }
    if (bias) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] += h2f(bias[n + e]);
    }
    if constexpr (VEC) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = static_cast<uint32_t>(f2h(v[2 * e])) | (static_cast<uint32_t>(f2h(v[2 * e + 1])) << 16);
        *reinterpret_cast<u32x4*>(y + static_cast<size_t>(m) * ldy + n) = o;
    } else {
        y[static_cast<size_t>(m) * ldy + n] = f2h(v[0]);
    }
}

// ---- configuration table ---------------------------------------------------------------------------------------
struct GemmConfig {
    const char* name;
    int bm, bn, waves, nstage;
    void (*k8)(const GemmArgs);
    void (*k4)(const GemmArgs);
    void (*k32)(const GemmArgs);
};

#define MIXQ_CFG(BM, BN, WMv, WNv, NS, LD)                                                                    \
    { #BM "x" #BN "_w" #WMv "x" #WNv "_s" #NS "_l" #LD, BM, BN, (WMv) * (WNv) + (LD), NS,                   \
      gemm_kernel<BM, BN, WMv, WNv, NS, 0, LD, 0>, gemm_kernel<BM, BN, WMv, WNv, NS, 1, LD, 0>,             \
      gemm_kernel<BM, BN, WMv, WNv, NS, 2, LD, 0> }
#define MIXQ_ABL(BM, BN, WMv, WNv, NS, LD, ABL)                                                               \
    { #BM "x" #BN "_w" #WMv "x" #WNv "_s" #NS "_l" #LD "_abl" #ABL, BM, BN, (WMv) * (WNv) + (LD), NS,       \
      gemm_kernel<BM, BN, WMv, WNv, NS, 0, LD, ABL>, gemm_kernel<BM, BN, WMv, WNv, NS, 1, LD, ABL>,         \
      gemm_kernel<BM, BN, WMv, WNv, NS, 2, LD, ABL> }

const GemmConfig g_cfgs[] = {
    MIXQ_CFG(256, 128, 4, 2, 5, 1),     // 0: 8 consumer waves (64x64) + 1 loader
    MIXQ_CFG(256, 128, 4, 2, 5, 0),     // 1: same tile, consumers issue the DMA themselves
    MIXQ_CFG(256, 128, 2, 2, 5, 1),     // 2: 4 consumer waves (128x64) + 1 loader
    MIXQ_CFG(256, 128, 2, 2, 5, 4),     // 3: 4 consumers (128x64) + 4 loaders
    MIXQ_CFG(256, 128, 4, 2, 5, 2),     // 4: 8 consumers + 2 loaders
    MIXQ_CFG(256, 128, 2, 2, 5, 2),     // 5: 4 consumers + 2 loaders
    MIXQ_CFG(256, 128, 4, 2, 5, 4),     // 6: 8 consumers + 4 loaders
    MIXQ_CFG(256, 128, 2, 2, 5, 0),     // 7: 4 consumers self-issuing
    MIXQ_CFG(128, 192, 2, 2, 5, 4),     // 8: 4 consumers 64 x 96 + 4 loaders
    MIXQ_CFG(128, 192, 2, 2, 5, 2),     // 9: ... + 2 loaders
    MIXQ_CFG(128, 128, 2, 2, 5, 2),     // 10
    MIXQ_CFG(64, 64, 2, 2, 5, 1),       // 11
    MIXQ_CFG(32, 128, 1, 4, 5, 1),      // 12: small-M
    MIXQ_CFG(256, 256, 4, 2, 5, 0),     // 13: 8 consumers 64(M) x 128(N) self-issuing
    MIXQ_CFG(128, 256, 2, 2, 5, 2),     // 14: 4 consumers 64 x 128 + 2 loaders
    MIXQ_CFG(128, 256, 2, 2, 5, 4),     // 15
    MIXQ_CFG(64, 128, 2, 2, 5, 2),      // 16: 4 consumers 32 x 64 + 2 loaders: N = 4096 at M = 512 is exactly 256 such tiles
    MIXQ_CFG(64, 128, 2, 2, 5, 4),      // 17
    MIXQ_CFG(64, 192, 2, 2, 5, 4),      // 18: 4 consumers 32 x 96
#ifdef MIXQ_TUNING                      // ablation forms (results are garbage by design): only in the tools build (make tuning)
    MIXQ_ABL(128, 192, 2, 2, 5, 4, 1),  // cfg 8, DMA only
    MIXQ_ABL(128, 192, 2, 2, 5, 4, 2),  // cfg 8, no DMA
    MIXQ_ABL(128, 192, 2, 2, 5, 4, 3),  // cfg 8, MFMA only
    MIXQ_ABL(128, 192, 2, 2, 5, 4, 6),  // cfg 8, epilogue without the optional terms compiled in (code size probe)
    MIXQ_ABL(128, 192, 2, 2, 5, 4, 8),  // cfg 8 without the k-loop barriers (timing probe)
#endif
};
constexpr int NUM_CFGS = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

MixqDevInt g_forced_cfg(-1);                       // per device (common.h): -1 = automatic
#ifdef MIXQ_TUNING
unsigned long long* g_trace = nullptr;             // diagnostics: see mixq_gemm_set_trace (tools build only)
#else
constexpr unsigned long long* g_trace = nullptr;
#endif

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Shape-aware choice: estimated time ~ rounds over 256 CUs x max(MFMA time, feed time) per k-step of a tile.
//   MFMA: BM*BN*64*2 ops at 8192 ops/clk/CU;  feed: (BM+BN)*64 bytes at ~48 B/clk/CU (packed) / 25 (plain rows).
// Shape-aware choice.  Score = rounds over the 256 CUs x measured time per k-step of one tile (microseconds, MI355X,
// packed operands, tools/sweep_gemm.py over the Llama-2-7b/70b and Llama-3-8B shapes).  The winner per
// shape in those sweeps is what this picks: whichever tile makes close to 256 tiles wins - 64x128 at N = 4096, 64x192 at
// N = 6144, 128x128 at N = 8192, 128x192 around N = 10-12k, 128x256 at N = 14k, 256x256 when there are >= ~200 such tiles.  Row-strided (plain) operands feed slower but rank the same.
struct PickEntry { int cfg; float tk; };
const PickEntry g_pick[] = {{0, 0.60f}, {1, 0.39f}, {2, 0.60f}, {3, 0.40f}, {4, 0.40f}, {5, 0.40f}, {6, 0.37f}, {7, 0.40f}, {8, 0.31f},
                            {9, 0.33f}, {10, 0.26f}, {11, 0.17f}, {12, 0.15f}, {13, 0.60f}, {14, 0.40f}, {15, 0.36f}, {17, 0.165f}, {18, 0.21f}};
constexpr int NUM_PICK = sizeof(g_pick) / sizeof(g_pick[0]);
int pick_config(int M, int N, int KB, bool packed) {
    (void)KB; (void)packed;
    double best = 1e30; int bi = 0;
    for (int pi = 0; pi < NUM_PICK; ++pi) {
        const int c = g_pick[pi].cfg;
        const GemmConfig& g = g_cfgs[c];
        const int tiles = cdiv(M, g.bm) * cdiv(N, g.bn);
        // the 64x64 tile (40 KB of LDS) runs two or more workgroups per CU; its per-k-step figure is for that regime
        const int rounds = cdiv(tiles, (g.bm == 64 && g.bn == 64) ? 512 : 256);
        // rows of a tile beyond M are wasted MFMA work but cost the same time: no correction needed; a tile much
        // taller than M (small-batch decode) just wastes LDS traffic, which the per-k-step numbers already contain
        const double t = rounds * static_cast<double>(g_pick[pi].tk);
        if (t < best * 0.999) { best = t; bi = c; }
    }
    return bi;
}

// Stream-K (gemm_sk.hip) pays when the data-parallel tiling leaves most CUs idle AND K is long: measured on MI355X
// (tools/sweep_gemm.py), M = 512: 4096x11008 46.1 -> 39.3 us, 4096x14336 54.9 -> 43.4 us with sk128x128; at K = 4096
// (4096x4096: 21.3 vs 22.7 us) and whenever the tiling already fills the chip (11008x4096: 30 vs 43 us) the partial-tile
// hand-off (~8 us per workgroup) costs more than the idle CUs.  Returns a gemm_sk.hip config id or -1.
// Not chosen automatically any more: since the data-parallel epilogue was rewritten (straight-line dequant, burst-prefetched
// outlier tail) the 64x64 / 128x128 tilings match stream-K without outlier columns (35.4 vs 35.8 us at 512 x 11008 -> 4096)
// and beat it with the 1 % outlier columns of the real workload, whose tail stream-K still pays per 32x32 block.  The
// kernels stay selectable through mixq_gemm_set_config and stay under test.
int g_auto_sk = 0;
int pick_stream_k(int M, int N, int KB) {
    if (!g_auto_sk || M < 128 || N < 128) return -1;
    const int nk = KB / 64;
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    if (tiles > 160 || nk < 128) return -1;
    return 3;                                            // sk128x128_w2x2_s5
}

// (Rounds 3-4 preferred this file's 256 x 256 tiling - cfg LDS256 - at prefill sizes; since the weights-in-registers kernels got the
// panel epilogue, 128 x 256 is ahead of it at every prefill shape measured, by 2.7-13 %: profiles/r05_prefill_sweep.txt.  The tiling stays
// selectable by configuration and reads the fragment-order weight image through a remapped DMA source.)
// (cfg 13: 256x256_w4x2_s5_l0)

int launch_gemm(GemmArgs& a, int mode, hipStream_t st, int cfg = -1) {
    const bool packed = a.x_packed && a.w_packed;
    const int forced = g_forced_cfg.get();
    const int c = cfg >= 0 ? cfg : ((forced >= 0 && forced < NUM_CFGS) ? forced : pick_config(a.M, a.N, a.KB, packed));
    const GemmConfig* g = &g_cfgs[c];
    a.tiles_m = cdiv(a.M, g->bm);
    a.tiles_n = cdiv(a.N, g->bn);
    a.gm = a.tiles_m <= 8 ? a.tiles_m : 8;
    a.trace = g_trace;
    void (*k)(const GemmArgs) = mode == 0 ? g->k8 : (mode == 1 ? g->k4 : g->k32);
    const size_t shm = static_cast<size_t>(g->bm + g->bn) * BKB * g->nstage;
    if (int rc = mixq_ensure_dynamic_lds(reinterpret_cast<const void*>(k), shm)) return rc;
    hipLaunchKernelGGL(k, dim3(a.tiles_m * a.tiles_n), dim3(g->waves * 64), shm, st, a);
    return mixq_launch_status();
}

int gemm_fused_common(const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                      const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                      const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int K,
                      int act, int layout, int bit, mixq_stream_t stream, uint32_t* row_amax = nullptr, const uint32_t* amax_mask = nullptr)
{
    const bool f6 = layout == MIXQ_XW_F6X128;                    // both operands as FP6 codes: the W4A4 form on the FP6 matrix pipe
    if (f6) { if (bit != 4) return MIXQ_EINVAL; layout = 0; }
    if (layout & ~(MIXQ_X_PACKED | MIXQ_W_PACKED | MIXQ_W_F16X64)) return MIXQ_EINVAL;
    if ((layout & MIXQ_W_PACKED) && (layout & MIXQ_W_F16X64)) return MIXQ_EINVAL;
    const bool wf16 = layout & MIXQ_W_F16X64;
    if (wf16 && !(layout & MIXQ_X_PACKED)) return MIXQ_EINVAL;   // fragment-order weights go with P16X64 activations
    if (M < 0 || N < 0 || K <= 0 || n_out < 0) return MIXQ_EINVAL;
    if (M > 0 && N > 0 && (!q_x || !q_w || !x_scale || !scale_col || !y)) return MIXQ_EINVAL;
    if (act != MIXQ_ACT_NONE && act != MIXQ_ACT_SILU && act != MIXQ_ACT_SILU_MUL && act != MIXQ_ACT_SILU_PAIR) return MIXQ_EINVAL;
    if (act == MIXQ_ACT_SILU_MUL && !addend) return MIXQ_EINVAL;            // the multiplier is mandatory
    const bool pair = act == MIXQ_ACT_SILU_PAIR;                            // interleaved gate / up rows: y has N / 2 columns
    if (pair && addend) return MIXQ_EINVAL;
    if (pair && (!((bit == 8 && wf16) || f6) || (N & 15))) return MIXQ_ESHAPE;   // (the weights-in-registers kernels: int8 fragment-order weights, or FP6 codes)
    const int KB = bit == 8 ? K : K / 2;
    if ((KB % 64) || (bit == 4 && (K & 1)) || (N & 3) || (ldy & 3) || ldy < (pair ? N / 2 : N)) return MIXQ_ESHAPE;
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
    a.x_packed = (layout & MIXQ_X_PACKED) ? 1 : 0; a.w_packed = (layout & MIXQ_W_PACKED) ? 1 : 0;
    a.xrows16 = (M + 15) & ~15; a.wrows16 = (N + 15) & ~15;
    const int sk_num = mixq_sk_num_configs();
    const int skinny_id = NUM_CFGS + sk_num, wr0 = skinny_id + 1;
    const int g_forced = g_forced_cfg.get();                     // this device's forced configuration (-1: automatic)
    if (f6) {
        if (row_amax) return MIXQ_ESHAPE;
        if (g_forced >= 0 && g_forced < wr0) return MIXQ_EINVAL;                // only the weights-in-registers kernels read this format
        const int c = g_forced >= wr0 ? g_forced - wr0 : (pair ? mixq_wr_pick_pair(6, M, N, KB) : mixq_wr_pick(6, M, N, KB));
        if (pair && !mixq_wr_has_pair(6, c)) return MIXQ_EINVAL;
        int n1 = 0, c2 = -1;
        if (g_forced == -1 && !pair && mixq_wr_split(6, M, N, KB, c, &n1, &c2)) {   // (a partial last round of tiles: two launches, as for int8 below)
            if (int rc = mixq_wr_launch(c, 6, q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                                        ldy, M, N, KB, act, g_trace, mixq_stream(stream), nullptr, nullptr, 0, n1)) return rc;
            return mixq_wr_launch(c2, 6, q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                                  ldy, M, N, KB, act, nullptr, mixq_stream(stream), nullptr, nullptr, n1, N - n1);
        }
        return mixq_wr_launch(c, 6, q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                              ldy, M, N, KB, act, g_trace, mixq_stream(stream), nullptr, nullptr);
    }
    // small-batch form (gemm_skinny.hip): M <= 32, packed operands (either packed layout): a weight stream, no LDS staging
    {
        const bool both_packed = a.x_packed && (a.w_packed || wf16);
        // (wide layers with fragment-order int8 weights: the 32 x 64 weights-in-registers tiling streams faster, gemm_wreg.hip)
        const bool wide_wr = wf16 && bit == 8 && N >= 8192 && g_forced < 0;
        if (!pair && !wide_wr && (g_forced < 0 || g_forced == skinny_id) && mixq_skinny_applies(bit, M, N, KB, both_packed, both_packed)) {
            if (row_amax) return MIXQ_ESHAPE;                    // the row-maximum side output lives in the weights-in-registers kernels
            return mixq_skinny_launch(bit, q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                                      ldy, M, N, KB, act, wf16 ? 1 : 0, mixq_stream(stream));
        }
        if (g_forced == skinny_id && !pair) return MIXQ_EINVAL;
    }
    // fragment-order weights: the weights-in-registers kernels (gemm_wreg.hip); a forced tiling of this file reads the same F16X64 weight
    // image through a remapped DMA source
    if (wf16) {
        int c;
        if (pair) {
            c = g_forced >= wr0 ? g_forced - wr0 : mixq_wr_pick_pair(8, M, N, KB);
            if (!mixq_wr_has_pair(8, c)) return MIXQ_EINVAL;                // (a forced tiling without the paired epilogue)
        }
        else if (g_forced >= wr0) c = g_forced - wr0;
        else if (g_forced >= NUM_CFGS) return MIXQ_EINVAL;               // a stream-K form was forced: it takes P16X64 weights only
        else if (g_forced >= 0) {                                          // a tiling of this file, forced: it reads the fragment-order image too
            if (row_amax) return MIXQ_ESHAPE;
            a.w_packed = 1; a.w_f16 = 1;
            return launch_gemm(a, bit == 8 ? 0 : 1, mixq_stream(stream), g_forced);
        }
        else if (bit == 8 && !row_amax && mixq_wr_ksplit_pays(M, N, KB)) c = mixq_wr_ksplit_config();   // long K, few tiles: two workgroups per tile
        else {
            c = mixq_wr_pick(bit, M, N, KB);
            // a partial last round of tiles: the full rounds' columns with this tiling, the rest with the cheapest one, two launches
            int n1 = 0, c2 = -1;
            if (g_forced == -1 && bit == 8 && mixq_wr_split(8, M, N, KB, c, &n1, &c2)) {
                if (int rc = mixq_wr_launch(c, bit, q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                                            ldy, M, N, KB, act, g_trace, mixq_stream(stream), row_amax, amax_mask, 0, n1)) return rc;
                return mixq_wr_launch(c2, bit, q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                                      ldy, M, N, KB, act, nullptr, mixq_stream(stream), row_amax, amax_mask, n1, N - n1);
            }
        }
        return mixq_wr_launch(c, bit, q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                              ldy, M, N, KB, act, g_trace, mixq_stream(stream), row_amax, amax_mask);
    }
    if (row_amax) return MIXQ_ESHAPE;                            // (only the fragment-order-weights route computes it)
    if (g_forced >= wr0) return MIXQ_EINVAL;
    // stream-K form (gemm_sk.hip): packed operands, workspace registered, chosen explicitly or by the shape rule
    if (a.x_packed && a.w_packed && act != MIXQ_ACT_SILU_MUL) {       // (the stream-K epilogue has no multiplier form)
        int sk = -1;
        if (g_forced >= NUM_CFGS) sk = g_forced - NUM_CFGS;
        else if (g_forced < 0) sk = pick_stream_k(M, N, KB);
        if (sk >= 0 && mixq_sk_usable(sk))
            return mixq_sk_launch(sk, bit, q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda,
                                  bias, y, ldy, M, N, KB, act, mixq_stream(stream));
    }
    return launch_gemm(a, bit == 8 ? 0 : 1, mixq_stream(stream));
}

}  // namespace

extern "C" int mixq_gemm_i8_fused(const int8_t* q_x, const int8_t* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                                  const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out,
                                  const int32_t* n_out_dev, const uint16_t* addend, int lda, const uint16_t* bias,
                                  uint16_t* y, int ldy, int M, int N, int K, int act, int layout, mixq_stream_t stream)
{
    return gemm_fused_common(q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                             ldy, M, N, K, act, layout, 8, stream);
}

// The same GEMM that also leaves, for the quantiser of the NEXT layer, the per-row maximum of |y| over the columns that layer does not
// treat as outliers (include/mixq_hip.h).  int8, fragment-order weights, batches the weights-in-registers kernels serve.
extern "C" int mixq_gemm_i8_fused_amax(const int8_t* q_x, const int8_t* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                                       const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out,
                                       const int32_t* n_out_dev, const uint16_t* addend, int lda, const uint16_t* bias,
                                       uint16_t* y, int ldy, int M, int N, int K, int act, int layout, uint32_t* row_amax,
                                       const uint32_t* col_mask, mixq_stream_t stream)
{
    if (!row_amax) return MIXQ_EINVAL;
    if (!(layout & MIXQ_W_F16X64)) return MIXQ_ESHAPE;
    return gemm_fused_common(q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                             ldy, M, N, K, act, layout, 8, stream, row_amax, col_mask);
}
// 1 when mixq_gemm_i8_fused_amax serves (M, N, K) with this layout, else 0 (the caller then quantises the next layer the usual way)
extern "C" int mixq_gemm_amax_supported(int M, int N, int K, int layout)
{
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || !(layout & MIXQ_W_F16X64) || !(layout & MIXQ_X_PACKED)) return 0;
    const bool wide_wr = N >= 8192;
    if (!wide_wr && mixq_skinny_applies(8, M, N, K, true, true)) return 0;
    return 1;
}

extern "C" int mixq_gemm_i4_fused(const uint8_t* q_x, const uint8_t* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                                  const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out,
                                  const int32_t* n_out_dev, const uint16_t* addend, int lda, const uint16_t* bias,
                                  uint16_t* y, int ldy, int M, int N, int K, int act, int layout, mixq_stream_t stream)
{
    return gemm_fused_common(q_x, q_w, x_scale, scale_col, x_out, ldxo, w_out, ldwo, n_out, n_out_dev, addend, lda, bias, y,
                             ldy, M, N, K, act, layout, 4, stream);
}

extern "C" int mixq_gemm_i8(const int8_t* q_x, const int8_t* q_w, int32_t* y32, int ldy, int M, int N, int K,
                            mixq_stream_t stream)
{
    if (M < 0 || N < 0 || K <= 0 || (M > 0 && N > 0 && (!q_x || !q_w || !y32))) return MIXQ_EINVAL;
    if ((K % 64) || (N & 3) || (ldy & 3) || ldy < N) return MIXQ_ESHAPE;
    if (M == 0 || N == 0) return MIXQ_OK;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.qx = reinterpret_cast<const uint8_t*>(q_x); a.qw = reinterpret_cast<const uint8_t*>(q_w);
    a.y32 = y32; a.M = M; a.N = N; a.KB = K; a.ldy = ldy;
    a.xrows16 = (M + 15) & ~15; a.wrows16 = (N + 15) & ~15;
    return launch_gemm(a, 2, mixq_stream(stream));
}

extern "C" int mixq_dequant(const int32_t* y32, int ldy32, const uint16_t* x_scale, const uint16_t* scale_col,
                            const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int act,
                            mixq_stream_t stream)
{
    if (!y32 || !x_scale || !scale_col || !y || M < 0 || N < 0 || ldy32 < N || ldy < N) return MIXQ_EINVAL;
    if (act != MIXQ_ACT_NONE && act != MIXQ_ACT_SILU) return MIXQ_EINVAL;
    if (M == 0 || N == 0) return MIXQ_OK;
    const bool vec = (N % 8 == 0) && (ldy32 % 4 == 0) && (ldy % 8 == 0) && (!addend || lda % 8 == 0 || lda == 0) &&
                     ((reinterpret_cast<uintptr_t>(y32) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scale_col) |
                       reinterpret_cast<uintptr_t>(addend)) & 15) == 0;
    const long long total = static_cast<long long>(M) * (vec ? N / 8 : N);
    const dim3 g(static_cast<unsigned>((total + 255) / 256));
    if (vec) hipLaunchKernelGGL(dequant_kernel<true>, g, dim3(256), 0, mixq_stream(stream), y32, ldy32, x_scale, scale_col, addend, lda,
                                bias, y, ldy, M, N, act);
    else     hipLaunchKernelGGL(dequant_kernel<false>, g, dim3(256), 0, mixq_stream(stream), y32, ldy32, x_scale, scale_col, addend,
                                lda, bias, y, ldy, M, N, act);
    return mixq_launch_status();
}

// Config ids: [0, NUM_CFGS) data-parallel tilings of this file, then the stream-K forms, "decode32", then the
// weights-in-registers tilings of gemm_wreg.hip (MIXQ_FMT_F16X64 operands only).
static int total_configs() { return NUM_CFGS + mixq_sk_num_configs() + 1 + mixq_wr_num_configs(); }

// ---- per-device registry: the hand-off workspace (stream-K / pairwise split-K: tuning build) and the CU count -------------------------
namespace {
constexpr int MIXQ_MAX_DEV = 64;
MixqDevState g_dev_state[MIXQ_MAX_DEV] = {};
}
MixqDevState* mixq_dev_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MIXQ_MAX_DEV) { (void)hipGetLastError(); return nullptr; }
    return &g_dev_state[dev];
}
int mixq_num_cus() {
    MixqDevState* d = mixq_dev_state();
    if (!d) return 256;
    if (d->num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount <= 0) { (void)hipGetLastError(); return 256; }
        d->num_cu = p.multiProcessorCount;
    }
    return d->num_cu;
}
bool mixq_ws_get(void** ws, size_t* bytes, size_t* flag_bytes) {
    MixqDevState* d = mixq_dev_state();
    if (!d || !d->ws) return false;
    *ws = d->ws; *bytes = d->bytes; *flag_bytes = MIXQ_WS_FLAG_BYTES;
    return true;
}
#ifdef MIXQ_TUNING
extern "C" int mixq_gemm_set_fuse_probe(const void* x, void* scratch, void* counters, int K)    // (timing probe, tuning library only: include/mixq_hip.h)
{
    mixq_wr_fuse_probe(x, scratch, counters, K);
    return MIXQ_OK;
}
extern "C" int mixq_gemm_hint_next_weights(const void* w, long long bytes)    // (experiment, tuning library only: include/mixq_hip.h)
{
    if (bytes < 0 || (bytes > 0 && !w) || (reinterpret_cast<size_t>(w) & 127)) return MIXQ_EINVAL;
    mixq_wr_hint_next(w, bytes);
    return MIXQ_OK;
}
#endif
// (the product library hands no tile through memory: the registration is accepted so that callers of the C ABI need not care which build they link)
extern "C" int mixq_gemm_set_workspace(void* ws, long long bytes)
{
    if (bytes < 0 || (bytes > 0 && !ws)) return MIXQ_EINVAL;
    MixqDevState* d = mixq_dev_state();                  // the workspace belongs to the device that is current at registration
    if (!d) return MIXQ_ENODEV;
    d->ws = bytes ? ws : nullptr;
    d->bytes = static_cast<size_t>(bytes);
    return MIXQ_OK;
}
extern "C" long long mixq_gemm_workspace_bytes(void) { return mixq_sk_workspace_bytes(); }   // (0 in the product library)
extern "C" int mixq_gemm_set_config(int cfg) {
    if (cfg < -2 || cfg >= total_configs()) return MIXQ_EINVAL;            // (-1: automatic; -2: automatic without the N split - the A/B partner of tools/prefill_sweep.py)
    g_forced_cfg.set(cfg);
    return MIXQ_OK;
}
#ifdef MIXQ_TUNING
extern "C" int mixq_gemm_set_trace(unsigned long long* buf) { g_trace = buf; return MIXQ_OK; }
extern "C" int mixq_gemm_set_krot(int v) { return mixq_wr_set_krot(v); }
#endif                                             // (product build: no in-kernel stamps, no tile-order knob - and no entry points for them)
extern "C" int mixq_gemm_num_configs(void) { return total_configs(); }
extern "C" int mixq_gemm_config_name(int cfg, char* buf, int cap) {
    if (cfg < 0 || cfg >= total_configs() || !buf || cap <= 0) return MIXQ_EINVAL;
    const int sk0 = NUM_CFGS, dec = sk0 + mixq_sk_num_configs(), wr0 = dec + 1;
    snprintf(buf, cap, "%s", cfg < sk0 ? g_cfgs[cfg].name : (cfg < dec ? mixq_sk_config_name(cfg - sk0)
                                                                      : (cfg == dec ? "decode32" : mixq_wr_config_name(cfg - wr0))));
    return MIXQ_OK;
}
extern "C" int mixq_gemm_pick_config(int M, int N, int K, int bit) {
    if (M <= 0 || N <= 0 || K <= 0 || (bit != 4 && bit != 8)) return MIXQ_EINVAL;
    const int KB = bit == 8 ? K : K / 2;
    const int sk = pick_stream_k(M, N, KB);
    if (sk >= 0 && mixq_sk_usable(sk)) return NUM_CFGS + sk;
    return pick_config(M, N, KB, true);
}
// The config the automatic choice runs for weights in the packed format `fmt` (MIXQ_FMT_P16X64 / MIXQ_FMT_F16X64; the
// activations are P16X64 either way).
extern "C" int mixq_gemm_pick_split(int M, int N, int K, int bit, int fmt, int* n1, int* cfg2) {
    if ((bit != 8 && bit != 4) || M <= 0 || N <= 0 || K <= 0 || !n1 || !cfg2) return MIXQ_EINVAL;
    const bool f6 = fmt == MIXQ_FMT_F6X128 && bit == 4;
    if (!f6 && !(fmt == MIXQ_FMT_F16X64 && bit == 8)) return 0;
    const int KB = bit == 8 ? K : K / 2, wbit = f6 ? 6 : 8;
    if (!f6 && ((N < 8192 && mixq_skinny_applies(8, M, N, KB, true, true)) || mixq_wr_ksplit_pays(M, N, KB))) return 0;
    int c2 = -1;
    if (!mixq_wr_split(wbit, M, N, KB, mixq_wr_pick(wbit, M, N, KB), n1, &c2)) return 0;
    *cfg2 = NUM_CFGS + mixq_sk_num_configs() + 1 + c2;
    return 1;
}
extern "C" int mixq_gemm_pick_config_fmt(int M, int N, int K, int bit, int fmt) {
    if (M <= 0 || N <= 0 || K <= 0 || (bit != 4 && bit != 8)) return MIXQ_EINVAL;
    const int KB = bit == 8 ? K : K / 2;
    const int dec = NUM_CFGS + mixq_sk_num_configs();
    if (fmt == MIXQ_FMT_F6X128) return bit == 4 ? dec + 1 + mixq_wr_pick(6, M, N, KB) : MIXQ_EINVAL;   // (FP6-coded operands: the weights-in-registers kernels only)
    const bool wide_wr = fmt == MIXQ_FMT_F16X64 && bit == 8 && N >= 8192;                 // as in gemm_fused_common
    if (!wide_wr && fmt != MIXQ_FMT_PLAIN && mixq_skinny_applies(bit, M, N, KB, true, true)) return dec;
    if (fmt == MIXQ_FMT_F16X64) return dec + 1 + ((bit == 8 && mixq_wr_ksplit_pays(M, N, KB)) ? mixq_wr_ksplit_config() : mixq_wr_pick(bit, M, N, KB));
    return mixq_gemm_pick_config(M, N, K, bit);
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
