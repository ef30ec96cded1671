// (iii)+(iv) again, second structure: the WEIGHT operand never touches LDS.
//
//   Y[m,n] = fp16( act( (sum_k Xq[m,k] Wq[n,k]) * sx[m] * sw[n]  +  sum_j Xo[m,j] Wo[n,j]  + addend[m,n] ) + bias[n] )
//
// Why (round-1 measurements): in gemm.hip both operands go HBM/L2 -> LDS by LDS-DMA and come back by
// ds_read_b128.  DMA writes drain into the LDS at ~50 B/clk and fragment reads at 256 B/clk through the same port, and
// the two ADD: a 128 x 192 tile costs (128+192)*64/50 + 4*(64+96)*64/256 = 570 LDS cycles per 64-byte k-step against 384
// MFMA cycles, so the loop ran at 70 % MFMA occupancy and, everything being busy at once, at 1.7 GHz.
// Here:
//   * the 4 consumer waves of a workgroup sit side by side along N (1 x 4): each owns WN = 16*WNB weight rows nobody else
//     in the workgroup needs, so its weight fragments go global -> VGPR directly.  The operand layout MIXQ_FMT_F16X64
//     makes a fragment (16 rows x 64 bytes, lane l = row l&15, k-chunk l>>4: one v_mfma_i32_16x16x64_i8 A operand) ONE
//     contiguous KiB: a single fully coalesced global_load_dwordx4 with a wave-uniform base, D k-steps deep in a
//     register ring.  No barrier, no LDS, no duplication between waves for 60 % of the operand bytes.
//   * only the activation tile (16*MB rows, shared by the 4 waves) is staged: 1 KiB P16X64 blocks by LDS-DMA from dedicated
//     loader wave(s) into an NSTAGE ring, read back with ds_read_b128 (row l&15, chunk l>>4; the layout's swizzle makes the
//     16x16x64 fragment conflict-free).  X stays in P16X64 - a row's 64 bytes contiguous - because the quantise kernel
//     writes it row by row: fragment-order stores cost it 0.5-1.2 us (profiles/r02_quant_sweep.txt).  LDS traffic per k-step of the 128 x 192
//     tile drops from 20 KB written + 40 KB read to 8 KB written + 32 KB read (~290 LDS cycles < 384 MFMA cycles).
//   * a fragment of X is re-read in place right behind the last MFMA that used it (j-major MFMA order), so X needs MB
//     fragment registers, not 2 MB.
// Everything after the k loop is gemm.hip's epilogue restated for the 16x16 accumulator layout: in-place dequantisation,
// fp16 outlier tail on the same registers (v_mfma_f32_16x16x32_f16), optional addend / SiLU / bias / multiplier, fp16
// tile staged through LDS, 16-byte row stores.  Same arithmetic in the same order: results are bit-identical to gemm.hip.
// Reference call sites replaced: mixlib.int8FusedDequantize[Silu] / int4FusedDequantize[Silu] and the torch.mm + bias around
// them, /root/reference/mixquant/modules/linear.py:244-285, :330-373.
#include "common.h"
#include "gemm_wreg.h"
#include "gemm_sk.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

struct WrArgs {
    const uint8_t* qx;  const uint8_t* qw;            // [KB/64][blocks][1 KiB]: X in P16X64, W in F16X64
    const uint16_t* sx; const uint16_t* sw;
    const uint16_t* xo; const uint16_t* wo;
    const int32_t* n_out_dev;
    const uint16_t* addend; const uint16_t* bias;
    uint16_t* y;
    int M, N, KB;
    int ldxo, ldwo, n_out, lda, ldy, act;
    int tiles_m, tiles_n;
    int xblocks, wblocks;                             // 16-row blocks per k-step of each operand
    int gm;                                           // M tiles per group of the tile order (see the tile map in the kernel)
    int n_begin;                                      // first weight row (output column) of this launch: tiles_n tiles of BN rows from here (N split)
    // reciprocals for the tile map's two divisions (q = mulhi(n, m), exact for n d < 2^32, m = floor(2^32 / d) + 1; 0 stands for d = 1):
    // by gm x tiles_n, by gm, by the last group's size - three run-time divisions (~25 dependent scalar instructions each, through the
    // float reciprocal) stood in front of every wave's first operand request
    unsigned int mg_group, mg_gm, mg_last;
    unsigned long long* trace;
    // optional side output for the NEXT layer's quantiser (down_proj behind gate_proj): row_amax[m] = max over the columns n whose bit in
    // amax_mask is clear of |fp16 bits of y[m,n]| (atomic max of bit patterns: order-independent, exact); amax_mask: bit n of a uint32 array
    uint32_t* row_amax;
    const uint32_t* amax_mask;
    // pairwise split-K (ABL = 50): one int32 slot of BM x BN per tile and one flag word per tile (zero between launches)
    uint8_t* ks_slots;
    unsigned int* ks_flags;
#ifdef MIXQ_TUNING
    // cross-layer prefetch experiment (round 6, mixq_gemm_hint_next_weights; measured negative, see the loader): the NEXT launch's weight image
    const uint8_t* pf;
    unsigned int pf_lines;                            // 128-byte lines of it
    int pf_mode;                                      // bit 0: scalar-cache touches, bit 1: vector touches (one lane per line), bit 2: ... with the nt hint
    // one-launch probe (ABL 77, mixq_gemm_set_fuse_probe): a stand-in quantise phase in front of the GEMM - fp16 rows in, bytes out, a counter per M tile
    const uint16_t* fx; uint8_t* fq; unsigned int* fcnt; int fK;
    int fflags;                                       // (probe breakdown, env MIXQ_FUSE_FLAGS: 1 no stores, 2 no drain, 4 no atomic add, 8 no poll, 16 ordinary instead of write-through stores)
#endif
};

constexpr int WR_CW = 4;
// every lambda of the kernels below is force-inlined: at 256-row tiles their bodies are big enough for the inliner to leave them as
// functions, and a by-reference capture of the accumulator array behind a real call puts the whole frame in scratch memory
constexpr int MIXQ_XR = 4;                                                     // FP6 form: activation tuples held per wave (window)
#define MIXQ_INL __attribute__((always_inline))                              // consumer waves, 1 x 4 along N

template <int N> __device__ __forceinline__ void wr_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

__device__ __forceinline__ void wr_glds16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ float wr_silu(float v) { return mixq_silu(v); }   // (common.h: never contracted with the bias addition)

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N) (ring slots are compile-time registers)
template <int I, int N, class F> __device__ __forceinline__ void wr_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); wr_static_for<I + 1, N>(f); }
}

// MB: 16-row activation blocks per tile (BM = 16 MB), WNB: 16-row weight blocks per wave (BN = 64 WNB), NSTAGE: X ring depth,
// D: weight register ring depth (k-steps), Q: operand coding (0 int8; 1 nibble-packed int4, expanded to int8 in registers; 2 int4
// carried as FP6 codes on the FP6 matrix pipe, MIXQ_FMT_F6X128 operands, k-steps of 128 elements), LOADERS: DMA waves, ABL (tuning): 0 normal,
// 1 no weight loads, 2 no X traffic (no DMA, no LDS reads), 3 MFMA only, 4 weight loads issued but never waited for, 5 the loader
// never waits for its DMA, 6 no k-loop barriers (4-6: timing probes, results are garbage), 7 no stores of Y, 8 ordinary instead of
// nt stores, 9 return at entry, 12 no prologue ramp in the loader.
constexpr int MIXQ_Y_ST = 2;                // cache policy of the stores of Y (buffer-store aux bits): 2 = nt (streaming), 16 = sc1 (write-through), 18 = both, 0 = ordinary
constexpr int MIXQ_Y_WT_MB = 4;             // tiles of at most this many 16-row blocks store Y write-through (sc1 | nt): the layers they serve write a few MB of Y in ONE round of
                                            // tiles, and lines written through during the epilogue are not left for the kernel's closing release to write back: -0.1 ... -0.25 us per launch at
                                            // N = 4096 / small batches, nothing at the metric tile, +0.6 % at 2048 tokens if the big tiles did it too (NOTEBOOK.md round 5)
constexpr int MIXQ_PAIR_ST = 2;             // cache policy of the joint gate / up form's stores of Y: 2 = nt (streaming), 0 = ordinary
template <int MB, int WNB, int NSTAGE, int D, int Q, int LOADERS, int ABL>
__global__ __launch_bounds__((WR_CW + LOADERS) * 64) void gemm_wreg_kernel(const WrArgs a)
{
    // ABL = 90 is not an ablation but a kernel FORM - the paired gate / up epilogue (PAIR below) on the shipped loop: everywhere else in this
    // body the loop variant is ABLK, which reads 0 for it
    // (ABL 16 / 17 / 18: the feed ablations 3 / 2 / 1 with PSEUDO-RANDOM bytes in the operands that are never loaded - the MFMA's power, hence the clock of a
    // power-limited launch, depends on how its operands switch: lane-constant stand-ins flatter every ablation; profiles/r05_ablations_random_operands.txt)
    constexpr bool RNDOPS = ABL >= 16 && ABL <= 18;
    // ABL in [100, 228): the SHIPPED loop with probe features switched on bit by bit (tuning build; results are correct): FL bit 0 the FP6
    // form's activation window holds all MB tuples, bit 1 the loaders sleep 64 cycles behind every DMA piece, bit 2 the loaders sleep 192 cycles
    // in front of a stage's pieces, bit 3 loaders at priority 0, bit 4 consumers at priority 3, bit 5 the in-k-step timeline (as ABL 36),
    // bit 6 ... stamped by consumer wave 2 (a SIMD no loader wave shares) instead of wave 0
    constexpr int FL = (ABL >= 100 && ABL < 1124) ? ABL - 100 : 0;                 // (bit 7: the FP6 form's LDS image of rounds 3-5, see OLDSWZ)
    constexpr int ABLK = ABL == 90 ? 0 : (RNDOPS ? 19 - ABL : (ABL >= 100 ? 0 : ABL));
    constexpr bool TL = ABL == 36 || (FL & 32);
    constexpr bool WPROBE = ABLK == 76;                                            // (timing probe: the weights through LDS - see the loader's stage())
    constexpr int TLW = (FL & 64) ? 2 : 0;
    constexpr int CW = WR_CW;
    constexpr int NT = (CW + LOADERS) * 64;
    constexpr int BM = MB * 16, WN = WNB * 16, BN = CW * WN;
    // F6: the W4A4 form on the FP6 matrix pipe.  Every integer of [-8, 8] is an FP6 E3M2 value, products are integers <= 64 and
    // the fp32 accumulator is exact below 2^24, so v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales computes the int4
    // contraction bit-exactly (tools/ubench_fp6.hip: K up to 262144, worst-case operands) at 1.6x the MACs per cycle of
    // v_mfma_i32_16x16x64_i8 - gfx950 has no int4 MFMA, and the nibble form (Q = 1) pays the int8 rate plus the expansion.
    // Operand blocks are 16 rows x 128 elements = 1.5 KiB in fragment order (MIXQ_FMT_F6X128, include/mixq_hip.h): a lane's 24 bytes
    // are a 16-byte piece at block + 16 lane and an 8-byte piece at block + 1024 + 8 lane.
    // Q = 3 (F6R): the same with the weight ring held AS the 6-register operand tuples.  One load cannot fill a 192-bit tuple, so each
    // fragment's dwordx4 + dwordx2 are two outputs of one asm statement that are concatenated into the tuple right there - a pure
    // renaming IF the register coalescer assigns the outputs to the tuple's sub-registers (it does: no v_mov in the loop, checked on the
    // ISA by tools/check_wreg_asm.py and by the bit-exact tests - a copy at that point would read registers whose loads are in flight).
    // That removes the 9 v_mov_b64 + wait states in front of each k-step's first MFMA of the Q = 2 form.  Loads are unconditional
    // (requests past the end of K wrap around to the first k-steps: valid addresses, never consumed, drained behind the loop), so
    // every wait is a compile-time count.
    constexpr bool I4 = Q == 1, F6 = Q >= 2, F6R = Q == 3;
    // WRAP: every k-step requests weights unconditionally - past the end of K the requests wrap around to the first k-steps (valid
    // addresses, never consumed, drained behind the loop) - so the last D + NSLOT k-steps need no conditional loads and no run-time wait
    // tables.  The tuple-ring FP6 form and the int8 form of the shipped loop: -0.8 ... -3.2 % over the BASELINE shapes between two product
    // builds (-1.3 % at the metric shape; profiles/r03_ab_wrap_tail.txt).
    constexpr bool WRAP = F6R || ABLK == 41 || (Q == 0 && (ABLK == 0 || ABLK == 6 || (ABLK >= 60 && ABLK < 80)) && LOADERS != 0);
    constexpr int BLK = F6 ? 1536 : 1024;                // bytes of one 16-row operand block of one k-step
    constexpr int STAGE_BYTES = MB * BLK;
    constexpr int LOOK = NSTAGE - 2, NEWER = LOOK - 1;
    // SELF (LOADERS = 0, the prefill form): four FAT waves, one per SIMD with the whole 512-entry register file each (256 accumulator
    // registers at MB x WNB = 16 x 4: a 256 x 256 tile), and no room for loader waves - a fifth wave would halve every wave's register
    // budget - so each consumer wave issues its quarter of the activation DMA itself, between its MFMAs.
    constexpr bool SELF = LOADERS == 0;
    constexpr int ISSUERS = SELF ? CW : LOADERS;
    constexpr int LOADS = STAGE_BYTES / 1024 / ISSUERS;  // 1 KiB DMA pieces per issuing wave and stage
    constexpr int TLOADS = MB / ISSUERS;                 // the fp16 tail's X_out blocks per issuing wave
    constexpr int WL = F6 ? 2 * WNB : WNB;               // load instructions of one k-step's weight fragments (per wave)
    // PAIR (template parameter ABL = 90; MIXQ_ACT_SILU_PAIR): gate_proj and up_proj of an MLP block as ONE GEMM.  The weight rows are interleaved in groups of
    // four - up[2g], up[2g+1], gate[2g], gate[2g+1] (scale_col, bias and weight_cache rows likewise) - so the four consecutive channels a
    // lane holds of a 16 x 16 accumulator block ARE the two (up, gate) pairs of output channels 2g, 2g+1: the product
    // (silu(gate) + bias_gate) * fp16(up + bias_up) is lane-local, leaves the epilogue as ONE packed fp16 pair, and the tile stages and
    // stores BN / 2 output columns.  Same operations per element as up_proj's launch followed by gate_proj's MIXQ_ACT_SILU_MUL launch
    // (mixquant/modules/fused/mlp.py:57-63): bit-identical; up's 11 MB round trip through memory and one launch disappear.
    constexpr bool PAIR = ABL == 90;
    constexpr int BNO = PAIR ? BN / 2 : BN;              // output columns of the staged tile
    constexpr int OPITCH = BNO * 2 + 16;
    constexpr bool PREBIAS = !SELF && (MB * WNB * 4 + (D + 1) * WNB * (F6 ? 6 : 4) * (I4 ? 2 : 1) + MB * (F6 ? 6 : 4) + 40 <= 232);   // registers to spare for the bias prefetch
    constexpr int AMAX_OFF = (BM * OPITCH + 15) & ~15;   // LDS: [CW][4][BM] row maxima (one slot per wave and lane group), behind the staging tile
    // tail k-steps (32 outlier columns each) whose X_out blocks go through LDS: two; four in the FP6 form - 4-bit layers carry 128
    // static fp16 columns (linear.py:100, fp_features_num), and k-steps beyond TQ cost every wave a global round trip for the same rows
    constexpr int TQ = SELF ? 0 : (F6 ? 4 : 2);
    constexpr int TAILX = NSTAGE * STAGE_BYTES;          // LDS offset of those blocks: behind the ring, [TQ][MB] x 1 KiB
    // EPI2 (round 4): the epilogue as a pipeline of ROW PANELS (PJ = 2 blocks = 32 token rows each).  A consumer wave dequantises panel
    // p+1 between the fp16 tail MFMAs of panel p (the VALU work hides the MFMAs' latency and the other way round; the first form ran
    // 24 dependent MFMA pairs back to back and stalled in order behind each), converts panel p to fp16 (v_cvt_pk_f16_f32), stages it in
    // LDS and raises a flag word in LDS behind its stores (one LDS unit per CU executes a wave's operations in order: whoever reads the flag
    // set reads the panel complete - no wait, no barrier on the producing side); the LOADER waves - idle since their last DMA - poll the
    // flags and copy staged panels out to Y with 16-byte nt row stores while the consumers work on the next panels, so only the last
    // panels' stores (shared by all six waves behind the one closing barrier) are exposed behind the arithmetic.
    // Same operations per element in the same order as the first form (kept for the fat prefill tile, the pairwise split-K form and as
    // the tuning build's A/B partner, ABLK 60): bit-identical (tests/test_gpu_round4.py).
    constexpr int PJ = MB >= 2 ? 2 : 1, NPAN = MB / PJ, PROWS = PJ * 16;
    // panels the loader waves copy out on their own (the rest is shared by all waves behind the final barrier: the path to memory takes
    // ~30 cycles per KiB stored, about as long as the consumers need to produce it, so two waves cannot drain more than half the tile in time)
    // (Tried and dropped, profiles/r04_ab_direct_stores.txt: no staging at all - block pairs exchanged between lanes with v_permlane16_swap so
    // that a lane holds 16 contiguous bytes, stored straight from registers as 64- / 32-byte row segments.  Bit-identical, 28.6 vs 24.6 us:
    // the memory system chokes on partial-line writes; full 384-byte row segments out of LDS are what it takes.)
    constexpr int LP = (SELF || NPAN < 3) ? 0 : (ABLK == 61 ? 0 : (ABLK == 67 ? NPAN - 1 : NPAN - 2));   // (NPAN - 1, probe 67: the consumers end up waiting for the loaders: 24.14 vs 23.94 us)
    constexpr int FLAG_OFF = TAILX + TQ * MB * 1024;     // LDS: "panel p staged by consumer wave w" words [NPAN][CW], behind the tail's X_out blocks
    // EPI2: the epilogue's scales (x_scale of the tile's rows, scale_col and bias of its columns) do not ride through the k loop in the
    // consumers' registers (20 VGPRs of the 128 x 192 tile, 14 requests and their address arithmetic in front of its second weight
    // request): a loader wave brings them into LDS by 16-bit LDS-DMA - one dword slot per element, zero-extended; rows past M and columns
    // past N read as zero through the buffer descriptor's range check - while the main loop runs, and the epilogue reads them from there.
    constexpr int SX_OFF = FLAG_OFF + 64, SW_OFF = SX_OFF + BM * 4, BI_OFF = SW_OFF + BN * 4, SC_END = BI_OFF + BN * 4;
    constexpr bool EPI2 = !SELF && ABLK != 50 && ABLK != 60 && (MB % PJ == 0) &&
                          (((BM * OPITCH + 15) & ~15) + WR_CW * 4 * BM * 4 <= TAILX);   // (the staged tile and the row-maximum slots stay clear of the tail's X_out blocks)
    static_assert(MB % ISSUERS == 0 && (STAGE_BYTES / 1024) % ISSUERS == 0, "pieces must divide evenly over the issuing waves");
    // (round 6: the tuple-ring form's feed ablations - 1 no weight loads, 2 no X traffic, 3 MFMA only, 6 no k-loop barriers, 32 DMA without fragment
    // reads, 33 fragment reads without DMA - with pseudo-random stand-ins, and 36: the in-k-step timeline of tools/trace_kstep.py)
    static_assert(!F6 || (!SELF && (ABLK == 0 || ABLK == 1 || ABLK == 2 || ABLK == 3 || (F6R && (ABLK == 6 || ABLK == 32 || ABLK == 33 || ABLK == 36 || ABLK == 38 || ABLK == 39 || ABL >= 100)))), "the FP6 form exists for the shipped loop (and its feed ablations) only");
    static_assert(LOOK >= 1 && (LOADS + (ABLK == 76 ? (BN / 16) / (LOADERS ? LOADERS : 1) : 0)) * NEWER < 64, "vmcnt range");
    static_assert(!SELF || (LOOK == D + 1 && !I4 && ABLK == 0 && (D - 1) * (WNB + LOADS) < 64), "self-loading form: X stage kt+1 and the weights of k-step kt are requested in the same k-step");
    static_assert(!EPI2 || SC_END <= 160 * 1024, "X ring + tail blocks + panel flags + scales must fit the 160 KiB of LDS");
    static_assert(!PAIR || (EPI2 && (Q == 0 || Q == 3)), "the paired gate / up epilogue exists in the panel form of the int8 and the FP6 tuple-ring kernels");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if constexpr (ABLK == 9) { if (a.act != 12345) return; }                      // launch floor of this grid / LDS footprint
    // Every argument the prologue needs, asked for in the FIRST basic block: the compiler requests a kernel argument where it is first used,
    // one scalar load per use site with a wait behind each (nine dependent round trips to a cold scalar cache in front of the first
    // operand request: ~0.3 us of the 1.0 us between entry and the first k-step); named together here they become a few wide loads and ONE wait.
    asm volatile("" :: "s"(a.qx), "s"(a.qw), "s"(a.sx), "s"(a.sw), "s"(a.M), "s"(a.N), "s"(a.KB), "s"(a.tiles_m), "s"(a.tiles_n),
                 "s"(a.xblocks), "s"(a.wblocks), "s"(a.gm), "s"(a.n_begin), "s"(a.n_out_dev), "s"(a.bias), "s"(a.n_out), "s"(a.mg_group), "s"(a.mg_gm),
                 "s"(a.mg_last), "s"(a.ldy), "s"(a.y));
    const bool staged = (((PAIR ? a.N >> 1 : a.N) & 7) == 0) && ((a.ldy & 7) == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);

    // KS (ABLK = 50): pairwise split-K.  Long-K layers with few output tiles (11008 -> 4096 at 512 tokens: 128 tiles of 128 x 128, half the
    // CUs idle, or 256 tiles of 64 x 128 at a third fewer MACs per operand byte) run TWO workgroups per tile, each over half of K: the
    // second publishes its int32 accumulators (register order, write-through 16-byte stores) into the tile's workspace slot and raises the
    // tile's flag, the first adds them to its own and runs the epilogue.  Integer sums: exact in any order, the result is bit-identical.
    // The hand-off costs 2.9 us between partners on one XCD (tools/ubench_handoff.hip) - the tile map below puts the halves of a tile next
    // to each other in an XCD's run.  Both halves must be resident at once: the host launches this form only when 2 x tiles <= CUs.
    constexpr bool KS = ABLK == 50;
    const int ntiles = a.tiles_m * a.tiles_n;
    int tile, kz = 0;
    {
        const int nu = KS ? 2 * ntiles : ntiles;
        const int b = blockIdx.x, q = nu >> 3, r = nu & 7, x = b & 7, s = b >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;          // XCD-aware, bijective for any count
        if constexpr (KS) { kz = tile & 1; tile >>= 1; }
    }
    // Tile order inside an XCD's contiguous run of tiles: groups of gm M tiles, M fastest inside a group, then N, then the next group.
    // With gm = tiles_m (few M tiles, e.g. the 4 of a 512-token batch) that is "M fastest": the CUs of an XCD share a handful of
    // weight panels and ALL of the activation.  At prefill sizes (32 M tiles of 128 rows at 4096 tokens) M-fastest makes the 32 CUs
    // of an XCD stream 32 different activation slabs against ONE weight panel - 17 MB of L2 misses per 32 tiles; groups of 8 M tiles
    // x 4 weight panels need 8 MB for the same work.
    int tm, tn;
    {
        auto fdiv = [](int n, unsigned int m) MIXQ_INL { return m ? static_cast<int>(__umulhi(static_cast<unsigned int>(n), m)) : n; };
        const int per_group = a.gm * a.tiles_n, grp = fdiv(tile, a.mg_group), first_m = grp * a.gm;
        const bool last = a.tiles_m - first_m < a.gm;
        const int gsz = last ? a.tiles_m - first_m : a.gm;
        const int r = tile - grp * per_group;
        tn = fdiv(r, last ? a.mg_last : a.mg_gm); tm = first_m + (r - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN + a.n_begin;                            // (n_begin: this launch covers the weight rows from there - mixq_wr_split)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int nk_all = a.KB >> 6;
    const int kbeg = KS ? (kz ? nk_all >> 1 : 0) : 0;                            // this workgroup's k-steps: [kbeg, kbeg + nk)
    const int nk = KS ? (kz ? nk_all - (nk_all >> 1) : nk_all >> 1) : nk_all;
    // (Rounds 1-2 could start neighbouring weight panels a few k-steps apart - integer accumulation is exact in any order - to keep the CUs of
    // an XCD from asking for the same activation slab at the same moment: it never changed a timing and cost two scalar counters per wave; removed.)
    auto stamp = [&](int slot) MIXQ_INL {                         // diagnostics (mixq_gemm_set_trace): tools build only
#ifdef MIXQ_TUNING
        if (a.trace && tid == 0) {
            a.trace[blockIdx.x * 16 + slot] = wall_clock64();
            a.trace[blockIdx.x * 16 + 8 + slot] = __builtin_readcyclecounter();
        }
#else
        (void)slot;
#endif
    };
    stamp(0);
#ifdef MIXQ_TUNING
    // ---- ABL 77, timing probe (NOTEBOOK.md round 6, "the one-launch form"): what would ONE launch for quantise + GEMM cost?  Every workgroup first runs a
    // stand-in quantise pass over its share of the token rows - the row's fp16 values in (16 bytes per thread and chunk), maximum through DPP + LDS,
    // scale, eight conversions per chunk, 8 bytes out per chunk with WRITE-THROUGH stores into a scratch image (the GEMM below keeps reading the real q_x:
    // the launch's results stay correct), drain, ONE agent-scope atomic add per row on the counter of the row's 128-row M tile - then the GEMM as it is,
    // whose loader waves poll their M tile's counter (bounded: a bug ends in a slow launch, not a hung box) in front of their first activation
    // stage while the consumer waves are already requesting weights; the last poller of a tile resets its counters.  No outlier handling in the
    // stand-in (the real pass spends ~0.9 of its 3 us there).
    if constexpr (ABLK == 77) {
        if (a.fx) {
            const int nch = a.fK >> 3;                                             // 16-byte chunks per row
            if (tid == 0) *reinterpret_cast<volatile uint32_t*>(lds + NSTAGE * STAGE_BYTES - 16) = 0u;      // (the loaders' relay word: LDS is not cleared between launches)
            // ONE ROW PER WAVE (wave w of workgroup b: row w x workgroups + b): one far round trip for the row, the maximum inside the wave, no LDS, no
            // workgroup barrier; the first form walked a workgroup's two or three rows one after the other with all six waves (47.7-59 us per forward)
            const int r = wave * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x);
            if (r < a.M && nch <= 64 * 16) {
                const uint16_t* xr = a.fx + static_cast<size_t>(r) * a.fK;
                constexpr int CH = 16;                                             // chunks per lane at most (K <= 8192)
                u32x4 c[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q) { c[q] = u32x4{0, 0, 0, 0}; if (q * 64 + lane < nch) c[q] = *reinterpret_cast<const u32x4*>(xr + (q * 64 + lane) * 8); }
                uint32_t m = 0;
#pragma unroll
                for (int q = 0; q < CH; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const uint32_t v = c[q][e] & 0x7fff7fffu; const uint32_t lo = v & 0xffffu, hi = v >> 16; m = m > lo ? m : lo; m = m > hi ? m : hi; }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(m, off); m = m > o ? m : o; }
                const float amax = h2f(static_cast<uint16_t>(m));
                const float inv = amax > 0.f ? 127.f / amax : 0.f;
                const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(a.fq + static_cast<size_t>(r) * a.fK, 0, a.fK, 0x00020000);
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int ch = q * 64 + lane;
                    if (ch < nch) {
                        u32x2 o = {0u, 0u};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int q0 = static_cast<int>(__builtin_rintf(h2f(static_cast<uint16_t>(c[q][e] & 0xffffu)) * inv));
                            const int q1 = static_cast<int>(__builtin_rintf(h2f(static_cast<uint16_t>(c[q][e] >> 16)) * inv));
                            const uint32_t pair = (static_cast<uint32_t>(q0) & 0xffu) | ((static_cast<uint32_t>(q1) & 0xffu) << 8);
                            if (e < 2) o.x |= pair << (16 * e); else o.y |= pair << (16 * (e - 2));
                        }
                        if (!(a.fflags & 1)) { if (a.fflags & 16) __builtin_amdgcn_raw_buffer_store_b64(o, rq, ch * 8, 0, 0); else __builtin_amdgcn_raw_buffer_store_b64(o, rq, ch * 8, 0, 16 /* sc1: write-through */); }
                        else asm volatile("" :: "v"(o));
                    }
                }
                if (!(a.fflags & 2)) wr_wait_vmcnt<0>();
                if (lane == 0 && !(a.fflags & 4)) __hip_atomic_fetch_add(a.fcnt + r / BM, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                                                      // (the relay word is cleared before a loader can set it)
        }
    }
#endif
    // second set of stamps (tools/trace_gemm.py --panels): the same buffer 16 x 4096 entries further on; `who`: the thread that stamps
    auto stamp2 = [&](int slot, int who) MIXQ_INL {
#ifdef MIXQ_TUNING
        if (a.trace && tid == who) a.trace[16 * 4096 + blockIdx.x * 16 + slot] = wall_clock64();
#else
        (void)slot; (void)who;
#endif
    };
    // EPI2 copy-out: rows [p PROWS, (p + 1) PROWS) of the staged fp16 tile -> Y, 16 bytes per lane, every LDS read of a lane issued before
    // its first store (ONE LDS round trip), streaming (nt) stores - Y is written once and not re-read by this kernel.  The first form
    // spent its "store issue" time on ADDRESSES (a division by the row length, 64-bit multiply-adds and two bounds tests per 16 bytes:
    // ~20 VALU instructions, some quarter-rate, per store).  Here a thread's chunk geometry - byte offset in Y relative to the panel's
    // first row, byte offset in the staged tile - is worked out ONCE (copy_setup: the loaders during their idle drain, the consumers
    // before the barrier) and is the same for every panel; a panel's stores are buffer stores on a descriptor that starts at the panel's
    // first row and ends behind its last valid one, so rows past M are dropped by the hardware's range check and chunks past N (or past
    // the panel) carry an offset that is always out of range: no address arithmetic and no test per store.
    constexpr int CPR = BNO / 8;                                                 // 16-byte chunks per tile row
    constexpr int CPCH = PROWS * CPR;                                            // ... per panel
    constexpr int IT_L = SELF ? 1 : (CPCH + LOADERS * 64 - 1) / (LOADERS * 64 > 0 ? LOADERS * 64 : 1);   // per loader thread
    constexpr int IT_A = (CPCH + NT - 1) / NT;                                   // per thread when every wave copies (the last panel)
    auto copy_setup = [&](int t, auto nthr_c, uint32_t* voff, int* loff) MIXQ_INL {
        constexpr int NTHR = decltype(nthr_c)::value, IT = (CPCH + NTHR - 1) / NTHR;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int q = t + it * NTHR, r = q / CPR, c = q - r * CPR;
            const int no = (PAIR ? n0 >> 1 : n0) + c * 8;                         // first output column of the chunk
            const bool ok = q < CPCH && no < (PAIR ? a.N >> 1 : a.N);
            voff[it] = ok ? static_cast<uint32_t>(r * a.ldy + no) * 2u : 0x80000000u;
            loff[it] = ok ? r * OPITCH + c * 16 : 0;
        }
    };
    auto copy_panel = [&](int p, auto nthr_c, const uint32_t* voff, const int* loff) MIXQ_INL {
        constexpr int NTHR = decltype(nthr_c)::value, IT = (CPCH + NTHR - 1) / NTHR;
        int rows = a.M - m0 - p * PROWS;                                         // valid rows from the panel's first one on
        rows = rows < 0 ? 0 : (rows > PROWS ? PROWS : rows);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.y + static_cast<size_t>(m0 + p * PROWS) * a.ldy, 0, rows * a.ldy * 2, 0x00020000);
        u32x4 v[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) v[it] = *reinterpret_cast<const u32x4*>(lds + p * PROWS * OPITCH + loff[it]);
        if (ABLK != 7 || a.act == 12345) {
#pragma unroll
            for (int it = 0; it < IT; ++it) __builtin_amdgcn_raw_buffer_store_b128(v[it], rs, voff[it], 0, ABLK == 8 ? 0 : (PAIR ? MIXQ_PAIR_ST : (MB <= MIXQ_Y_WT_MB ? 18 : MIXQ_Y_ST)));
        }
    };
    uint32_t voffA[IT_A];                                                        // this thread's chunks when all NT threads copy a panel
    int loffA[IT_A];

    // =================================================================================================================
    // loader wave(s): X stage kt+LOOK issued, stage kt+1 retired, then the k-step's barrier
    // =================================================================================================================
    if constexpr (!SELF) {
    if (wave >= CW) {
        __builtin_amdgcn_s_setprio((FL & 8) ? 0 : 2);
        const int lw = wave - CW;
        const uint8_t* src[LOADS];
        int dsto[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int p = lw + i * LOADERS;
            if constexpr (F6) {
                // 1 KiB pieces of a stage of 1.5 KiB blocks (a piece may straddle two blocks).  The activation image (MIXQ_FMT_R6X128) is
                // row-major inside a block - 96 bytes per row: four 16-byte pieces, then four 8-byte pieces - which is what a quantise
                // kernel writes well; the LDS image is the same bytes with the two 8-byte-piece PAIRS (the 16-byte units at 64 and 80 of a row)
                // SWAPPED in rows 8 .. 15: the 16-byte piece g of row r sits at 96 r + 16 g (bank quads 6 r + g mod 16: conflict-free for the
                // lane groups ds_read_b128 is served in), the 8-byte piece at 96 r + 64 + 8 (g ^ 2 (r >> 3)) (conflict-free for the 32-lane
                // groups of ds_read_b64: rows r and r + 8 would collide).  A DMA lane's source address is free, a 32-byte pair never
                // straddles a KiB, so every DMA instruction still moves one contiguous KiB.
                const int o = p * 1024 + lane * 16, blk = o / BLK, within = o - blk * BLK;
                int rb = (m0 >> 4) + blk; rb = rb < a.xblocks ? rb : a.xblocks - 1;
                // (round 6) ONLY the pair of 8-byte pieces is swapped: rounds 3-5 swapped the 16-byte pieces of rows 8 .. 15 as well, for a service
                // order of 16 CONSECUTIVE lanes per LDS cycle - the hardware serves ds_read_b128 in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
                // 28-31}, ... (MI355X_MICROARCH.md, LDS), for which the UNswapped rows are conflict-free (bank quads 6 r + g mod 16: evens from one
                // half of a group, odds from the other) and the swapped ones collide two-way in every group: SQ_LDS_BANK_CONFLICT 1.10 M cycles per
                // launch = 128 per k-step and CU, every b128 fragment read at half rate (profiles/r06_w4a4_loop.txt)
                constexpr bool OLDSWZ = (FL & 128) != 0;
                const int off = within ^ (((within / 96) >= 8 && (OLDSWZ || (within % 96) >= 64)) ? 16 : 0);
                src[i] = a.qx + static_cast<size_t>(rb) * BLK + off;
            } else {
                int rb = (m0 >> 4) + p; rb = rb < a.xblocks ? rb : a.xblocks - 1;  // blocks past M: loaded, computed, dropped
                src[i] = a.qx + static_cast<size_t>(rb) * 1024 + lane * 16;
            }
            dsto[i] = p * 1024;
        }
        const size_t xks = static_cast<size_t>(a.xblocks) * BLK;
        if constexpr (KS) {                              // (this workgroup's half of K starts at k-step kbeg)
#pragma unroll
            for (int i = 0; i < LOADS; ++i) src[i] += static_cast<size_t>(kbeg) * xks;
        }
        size_t xoff = 0;                                 // byte offset of the k-step the next stage reads
        // (ABLK 76, timing probe, results are garbage: what would the loop cost with the WEIGHTS staged through LDS by the loader waves - 1 KiB
        // LDS-DMA pieces of the tile's weight blocks into one 12 KiB region behind the scales, the consumers reading their fragments from there with
        // ds_read_b128 instead of issuing vector loads themselves.  A vector load costs the issuing consumer wave ~39 cycles of its k-step, a fragment
        // read ~13: profiles/r05_ablations_random_operands.txt, r06_w8a8_kstep_timeline.txt)
        constexpr int WPL = WPROBE ? (BN / 16) / LOADERS : 0;
        const uint8_t* wsrc[WPL > 0 ? WPL : 1];
        size_t woffL = 0;
        if constexpr (WPROBE) {
#pragma unroll
            for (int i = 0; i < WPL; ++i) {
                int rb = (n0 >> 4) + lw + i * LOADERS; rb = rb < a.wblocks ? rb : a.wblocks - 1;
                wsrc[i] = a.qw + static_cast<size_t>(rb) * 1024 + lane * 16;
            }
        }
        auto stage = [&](int slot) MIXQ_INL {
            if constexpr (ABLK != 2 && ABLK != 3 && ABLK != 33) {
#pragma unroll
                for (int i = 0; i < LOADS; ++i) {
                    wr_glds16(src[i] + xoff, lds + slot * STAGE_BYTES + dsto[i]);
                    if constexpr (ABLK == 22 || ABLK == 23 || (FL & 2)) __builtin_amdgcn_s_sleep(1);     // probe: the loader's pieces spread over the k-step
                    if constexpr (ABLK == 24) __builtin_amdgcn_s_sleep(2);
                }
                if constexpr (WPROBE) {
#pragma unroll
                    for (int i = 0; i < WPL; ++i) wr_glds16(wsrc[i] + woffL, lds + SC_END + (lw + i * LOADERS) * 1024);
                    woffL += static_cast<size_t>(a.wblocks) * 1024;
                    if (woffL >= static_cast<size_t>(a.wblocks) * 1024 * nk_all) woffL = 0;
                }
            }
            xoff += xks;
        };
        constexpr int LPS = LOADS + WPL;                 // load instructions per stage and loader wave (the counted waits below)
        // Deep rings start with a RAMP: all 232 workgroups asking for LOOK stages at once (14 x 8 KiB each at the metric tile) puts
        // 26 MB of requests in front of everybody's first stage.  RP stages are requested up front, then two per k-step until the
        // loader is LOOK stages ahead.  (slot of stage RP + 2 i + 1 was last read LOOK + 2 k-steps earlier: free, as in steady state)
        // (FP6 form at 128-row tiles: 8 stages of 12 KiB - three up front instead of all six: -0.6 .. -1.1 % per launch, same-run A/B of two builds)
        constexpr int RP = (LOOK > 6 && ABLK != 12) ? 4 : (F6 && LOOK > 3 ? 3 : LOOK);
#ifdef MIXQ_TUNING
        if constexpr (ABLK == 77) {
            if (a.fx && !(a.fflags & 8)) {                // (probe: the loaders wait for their M tile's rows - the consumers are already requesting weights)
                const unsigned want = static_cast<unsigned>(a.M - m0 < BM ? a.M - m0 : BM);
                int spins = 0;
                // ONE lane of ONE loader wave polls, every ~0.5 us (all 64 lanes of 464 waves on four words: 55-59 us per forward - the words' memory channel
                // serialises the polls and the producers' adds behind them); the other loader wave watches a word in LDS (SC_END: behind the scales)
                volatile uint32_t* ldsflag = reinterpret_cast<volatile uint32_t*>(lds + NSTAGE * STAGE_BYTES - 16);     // (the ring's last 16 bytes: stage NSTAGE - 1 lands long after)
                if (lw == 0) {
                    if (lane == 0) {
                        while (__hip_atomic_load(a.fcnt + tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 14)) __builtin_amdgcn_s_sleep(16);
                        *ldsflag = 0x600DF00Du;
                    }
                } else if (lane == 0) {
                    while (*ldsflag != 0x600DF00Du && ++spins < (1 << 18)) __builtin_amdgcn_s_sleep(2);
                }
                if (lw == 0 && lane == 0) {
                    const unsigned seen = __hip_atomic_fetch_add(a.fcnt + a.tiles_m + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen == static_cast<unsigned>(a.tiles_n) - 1u) {                // the tile's last poller: everybody has passed
                        __hip_atomic_store(a.fcnt + a.tiles_m + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(a.fcnt + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
#endif
        int nxt, kt = 0;
        if (RP < LOOK && nk >= 2 * LOOK) {
#pragma unroll
            for (int s = 0; s < RP; ++s) { stage(s); if constexpr (ABLK == 26 || ABLK == 27) { if (s == 0) __builtin_amdgcn_s_sleep(8); } }
            wr_wait_vmcnt<LPS * (RP - 1)>();
            __builtin_amdgcn_s_barrier();                                        // B0: stage 0 landed
            wr_static_for<0, LOOK - RP>([&](auto i_c) MIXQ_INL {
                constexpr int i = decltype(i_c)::value;
                stage((RP + 2 * i) % NSTAGE);
                stage((RP + 2 * i + 1) % NSTAGE);
                wr_wait_vmcnt<LPS * (RP + i)>();                               // stage i + 1 landed: RP + i younger stages may be in flight
                if constexpr (ABLK != 6) __builtin_amdgcn_s_barrier();
            });
            kt = LOOK - RP;
            nxt = (2 * LOOK - RP) % NSTAGE;
        } else {
#pragma unroll
            for (int s = 0; s < LOOK; ++s)
                if (s < nk) stage(s);
            if (NEWER < nk) wr_wait_vmcnt<LPS * NEWER>(); else wr_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                                        // B0: stage 0 landed
            nxt = LOOK % NSTAGE;
        }
        // TOUCH (probe, ABLK 70 / 71): the weight lines of k-step kt + LOOK are pulled into this XCD's L2 ten k-steps before the consumers
        // ask for them - one dword per 64-byte sector, this workgroup's share of the panel only (the gm workgroups of an XCD that stream
        // the same weight panel split its 12 KB per k-step between them), result never read.  Cold weights (a model's layers: every
        // launch streams 45 MB from HBM) stall the k loop by 100 cycles per k-step (profiles/r04_gemm_trace_warm_cold.txt): long-latency
        // requests hold the CU's memory queue.  Issued behind the k-step's wait: the hand-counted waits only get stricter by it.
        constexpr bool TOUCH = ABLK == 70 || ABLK == 71;
        const uint8_t* tbase = nullptr;
        int tlanes = 0;
        if constexpr (TOUCH) {
            const int nshare = a.gm < a.tiles_m ? a.gm : a.tiles_m;              // workgroups of this XCD sharing the panel (M fastest inside a group)
            const int sectors = BN;                                              // 64-byte sectors of the panel per k-step (BN x 64 bytes)
            const int per = (sectors + nshare - 1) / nshare;
            const int q = tm % nshare;
            int sec = q * per + lane;
            const size_t slab = static_cast<size_t>(a.wblocks) * 1024;           // bytes of one k-step of the image
            size_t off = static_cast<size_t>(n0 >> 4) * 1024 + static_cast<size_t>(sec) * 64;
            if (off + 64 > slab) off = slab - 64;                                // (the last tile's panel hangs over the image's rows)
            tbase = a.qw + off;
            tlanes = (lane < per && sec < sectors && lw == (ABLK == 71 ? 1 : 0)) ? 1 : 0;
        }
        size_t toff = static_cast<size_t>(kt + LOOK) * static_cast<size_t>(a.wblocks) * 1024;   // (the k-step whose stage the next iteration requests)
        int tsink = 0;
        // ABLK 36 (tools/trace_kstep.py): s_memtime stamps inside ONE k-step (k-step nk / 2; the test sits inside the asm statement, so the other
        // k-steps pay a compare and a branch per stamp): the loader's issue / wait / barrier here, the consumer's waits and MFMA groups below
        unsigned long long lst[4] = {0ull, 0ull, 0ull, 0ull};
        const int kmark_l = nk >> 1;
        auto lstamp = [&](int i, int ktv) MIXQ_INL {
#if defined(MIXQ_TUNING) && defined(__HIP_DEVICE_COMPILE__)
            if constexpr (TL) { const int ku = __builtin_amdgcn_readfirstlane(ktv); asm volatile("s_cmp_lg_u32 %1, %2\n\ts_cbranch_scc1 1f\n\ts_memtime %0\n1:" : "+s"(lst[i]) : "s"(ku), "s"(kmark_l) : "scc"); }
#else
            (void)i; (void)ktv;
#endif
        };
#ifdef MIXQ_TUNING
        // Cross-layer prefetch - built, measured, NOT shipped (tuning library only; profiles/r06_cold_prefetch_ab.txt,
        // r06_sprefetch_ubench.txt).  A model's layers stream their weights from HBM (the 256 MB memory-side cache holds the last ~5 layers'
        // images), which costs the metric launch +3 ... 4.6 us.  The idea: this launch's idle loader waves touch one dword per 128-byte line
        // of the NEXT launch's image so that it sits in the memory-side cache when its GEMM starts.  Two forms:
        //  (1) through the SCALAR cache - s_load_dword, a path to the L2 that bypasses the L1 and this wave's in-order vmcnt queue; up to PFB
        //      batches of four lines per k-step while fewer than 12 of the wave's 15 possible requests are in flight (IB_STS.LGKM_CNT), junk
        //      destination s98 / s99.  The L2 does fetch whole lines that way (FETCH_SIZE of a touch pass = that of a full read) but the
        //      memory-side cache does not KEEP them: a dependent-load chase over the image takes 0.70 us per load behind such a pass, as on a
        //      cold image (0.71), against 0.46 behind plain vector loads - the same holds for vector loads with the nt hint.  Forward with
        //      344 MB of images in rotation: 33.5 -> 34.3 ... 34.8 us; with 172 MB (all resident): 29.6 -> 33.0 us.
        //  (2) plain vector loads, 64 lines per instruction every few k-steps: the lines are kept (the next launch does run from the cache),
        //      but every far request holds one of the CU's L1 miss entries for an HBM round trip - the window the k loop's own requests turn
        //      over in (NOTEBOOK.md round 4) - and the two effects cancel: 33.8 -> 33.5 ... 33.7 us cold, 30.8 -> 31.4 us when all images are
        //      resident; with the nt hint (not kept, HBM read twice) 33.7 -> 38.0 us.
        // No path from a CU fills the memory-side cache without paying for it in the CU's own request window; the product library carries none.
        constexpr int PFB = 4;
        const uint8_t* pfp = nullptr;
        int pfn = 0;                                                             // batches of four lines left
        if constexpr (!KS) {
            if (a.pf && (a.pf_mode & 1)) {
                const unsigned per = (a.pf_lines / (static_cast<unsigned>(gridDim.x) * LOADERS)) & ~3u;
                pfp = a.pf + static_cast<size_t>(blockIdx.x * LOADERS + lw) * per * 128;
                pfn = static_cast<int>(per >> 2);
            }
        }
        // (mode 2, experiment: ONE vector load of one dword per line - 64 lines per instruction - every PFV k-steps.  It sits in this wave's in-order
        // vmcnt queue: the stage wait below only gets stricter by it, and passes unhindered as long as the far load returns within NEWER k-steps)
        int pfv = 0;                                                             // vector instructions left
        int pfv_every = 1, pfv_cnt = 0;
        const uint8_t* pfvp = nullptr;
        int pfsink = 0;
        if constexpr (!KS) {
            if (a.pf && (a.pf_mode & 2)) {
                const unsigned per = (a.pf_lines / (static_cast<unsigned>(gridDim.x) * LOADERS)) & ~63u;      // lines per wave, whole instructions
                pfvp = a.pf + static_cast<size_t>(blockIdx.x * LOADERS + lw) * per * 128 + static_cast<size_t>(lane) * 128;
                pfv = static_cast<int>(per >> 6);
                const int span = nk - LOOK - 6;
                pfv_every = pfv > 0 && span > pfv ? span / pfv : 1;
                if (!(a.pf_mode & 1)) pfn = 0;
            }
        }
        auto prefetch_tick = [&]() MIXQ_INL {
#if defined(__HIP_DEVICE_COMPILE__)
            if (pfv > 0 && ++pfv_cnt >= pfv_every) {
                pfv_cnt = 0;
                if (a.pf_mode & 4) asm volatile("global_load_dword %0, %1, off nt" : "+v"(pfsink) : "v"(pfvp) : "memory"); else asm volatile("global_load_dword %0, %1, off" : "+v"(pfsink) : "v"(pfvp) : "memory");
                pfvp += 64 * 128;
                --pfv;
            }
#pragma unroll 1
            for (int b = 0; b < PFB; ++b) {
                unsigned inflight;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS, 8, 4)" : "=s"(inflight));
                if (inflight > 11u || pfn <= 0) break;
                const uint8_t* p_ = pfp;
                asm volatile("s_load_dword s98, %0, 0x0\n\ts_load_dword s98, %0, 0x80\n\ts_load_dword s98, %0, 0x100\n\ts_load_dword s98, %0, 0x180"
                             :: "s"(p_) : "s98", "s99", "memory");
                pfp += 512;
                --pfn;
            }
#endif
        };
#endif
        for (; kt + LOOK < nk; ++kt) {
            lstamp(0, kt);
            if constexpr (FL & 4) __builtin_amdgcn_s_sleep(3);                    // probe: the consumers' first MFMA group behind the barrier runs without the loaders' burst
            stage(nxt);
#ifdef MIXQ_TUNING
            if constexpr (!KS) { if ((pfn > 0 || pfv > 0) && kt + LOOK + 4 < nk) prefetch_tick(); }
#endif
            lstamp(1, kt);
            if constexpr (ABLK != 5) wr_wait_vmcnt<LPS * NEWER>();              // stage kt+1 landed
            lstamp(2, kt);
            if constexpr (ABLK != 6) __builtin_amdgcn_s_barrier();
            lstamp(3, kt);
            nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
            if constexpr (TOUCH) {
                // (the destination stays a LIVE register until the loader's last vmcnt(0): a dead one would be handed to something else while
                // the load is still in flight, and the returning dword would land in it)
                if (tlanes) asm volatile("global_load_dword %0, %1, off" : "+v"(tsink) : "v"(tbase + toff) : "memory");
                toff += static_cast<size_t>(a.wblocks) * 1024;
            }
        }
        // The fp16 outlier tail's activation operand (X_out rows of this tile) is the same for the four consumer waves: like X it
        // goes through LDS once - fragment-ordered 16 x 32 blocks of the first TQ tail k-steps (64 outlier columns), DMA-ed
        // behind the ring - instead of being fetched from L2 by each wave (4 x 8 KiB per tail k-step: as much traffic as two
        // k-steps of the main loop, 0.6 us of a 27 us launch at 41 outlier columns).  Requested HERE, when the last X stage has been
        // issued and LOOK k-steps of the main loop are still to run, so it has landed long before the consumers reach the epilogue.
        // (round 6) ... and AFTER the first drain k-step's barrier, with COUNTED waits in the drain: rounds 4-5 issued the scales and the tail blocks right
        // behind the last stage and drained with vmcnt(0) - the first drain wait then sat on the stage requested a moment ago AND on the tail's far
        // loads (X_out has just been written by the quantise kernel, on other XCDs), the loader reached that k-step's barrier late and the whole
        // workgroup with it: the k loop grew with the outlier count (int8 18.4 / 18.8 / 19.2 us at 0 / 41 / 128 columns, FP6 13.2 / 13.8 / 14.2 us at
        // 0 / 64 / 128; profiles/r06_tail_drain.txt).  The counts must be compile-time, so both loader waves request the scales (the same bytes to
        // the same LDS words), the bias slot is always loaded (scale_col again when there is no bias: never read) and a launch with outlier operands
        // requests all TQ tail k-steps (chunks past the padded width read column 0; the fix-up below zeroes what is not a column).
        constexpr int SCL = EPI2 ? (BM + 63) / 64 + 2 * ((BN + 63) / 64) : 0;      // scale requests per loader wave
        constexpr int TLQ = TQ * TLOADS;                                         // tail requests per loader wave
        constexpr bool CDRAIN = LOOK >= 3 && LPS * (LOOK - 3) + SCL + TLQ < 64 && LPS * (LOOK - 2) < 64;   // (an upper bound of every count below)
        const bool has_tail = a.xo && a.wo;
        auto issue_scales = [&]() MIXQ_INL {
            if constexpr (EPI2) {
                // the scales into LDS (see SX_OFF)
                const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.sx) + m0, 0, (a.M - m0) * 2, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.sw) + n0, 0, (a.N - n0) * 2, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.bias ? a.bias : a.sw) + n0, 0, (a.N - n0) * 2, 0x00020000);
#pragma unroll
                for (int q = 0; q < (BM + 63) / 64; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(lds + SX_OFF + q * 256), 2, (q * 64 + lane) * 2, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < (BN + 63) / 64; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(lds + SW_OFF + q * 256), 2, (q * 64 + lane) * 2, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < (BN + 63) / 64; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void*)(lds + BI_OFF + q * 256), 2, (q * 64 + lane) * 2, 0, 0, 0);
            }
        };
        int kpad_l = 0;
        if (has_tail) {
            int n_out_l = a.n_out;
            if (a.n_out_dev) { const int nd = *a.n_out_dev; n_out_l = nd < n_out_l ? nd : n_out_l; }
            kpad_l = (n_out_l + 15) & ~15;
        }
        auto issue_tail = [&](int kk) MIXQ_INL {                                   // (has_tail: X_out blocks of tail k-step kk, TLOADS pieces per loader wave)
#pragma unroll
            for (int i = 0; i < TLOADS; ++i) {
                const int p = lw + i * LOADERS;
                int xr = m0 + p * 16 + (lane & 15); xr = xr < a.M ? xr : a.M - 1;
                int col = kk * 32 + (lane >> 4) * 8; col = col < kpad_l ? col : 0;        // chunks past the padded width: any valid address (zeroed by the fix-up)
                wr_glds16(reinterpret_cast<const uint8_t*>(a.xo + static_cast<size_t>(xr) * a.ldxo + col),
                          lds + TAILX + (kk * MB + p) * 1024);
            }
        };
        // ONE tail k-step per drain k-step (its TLOADS pieces take the loader a few hundred cycles of the ~600 it idles per k-step; all at once they
        // made it late for the next barrier), the scales with the first
        bool counted = false;
        if constexpr (CDRAIN && ABLK != 6 && ABLK != 5) {
            if (nk > LOOK) {                              // (the main loop has run: stage kt + 1 .. nk - 1 are the LOOK - 1 stages in flight)
                counted = true;
                wr_static_for<0, LOOK - 1>([&](auto d_c) MIXQ_INL {
                    constexpr int d = decltype(d_c)::value, issued = d < TQ ? d : TQ;     // tail k-steps requested in front of this wait
                    // stage kt + 1 landed: LOOK - 2 - d younger stages, the scales (d >= 1) and `issued` tail k-steps may be in flight
                    if (has_tail) wr_wait_vmcnt<LPS * (LOOK - 2 - d) + (d ? SCL : 0) + TLOADS * issued>();
                    else wr_wait_vmcnt<LPS * (LOOK - 2 - d) + (d ? SCL : 0)>();
                    __builtin_amdgcn_s_barrier();
                    if constexpr (d == 0) issue_scales();
                    if (has_tail) {
#pragma unroll
                        for (int kk = 0; kk < TQ; ++kk)
                            if (kk == d || (d == LOOK - 2 && kk > d)) issue_tail(kk);
                    }
                });
                kt = nk - 1;
            }
        }
        if (!counted) {                                   // (fewer k-steps than the ring is deep, probes: everything at once, drained with vmcnt(0))
            issue_scales();
            if (has_tail) {
#pragma unroll
                for (int kk = 0; kk < TQ; ++kk) issue_tail(kk);
            }
            for (; kt + 1 < nk; ++kt) {
                wr_wait_vmcnt<0>();
                if constexpr (ABLK != 6) __builtin_amdgcn_s_barrier();
            }
        }
        wr_wait_vmcnt<0>();                                                      // (nk = 1: no drain iteration waited for the tail blocks)
#ifdef MIXQ_TUNING
        if constexpr (!KS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pfsink) :: "s98", "s99", "memory");   // (the prefetch's junk returns: all in before anything else may use the registers)
#endif
        if constexpr (TOUCH) asm volatile("" :: "v"(tsink));
#ifdef MIXQ_TUNING
        if constexpr (TL) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lst[0]), "+s"(lst[1]), "+s"(lst[2]), "+s"(lst[3]));
            if (a.trace && lw == 0 && lane == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a.trace[2 * 16 * 4096 + blockIdx.x * 16 + 11 + i] = lst[i];
            }
        }
#endif
        uint32_t voffL[IT_L];
        int loffL[IT_L];
        if constexpr (EPI2) {
            // EPI2, in the shadow of the consumers' last k-step (which has no barrier): (1) the X_out blocks of the tail in LDS are made
            // safe to multiply as they are - columns >= n_out (the pad of the caller's buffer may hold anything, NaN patterns included) and
            // whole k-steps that do not exist (never requested: stale LDS bytes) become zeros - so the consumers spend no VALU on masks
            // (each loader wave fixes the blocks it requested itself: its own vmcnt(0) above covers them); (2) this thread's copy-out geometry.
            if constexpr (TQ > 0) {
                int n_out_l = 0;
                if (a.xo && a.wo) {
                    n_out_l = a.n_out;
                    if (a.n_out_dev) { const int nd = *a.n_out_dev; n_out_l = nd < n_out_l ? nd : n_out_l; }
                }
#pragma unroll
                for (int kk = 0; kk < TQ; ++kk) {
                    if (kk * 32 + 32 > n_out_l) {                               // (wave-uniform: full k-steps are left alone)
                        const int kb = kk * 32 + (lane >> 4) * 8;
                        u32x4 keep;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            uint32_t k = 0;
                            if (kb + 2 * d < n_out_l)     k |= 0x0000ffffu;
                            if (kb + 2 * d + 1 < n_out_l) k |= 0xffff0000u;
                            keep[d] = k;
                        }
#pragma unroll
                        for (int i = 0; i < TLOADS; ++i) {
                            u32x4* blk = reinterpret_cast<u32x4*>(lds + TAILX + (kk * MB + lw + i * LOADERS) * 1024 + lane * 16);
                            *blk = kk * 32 < n_out_l ? (*blk & keep) : u32x4{0, 0, 0, 0};
                        }
                    }
                }
            }
            if (tid - CW * 64 < NPAN * CW) *reinterpret_cast<uint32_t*>(lds + FLAG_OFF + (tid - CW * 64) * 4) = 0u;   // the panels' flags: clear before anybody can raise one
            copy_setup(tid - CW * 64, std::integral_constant<int, LOADERS * 64>{}, voffL, loffL);
            copy_setup(tid, std::integral_constant<int, NT>{}, voffA, loffA);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the fix-up's ds_writes: s_barrier does not wait for them
        }
        __builtin_amdgcn_s_barrier();                                            // the epilogue's first barrier (ring dead, X_out blocks landed)
        if constexpr (EPI2) {
            // EPI2: panels 0 .. LP-1 are copied out HERE, each as soon as all four consumer waves have raised its flag, under the consumers'
            // work on the later panels (which are shared by all waves behind the closing barrier).  Outputs that cannot be staged are
            // stored by the consumers themselves.
            if constexpr (ABLK == 62) __builtin_amdgcn_s_setprio(0);              // probe: the copying loaders below the consumers they share SIMDs with
            stamp2(8, CW * 64);
            if (staged) {
#pragma unroll
                for (int p = 0; p < LP; ++p) {
                    for (;;) {
                        const u32x4 f = *reinterpret_cast<const volatile u32x4*>(lds + FLAG_OFF + p * CW * 4);
                        if (f[0] & f[1] & f[2] & f[3]) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    asm volatile("" ::: "memory");                               // (the panel's reads stay behind the poll)
                    stamp2(9 + 2 * p, CW * 64);
                    copy_panel(p, std::integral_constant<int, LOADERS * 64>{}, voffL, loffL);
                    stamp2(10 + 2 * p, CW * 64);
                }
            }
            __builtin_amdgcn_s_barrier();                                        // every panel staged
        } else {
            __builtin_amdgcn_s_barrier();
        }
    }
    }

    // =================================================================================================================
    // consumer waves
    // =================================================================================================================
    const int lm = lane & 15, lq = lane >> 4;
    const int nw0 = n0 + wave * WN;                                              // first weight row of this wave (wave < CW)
    using acc_t = std::conditional_t<F6, f32x4, i32x4>;                          // (FP6 pipe: fp32 accumulators holding exact integers)
    acc_t acc[MB][WNB];
    // (tuning probes 78 / 79, garbage results: what would the k loop of the 128 x 256 tile cost on v_mfma_i32_32x32x32_i8 - half the MFMA instructions and half
    // the operand-register reads per MAC of the 16 x 16 x 64 form, same operand bytes?  78 = sixteen 32 x 32 x 32 MFMAs per k-step into shadow accumulators,
    // 79 = the shipped thirty-two 16 x 16 x 64 MFMAs into shadow accumulators; both run the epilogue on zeros, so the launches differ by the loop alone)
    typedef int i32x16_t __attribute__((ext_vector_type(16)));
    constexpr bool P32 = ABLK == 78, PSH = ABLK == 78 || ABLK == 79;
    static_assert(!PSH || (MB == 8 && WNB == 4 && Q == 0), "the 32 x 32 x 32 probe is written for the 128 x 256 int8 tile");
    i32x16_t acc32[P32 ? 4 : 1][P32 ? 2 : 1];
    i32x4 accs[ABLK == 79 ? MB : 1][ABLK == 79 ? WNB : 1];
    if constexpr (P32) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc32[m][nb][e] = 0;
    }
    if constexpr (ABLK == 79) {
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int i = 0; i < WNB; ++i) accs[j][i] = i32x4{0, 0, 0, 0};
    }
    uint16_t sxh[MB];
    u32x2 swp[WNB], bvp[WNB];                                                    // scale_col and bias of this wave's columns, requested with the scales
    int n_out_dev_v = 0;

    if (wave < CW) {
        if constexpr (ABLK == 28 || (FL & 16)) __builtin_amdgcn_s_setprio(3);     // probe: the MFMA waves above the loaders
        if (a.n_out_dev) n_out_dev_v = *a.n_out_dev;
        auto zero_acc = [&]() MIXQ_INL {
#pragma unroll
            for (int j = 0; j < MB; ++j)
#pragma unroll
                for (int i = 0; i < WNB; ++i) acc[j][i] = acc_t{0, 0, 0, 0};
        };
        if constexpr (SELF) zero_acc();                                          // (the others: behind their first operand requests)

        // weight stream (round 6): ONE buffer resource over the image, this wave's first block and lane in ONE offset register, the fragment's
        // block as the instruction's immediate offset, the k-step's byte offset as its scalar offset - no address arithmetic per load and two
        // scalar instructions per k-step.  (Rounds 2-5: a 64-bit base per fragment, s_add_u32 + s_addc_u32 in front of every global_load and a
        // five-instruction conditional advance - 11 scalar instructions per k-step of a loop that turned out to be bound by what the consumer
        // waves ISSUE between their MFMAs: profiles/r06_xwait_bisect.txt, 14 no-op-grade instructions per k-step cost 5 % of the launch.)
        // Blocks past N (computed and dropped) read whatever follows - the next k-step's first blocks, zeros past the image's end (range check).
        const unsigned wks = static_cast<unsigned>(a.wblocks) * BLK;               // bytes of one k-step of the image (images stay below 2 GB)
        const i32x4 wrs = {static_cast<int>(reinterpret_cast<size_t>(a.qw)), static_cast<int>((reinterpret_cast<size_t>(a.qw) >> 32) & 0xffff),
                           static_cast<int>(wks * static_cast<unsigned>(nk_all)), 0x00020000};
        const int wvo = (nw0 >> 4) * BLK + lane * 16;
        const int wvo8 = (nw0 >> 4) * BLK + 1024 + lane * 8;                       // (F6: the 8-byte pieces of a 1.5 KiB block, behind its sixty-four 16-byte pieces)
        const int wvoB = wvo + 2 * BLK, wvo8B = wvo8 + 2 * BLK;                   // (immediates reach 4095: the FP6 form's fourth block goes through these)
        auto wv16 = [&](int i) MIXQ_INL { return i * BLK < 4096 ? wvo : wvoB; };
        auto wv8 = [&](int i) MIXQ_INL { return i * BLK < 4096 ? wvo8 : wvo8B; };
        auto wimm = [&](int i) MIXQ_INL -> int { return i * BLK < 4096 ? i * BLK : (i - 2) * BLK; };
        unsigned woff = static_cast<unsigned>(kbeg) * wks;                         // byte offset of the k-step the next weight loads read
        const unsigned wlast = static_cast<unsigned>(kbeg + nk - 1) * wks;
        i32x6 wr6[F6R ? D + 1 : 1][F6R ? WNB : 1];                               // F6R: the weight ring as operand tuples (D + 1 slots, as wq below)
        // WRAP forms request on EVERY k-step; once the tile's last k-step has been requested the surplus requests REPEAT it (valid, never
        // consumed, and the lines are L1 / L2-hot: round 3 wrapped around to the tile's first k-steps instead - 3 MB of long-evicted weight
        // blocks re-fetched per launch, counter traffic 1.42x the algorithmic bytes, and a drain behind the loop that waited for them)
        auto wadvance = [&](int cond) MIXQ_INL {                    // once per requested k-step (cond: wave-uniform 0 / 1; F6R requests always)
            if constexpr (WRAP) { (void)cond; const unsigned n = woff + wks; woff = n < wlast ? n : wlast; }
            else if (cond) woff += wks;
        };
        auto wload6 = [&](auto d_c, int i) MIXQ_INL {               // F6R: fragment i of the k-step at woff -> tuple i of ring slot d
            constexpr int d = decltype(d_c)::value;
            if constexpr (F6R && ABLK != 1 && ABLK != 3) {
                const i32x4 rs_ = wrs; const unsigned so_ = woff; const int im_ = wimm(i);   // (named outside the statement: implicit capture does not look into asm operands)
                const int l16 = wv16(i), l8 = wv8(i);
                i32x4 lo; i32x2 hi;
                asm volatile("buffer_load_dwordx4 %0, %2, %4, %5 offen offset:%6\n\tbuffer_load_dwordx2 %1, %3, %4, %5 offen offset:%6"
                             : "=&v"(lo), "=&v"(hi) : "v"(l16), "v"(l8), "s"(rs_), "s"(so_), "i"(im_) : "memory");
                wr6[d][i] = i32x6{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1]};
            }
        };
        // activation fragment (row lm, k-chunk lq) inside a P16X64 block: conflict-free by the layout's swizzle (common.h);
        // F6: the fragment's 16-byte and 8-byte pieces in the LDS image of an R6X128 block (see the loader)
        const int xoff8 = lm * 96 + 64 + ((lq ^ ((lm >> 3) << 1)) << 3);
        const int xoff = F6 ? lm * 96 + ((((FL & 128) ? lq ^ (lm >> 3) : lq)) << 4) : lm * 64 + ((lq ^ ((0 - (lm >> 2)) & 3)) << 4);
        // Weight register ring: NSLOT = D + 1 slots.  k-step kt is consumed from slot kt % NSLOT while the loads of k-step kt + D
        // go into slot (kt + D) % NSLOT - the slot the PREVIOUS k-step freed - so they can be issued anywhere inside the step,
        // one behind every few MFMAs, instead of as a burst at its end (a VMEM issue blocks its wave while the address unit is
        // busy, and the four waves, released by the same barrier, would all burst at the same moment).
        constexpr int NSLOT = D + 1;
        i32x4 wq[NSLOT][WNB];
        i32x4 xf[MB];
        i32x2 wq2[F6 ? NSLOT : 1][F6 ? WNB : 1];                                 // F6: the 8-byte pieces of the 24-byte weight fragments
        // F6: activation fragments as the 6-register operand tuples - a rotating window of XR of a k-step's MB fragments (all MB would
        // not fit next to 6-register weight fragments: 256 registers per wave at 6 waves per CU).  Group j runs on window slot j % XR and
        // requests fragment j + XR behind its last MFMA - of this stage, or of the next one (landed: the k-step's barrier) once j + XR >= MB.
        constexpr int XR = F6 ? ((ABLK == 38 || (FL & 1)) ? MB : (ABLK == 39 ? 2 : (MB < MIXQ_XR ? MB : MIXQ_XR))) : 1;   // (ABLK 38 / 39, tuning build: window of MB / of 2 tuples)
        i32x6 xf6[XR];
        static_assert(!F6 || MB % XR == 0, "window slots must be compile-time");
#pragma unroll
        for (int d = 0; d < NSLOT; ++d)
#pragma unroll
            for (int i = 0; i < WNB; ++i) {
                // the ring's registers are read-write operands of the load statements, so they need a definition in front of the first one:
                // an EMPTY asm output (no instruction; 15 x 4 v_mov in front of the first weight request otherwise).  Ablation builds that
                // never load them get defined, opaque values.
                if constexpr (ABLK == 0 || ABLK == 6 || ABLK == 50 || (ABLK >= 60 && ABLK < 80)) {
                    asm volatile("" : "=v"(wq[d][i]));
                    if constexpr (F6) asm volatile("" : "=v"(wq2[d][i]));
                } else {
                    if constexpr (RNDOPS) { const unsigned h = static_cast<unsigned>(lane) * 0x9E3779B1u + static_cast<unsigned>(d * WNB + i + 1) * 0x85EBCA6Bu; wq[d][i] = i32x4{static_cast<int>(h), static_cast<int>(h * 0x27D4EB2Fu), static_cast<int>(h ^ 0x5bd1e995u), static_cast<int>(h * 0x165667B1u)}; }
                    else wq[d][i] = i32x4{lane, lane, lane, lane};
                    if constexpr (F6) wq2[d][i] = i32x2{lane, lane};
                    asm volatile("" : "+v"(wq[d][i]));
                }
            }
        if constexpr (!F6) {                               // (the fragment registers are read-write operands of the read statements: defined, no instruction)
#pragma unroll
            for (int j = 0; j < MB; ++j) asm volatile("" : "=v"(xf[j]));
        }
        if constexpr (ABLK != 0) {                         // ablation builds: never-loaded operands get defined, opaque values
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                if constexpr (RNDOPS) { const unsigned h = static_cast<unsigned>(lane) * 0xC2B2AE35u + static_cast<unsigned>(j + 1) * 0x9E3779B1u; xf[j] = i32x4{static_cast<int>(h), static_cast<int>(h * 0x27D4EB2Fu), static_cast<int>(h ^ 0x2545F491u), static_cast<int>(h * 0x165667B1u)}; }
                else xf[j] = i32x4{lane, 1, lane, 1};
                asm volatile("" : "+v"(xf[j]));
            }
            if constexpr (F6R) {
                // (the tuple-ring form's ablations: pseudo-random six-bit codes in whatever is never loaded - see RNDOPS)
                auto rnd6 = [&](unsigned salt) MIXQ_INL {
                    const unsigned h = static_cast<unsigned>(lane) * 0x9E3779B1u + salt * 0x85EBCA6Bu;
                    return i32x6{static_cast<int>(h), static_cast<int>(h * 0x27D4EB2Fu), static_cast<int>(h ^ 0x5bd1e995u), static_cast<int>(h * 0x165667B1u),
                                 static_cast<int>(h * 0xC2B2AE35u), static_cast<int>((h >> 3) * 0x2545F491u)};
                };
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int j = 0; j < XR; ++j) { xf6[j] = rnd6(100u + j); asm volatile("" : "+v"(xf6[j])); }
                if constexpr (ABLK == 1 || ABLK == 3) {
#pragma unroll
                    for (int d = 0; d < NSLOT; ++d)
#pragma unroll
                        for (int i = 0; i < WNB; ++i) { wr6[d][i] = rnd6(1u + d * WNB + i); asm volatile("" : "+v"(wr6[d][i])); }
                }
#endif
            } else if constexpr (F6) {
#pragma unroll
                for (int j = 0; j < XR; ++j) xf6[j] = i32x6{lane, 1, lane, 1, 2, 3};
#pragma unroll
                for (int d = 0; d < NSLOT; ++d)
#pragma unroll
                    for (int i = 0; i < WNB; ++i) asm volatile("" : "+v"(wq2[d][i]));
            }
        }
        // The weight loads are inline asm with hand-counted waits: left to the compiler, the loop header of the unrolled k loop
        // gets `s_waitcnt vmcnt(0)` (its counter model merges the preheader and latch states conservatively), which drains the
        // whole register ring every group.  The destination is a READ-WRITE operand and the "do I load at all" test sits INSIDE
        // the statement: the compiler sees one straight-line definition per slot and k-step, never a branch around a load whose
        // merge it might resolve with a v_mov of a register the memory system has not written yet.  A slot's registers are only
        // meaningful after wwait() for that slot.
        auto wload1 = [&](auto d_c, int i, int cond) MIXQ_INL {
            constexpr int d = decltype(d_c)::value;
            if constexpr (F6R) { (void)cond; wload6(d_c, i); }
            else if constexpr (WRAP) {                     // (int8 form with unconditional, wrapping requests: compile-time waits in the tail)
                i32x4& dst = wq[d][i];
                const i32x4 rs_ = wrs; const unsigned so_ = woff; const int im_ = wimm(i);   // (named outside the statement: implicit capture does not look into asm operands)
                const int l16 = wv16(i);
                if constexpr (WPROBE) {                    // (probe: the fragment comes out of the loaders' LDS copy of the tile's weight blocks)
                    const unsigned wa = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) uint8_t*)lds)) + SC_END + (wave * WNB) * 1024 + lane * 16;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(wa), "i"(i * 1024) : "memory");
                } else
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(dst) : "v"(l16), "s"(rs_), "s"(so_), "i"(im_) : "memory");
            }
            else if constexpr (ABLK != 1 && ABLK != 3) {
                const int cs = __builtin_amdgcn_readfirstlane(cond);             // provably wave-uniform for the "s" constraint
                i32x4& dst = wq[d][i];                     // (named outside the statement: implicit capture does not look into asm operands)
                const i32x4 rs_ = wrs; const unsigned so_ = woff; const int im_ = wimm(i);   // (named outside the statement: implicit capture does not look into asm operands)
                const int l16 = wv16(i);
                if constexpr (F6) {
                    i32x2& dst2 = wq2[d][i];
                    const int l8 = wv8(i);
                    asm volatile("s_cmp_eq_u32 %6, 0\n\ts_cbranch_scc1 1f\n\tbuffer_load_dwordx4 %0, %2, %4, %5 offen offset:%7\n\tbuffer_load_dwordx2 %1, %3, %4, %5 offen offset:%7\n1:"
                                 : "+v"(dst), "+v"(dst2) : "v"(l16), "v"(l8), "s"(rs_), "s"(so_), "s"(cs), "i"(im_) : "memory", "scc");
                } else
                asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 1f\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%5\n1:"
                             : "+v"(dst) : "v"(l16), "s"(rs_), "s"(so_), "s"(cs), "i"(im_) : "memory", "scc");
            }
        };
        auto wload1_always = [&](auto d_c, int i) MIXQ_INL {
            constexpr int d = decltype(d_c)::value;
            if constexpr (F6R) wload6(d_c, i);
            else if constexpr (ABLK != 1 && ABLK != 3) {
                i32x4& dst = wq[d][i];
                const i32x4 rs_ = wrs; const unsigned so_ = woff; const int im_ = wimm(i);   // (named outside the statement: implicit capture does not look into asm operands)
                const int l16 = wv16(i);
                if constexpr (F6) {
                    i32x2& dst2 = wq2[d][i];
                    const int l8 = wv8(i);
                    asm volatile("buffer_load_dwordx4 %0, %2, %4, %5 offen offset:%6\n\tbuffer_load_dwordx2 %1, %3, %4, %5 offen offset:%6"
                                 : "+v"(dst), "+v"(dst2) : "v"(l16), "v"(l8), "s"(rs_), "s"(so_), "i"(im_) : "memory");
                } else
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(dst) : "v"(l16), "s"(rs_), "s"(so_), "i"(im_) : "memory");
            }
        };
        // wait until at most CNT loads issued after slot d's are outstanding (vmcnt retires in order); naming the slot's registers
        // as read-write operands makes every MFMA that uses them depend on this statement
        auto wwait = [&](auto d_c, auto cnt_c) MIXQ_INL {
            constexpr int d = decltype(d_c)::value, CNT = decltype(cnt_c)::value;
            if constexpr (F6R) {
#if defined(__HIP_DEVICE_COMPILE__)                  // (192-bit "v" operands: device pass only)
                if constexpr (WNB == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wr6[d][0]) : "i"(CNT));
                if constexpr (WNB == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wr6[d][0]), "+v"(wr6[d][1]) : "i"(CNT));
                if constexpr (WNB == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(wr6[d][0]), "+v"(wr6[d][1]), "+v"(wr6[d][2]) : "i"(CNT));
                if constexpr (WNB == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wr6[d][0]), "+v"(wr6[d][1]), "+v"(wr6[d][2]), "+v"(wr6[d][3]) : "i"(CNT));
#endif
                __builtin_amdgcn_sched_barrier(0);
            } else
            if constexpr (ABLK != 1 && ABLK != 3) {
                if constexpr (WNB == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wq[d][0]) : "i"(CNT));
                if constexpr (WNB == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wq[d][0]), "+v"(wq[d][1]) : "i"(CNT));
                if constexpr (WNB == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]) : "i"(CNT));
                if constexpr (WNB == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]), "+v"(wq[d][3]) : "i"(CNT));
                if constexpr (F6) {                        // (volatile statements keep their order: the 8-byte pieces are "defined" behind the wait as well)
#pragma unroll
                    for (int i = 0; i < WNB; ++i) asm volatile("" : "+v"(wq2[d][i]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // the same with a run-time count (tail of the k loop): CNT in {0, WNB, 2 WNB, ...}, selected inside ONE statement
        auto wwait_rt = [&](auto d_c, int younger) MIXQ_INL {
            constexpr int d = decltype(d_c)::value;
            if constexpr (WRAP) { (void)younger; wwait(d_c, std::integral_constant<int, WL * (D - 1)>{}); }   // (every k-step requests: the count is fixed)
            else if constexpr (ABLK != 1 && ABLK != 3) {
                const int sel = __builtin_amdgcn_readfirstlane(younger >= D - 1 ? D - 1 : younger);   // k-steps requested after this one, capped at the ring depth
#define MIXQ_WR_W1(n, l) "s_cmp_lt_u32 %[sel], " #n "\n\ts_cbranch_scc1 " #l "f\n\t"
#define MIXQ_WR_W2(l, c) #l ":\n\ts_waitcnt vmcnt(%[" #c "])\n\ts_branch 199f\n"
#define MIXQ_WR_WAITS                                                                                                           \
                MIXQ_WR_W1(1, 100) MIXQ_WR_W1(2, 101) MIXQ_WR_W1(3, 102) MIXQ_WR_W1(4, 103) MIXQ_WR_W1(5, 104) MIXQ_WR_W1(6, 105)            \
                MIXQ_WR_W1(7, 106) MIXQ_WR_W1(8, 107) MIXQ_WR_W1(9, 108) MIXQ_WR_W1(10, 109) MIXQ_WR_W1(11, 110) MIXQ_WR_W1(12, 111)         \
                MIXQ_WR_W1(13, 112) MIXQ_WR_W1(14, 113) MIXQ_WR_W1(15, 114) "s_waitcnt vmcnt(%[c15])\n\ts_branch 199f\n"                   \
                "100:\n\ts_waitcnt vmcnt(0)\n\ts_branch 199f\n" MIXQ_WR_W2(101, c1) MIXQ_WR_W2(102, c2) MIXQ_WR_W2(103, c3)               \
                MIXQ_WR_W2(104, c4) MIXQ_WR_W2(105, c5) MIXQ_WR_W2(106, c6) MIXQ_WR_W2(107, c7) MIXQ_WR_W2(108, c8) MIXQ_WR_W2(109, c9)      \
                MIXQ_WR_W2(110, c10) MIXQ_WR_W2(111, c11) MIXQ_WR_W2(112, c12) MIXQ_WR_W2(113, c13) MIXQ_WR_W2(114, c14) "199:"
#define MIXQ_WR_CL(v) ((v) < 64 ? (v) : 63)
#define MIXQ_WR_WAIT_IN [sel] "s"(sel), [c1] "i"(MIXQ_WR_CL(WL)), [c2] "i"(MIXQ_WR_CL(2 * WL)), [c3] "i"(MIXQ_WR_CL(3 * WL)),               \
                [c4] "i"(MIXQ_WR_CL(4 * WL)), [c5] "i"(MIXQ_WR_CL(5 * WL)), [c6] "i"(MIXQ_WR_CL(6 * WL)), [c7] "i"(MIXQ_WR_CL(7 * WL)),      \
                [c8] "i"(MIXQ_WR_CL(8 * WL)), [c9] "i"(MIXQ_WR_CL(9 * WL)), [c10] "i"(MIXQ_WR_CL(10 * WL)), [c11] "i"(MIXQ_WR_CL(11 * WL)), \
                [c12] "i"(MIXQ_WR_CL(12 * WL)), [c13] "i"(MIXQ_WR_CL(13 * WL)), [c14] "i"(MIXQ_WR_CL(14 * WL)), [c15] "i"(MIXQ_WR_CL(15 * WL))
                if constexpr (WNB == 1) asm volatile(MIXQ_WR_WAITS : "+v"(wq[d][0]) : MIXQ_WR_WAIT_IN : "scc");
                if constexpr (WNB == 2) asm volatile(MIXQ_WR_WAITS : "+v"(wq[d][0]), "+v"(wq[d][1]) : MIXQ_WR_WAIT_IN : "scc");
                if constexpr (WNB == 3) asm volatile(MIXQ_WR_WAITS : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]) : MIXQ_WR_WAIT_IN : "scc");
                if constexpr (WNB == 4) asm volatile(MIXQ_WR_WAITS : "+v"(wq[d][0]), "+v"(wq[d][1]), "+v"(wq[d][2]), "+v"(wq[d][3]) : MIXQ_WR_WAIT_IN : "scc");
                if constexpr (F6) {
#pragma unroll
                    for (int i = 0; i < WNB; ++i) asm volatile("" : "+v"(wq2[d][i]));
                }
#undef MIXQ_WR_WAITS
#undef MIXQ_WR_WAIT_IN
#undef MIXQ_WR_W1
#undef MIXQ_WR_W2
#undef MIXQ_WR_CL
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // (round 6) The activation fragment reads are inline asm with hand-counted waits, like the weight loads.  Left to the compiler, every
        // wait for a fragment came out as `s_waitcnt lgkmcnt(0)`: at the k-step boundary of the int8 loop (the re-read of fragment MB - 1,
        // issued a few cycles earlier, was an exposed LDS round trip per k-step) and in front of groups 0 and XR of the FP6 loop (176 and 148
        // cycles for those two groups against ~80 for the others, profiles/r06_w4a4_kstep_timeline_old_lds_image.txt).  The LDS counter
        // retires in order, so group j waits for "at most xyounger(j) DS operations behind my fragment's": the reads issued since.
        // The destinations are READ-WRITE operands (the register stays the fragment's own); xdrain() behind the loops names them all.
        const unsigned xbase32 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) uint8_t*)lds)) + static_cast<unsigned>(xoff);
        const unsigned xbase32_8 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) uint8_t*)lds)) + static_cast<unsigned>(xoff8);
        auto xread = [&](int slot, int j) MIXQ_INL {
            if constexpr (F6 && (ABLK == 2 || ABLK == 3 || ABLK == 32)) {
                // (ablation: no LDS reads)
            } else if constexpr (F6) {
#if defined(__HIP_DEVICE_COMPILE__)
                const unsigned a16 = xbase32 + static_cast<unsigned>(slot * STAGE_BYTES), a8 = xbase32_8 + static_cast<unsigned>(slot * STAGE_BYTES);
                i32x4 p4; i32x2 p2;
                // (EARLY-CLOBBER outputs: two instructions in one statement - without it the allocator put the first read's destination over the second read's
                // address register, the one it considered dead behind the statement (wr32x64: wrong tiles whenever the second read was held back behind a full LDS queue))
                asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4" : "=&v"(p4), "=&v"(p2) : "v"(a16), "v"(a8), "i"(j * BLK) : "memory");
                xf6[j % XR] = i32x6{p4[0], p4[1], p4[2], p4[3], p2[0], p2[1]};
#endif
            } else
            if constexpr (ABLK != 2 && ABLK != 3) {
                const unsigned a16 = xbase32 + static_cast<unsigned>(slot * STAGE_BYTES);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(xf[j]) : "v"(a16), "i"(j * 1024) : "memory");
            }
        };
        // One wait serves XWG consecutive groups: a wait (and the wait state the compiler puts behind a statement that "writes" an MFMA operand)
        // in front of EVERY group cost the int8 loop 5 % of the launch - the consumer waves are bound by what they issue between their MFMAs
        // (profiles/r06_xwait_bisect.txt) - and one drain per k-step costs the int8 loop nothing measurable.  The FP6 forms' window of XR tuples
        // needs a wait every XR groups at least; theirs is counted, so the youngest read it waits for is XR - XWG groups old.
        constexpr int XWG_SEL = (FL >> 8) & 3;                                    // (tuning build: 1, 2, 3 -> a wait every 1, 2, 4 groups)
        constexpr int XWG_DEF = F6 ? (XR < 2 ? XR : 2) : MB;
        constexpr int XWG0 = ABLK == 73 ? 4 : (ABLK == 74 ? 2 : (ABLK == 75 ? 1 : (XWG_SEL == 1 ? 1 : (XWG_SEL == 2 ? 2 : (XWG_SEL == 3 ? 4 : XWG_DEF)))));
        constexpr int XWG = XWG0 > (F6 ? XR : MB) ? (F6 ? XR : MB) : XWG0;
        static_assert((F6 ? XR : MB) % XWG == 0, "a wait serves whole groups of the fragment window");
        // DS operations issued behind the read that filled group jt's fragment and in front of group j0 <= jt, in a k-step that re-reads from the
        // next stage (REFILL) or not (the tile's last k-step).  int8 forms: fragment j is re-read behind group j (one operation).  FP6 forms: group
        // g requests fragment g + XR - of this stage while g + XR < MB, of the next one (REFILL only) beyond - two operations each.
        auto xyounger = [&](int j0, int jt, bool refill) MIXQ_INL -> int {
            int n = 0;
            for (int g = jt - (F6 ? XR : MB) + 1; g < j0; ++g) n += (g < 0 || (F6 && g + XR < MB) || refill) ? 1 : 0;
            return F6 ? 2 * n : n;
        };
        // the fragments of groups j .. j + XWG - 1 have landed (a statement in front of group j when j is a multiple of XWG; nothing otherwise)
        auto xwait = [&](int j, bool refill) MIXQ_INL {
            if constexpr (F6 && (ABLK == 2 || ABLK == 3 || ABLK == 32)) { (void)j; (void)refill; }
            else if (j % XWG == 0) {
                const int jt = j + XWG - 1;
                if constexpr (F6) {
#if defined(__HIP_DEVICE_COMPILE__)
                    if constexpr (XWG == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf6[j % XR]) : "i"(xyounger(j, jt, refill)));
                    if constexpr (XWG == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xf6[j % XR]), "+v"(xf6[(j + 1) % XR]) : "i"(xyounger(j, jt, refill)));
                    if constexpr (XWG == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(xf6[j % XR]), "+v"(xf6[(j + 1) % XR]), "+v"(xf6[(j + 2) % XR]), "+v"(xf6[(j + 3) % XR]) : "i"(xyounger(j, jt, refill)));
#endif
                } else if constexpr (ABLK != 2 && ABLK != 3) {
                    if constexpr (XWG == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf[j]) : "i"(xyounger(j, jt, refill)));
                    if constexpr (XWG == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xf[j]), "+v"(xf[j + 1]) : "i"(xyounger(j, jt, refill)));
                    if constexpr (XWG == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(xf[j]), "+v"(xf[j + 1]), "+v"(xf[j + 2]), "+v"(xf[j + 3]) : "i"(xyounger(j, jt, refill)));
                    if constexpr (XWG == 8) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(xf[j]), "+v"(xf[j + 1]), "+v"(xf[j + 2]), "+v"(xf[j + 3]), "+v"(xf[j + 4]), "+v"(xf[j + 5]), "+v"(xf[j + 6]), "+v"(xf[j + 7]) : "i"(xyounger(j, jt, refill)));
                    if constexpr (XWG == 16) {
                        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(xf[j]), "+v"(xf[j + 1]), "+v"(xf[j + 2]), "+v"(xf[j + 3]), "+v"(xf[j + 4]), "+v"(xf[j + 5]), "+v"(xf[j + 6]), "+v"(xf[j + 7]) : "i"(xyounger(j, jt, refill)));
                        asm volatile("" : "+v"(xf[j + 8]), "+v"(xf[j + 9]), "+v"(xf[j + 10]), "+v"(xf[j + 11]), "+v"(xf[j + 12]), "+v"(xf[j + 13]), "+v"(xf[j + 14]), "+v"(xf[j + 15]));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // (tail k-steps: `refill` is a run-time value, the counts are not.  The choice sits INSIDE one statement: a branch around two wait
        // statements made the compiler merge the fragment's register through copies, one of them in front of its wait - tools/check_wreg_asm.py)
        auto xwait_rt = [&](int j, bool refill) MIXQ_INL {
            const int rf = __builtin_amdgcn_readfirstlane(refill ? 1 : 0);
#define MIXQ_XWAIT_RT "s_cmp_eq_u32 %[rf], 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt lgkmcnt(%[a])\n\ts_branch 2f\n1:\n\ts_waitcnt lgkmcnt(%[b])\n2:"
#define MIXQ_XWAIT_IN [rf] "s"(rf), [a] "i"(xyounger(j, jt, true)), [b] "i"(xyounger(j, jt, false))
            if constexpr (F6 && (ABLK == 2 || ABLK == 3 || ABLK == 32)) { (void)j; (void)rf; }
            else if (j % XWG == 0) {
                const int jt = j + XWG - 1;
                if constexpr (F6) {
#if defined(__HIP_DEVICE_COMPILE__)
                    if constexpr (XWG == 1) asm volatile(MIXQ_XWAIT_RT : "+v"(xf6[j % XR]) : MIXQ_XWAIT_IN : "scc");
                    if constexpr (XWG == 2) asm volatile(MIXQ_XWAIT_RT : "+v"(xf6[j % XR]), "+v"(xf6[(j + 1) % XR]) : MIXQ_XWAIT_IN : "scc");
                    if constexpr (XWG == 4) asm volatile(MIXQ_XWAIT_RT : "+v"(xf6[j % XR]), "+v"(xf6[(j + 1) % XR]), "+v"(xf6[(j + 2) % XR]), "+v"(xf6[(j + 3) % XR]) : MIXQ_XWAIT_IN : "scc");
#endif
                } else if constexpr (ABLK != 2 && ABLK != 3) {
                    if constexpr (XWG == 1) asm volatile(MIXQ_XWAIT_RT : "+v"(xf[j]) : MIXQ_XWAIT_IN : "scc");
                    if constexpr (XWG == 2) asm volatile(MIXQ_XWAIT_RT : "+v"(xf[j]), "+v"(xf[j + 1]) : MIXQ_XWAIT_IN : "scc");
                    if constexpr (XWG == 4) asm volatile(MIXQ_XWAIT_RT : "+v"(xf[j]), "+v"(xf[j + 1]), "+v"(xf[j + 2]), "+v"(xf[j + 3]) : MIXQ_XWAIT_IN : "scc");
                    if constexpr (XWG == 8) asm volatile(MIXQ_XWAIT_RT : "+v"(xf[j]), "+v"(xf[j + 1]), "+v"(xf[j + 2]), "+v"(xf[j + 3]), "+v"(xf[j + 4]), "+v"(xf[j + 5]), "+v"(xf[j + 6]), "+v"(xf[j + 7]) : MIXQ_XWAIT_IN : "scc");
                    if constexpr (XWG == 16) {
                        asm volatile(MIXQ_XWAIT_RT : "+v"(xf[j]), "+v"(xf[j + 1]), "+v"(xf[j + 2]), "+v"(xf[j + 3]), "+v"(xf[j + 4]), "+v"(xf[j + 5]), "+v"(xf[j + 6]), "+v"(xf[j + 7]) : MIXQ_XWAIT_IN : "scc");
                        asm volatile("" : "+v"(xf[j + 8]), "+v"(xf[j + 9]), "+v"(xf[j + 10]), "+v"(xf[j + 11]), "+v"(xf[j + 12]), "+v"(xf[j + 13]), "+v"(xf[j + 14]), "+v"(xf[j + 15]));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef MIXQ_XWAIT_RT
#undef MIXQ_XWAIT_IN
        };
        auto xdrain = [&]() MIXQ_INL {
            if constexpr (F6) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int j = 0; j < XR; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf6[j]));
#endif
            } else {
#pragma unroll
                for (int j = 0; j < MB; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[j]));
            }
        };
        uint32_t nib = 0xf0f0f0f0u;
        if constexpr (I4) asm volatile("s_mov_b32 %0, 0xf0f0f0f0" : "=s"(nib));
        auto lo4 = [&](i32x4 v) MIXQ_INL { i32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = static_cast<int>((static_cast<uint32_t>(v[e]) << 4) & nib);
            return o; };
        auto hi4 = [&](i32x4 v) MIXQ_INL { i32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = static_cast<int>(static_cast<uint32_t>(v[e]) & nib);
            return o; };

        unsigned long long kst[11] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};   // ABLK 36: see the loader's lstamp
        const int kmark = nk >> 1;
        int kcur = -1;                                                           // (the k-step `one` is running)
        auto kstamp = [&](int i) MIXQ_INL {
#if defined(MIXQ_TUNING) && defined(__HIP_DEVICE_COMPILE__)
            if constexpr (TL) { const int ku = __builtin_amdgcn_readfirstlane(kcur); asm volatile("s_cmp_lg_u32 %1, %2\n\ts_cbranch_scc1 1f\n\ts_memtime %0\n1:" : "+s"(kst[i]) : "s"(ku), "s"(kmark) : "scc"); }
#else
            (void)i;
#endif
        };
        // one k-step: MFMAs of ring slot C, X fragment j re-read from stage kt+1 right behind its last MFMA (REFILL), weight load
        // i of k-step kt+D issued into slot L = (C + D) % NSLOT behind MFMA group WPOS(i) (FULL: unconditionally)
        auto step = [&](auto c_c, auto full_c, bool refill, int issue, int rslot) MIXQ_INL {
            constexpr int C = decltype(c_c)::value, L = (C + D) % NSLOT;
            constexpr bool FULL = decltype(full_c)::value;
            using LC = std::integral_constant<int, L>;
            auto loads_behind = [&](int j) MIXQ_INL {                                     // j: compile-time after unrolling
#pragma unroll
                for (int i = 0; i < WNB; ++i) {
                    int pos = (WNB >= MB) ? (i % MB) : ((2 * i + 1) * MB) / (2 * WNB);      // spread evenly over the MB groups
                    if constexpr (ABLK == 20 || ABLK == 23) pos = MB - WNB + i;            // probe: behind the last groups (away from the loaders' burst after the barrier)
                    if constexpr (ABLK == 21) pos = (i * MB) / WNB;                        // probe: 0, 2, 5: earlier
                    if (pos == j) { if constexpr (FULL) wload1_always(LC{}, i); else wload1(LC{}, i, issue); }
                }
            };
            if constexpr (F6R) {
                const int unit = 0x7f7f7f7f;                                       // E8M0 block scales of 2^0
#if defined(__HIP_DEVICE_COMPILE__)
#define MIXQ_F6_MMA(ACC, W, X) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:3 blgp:3" \
                                            : "+v"(ACC) : "v"(W), "v"(X), "v"(unit))
#else
#define MIXQ_F6_MMA(ACC, W, X) (void)unit
#endif
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    if constexpr (FULL) xwait(j, true); else xwait_rt(j, refill);
                    MIXQ_F6_MMA(acc[j][0], wr6[C][0], xf6[j % XR]);
                    __builtin_amdgcn_sched_barrier(0);
                    loads_behind(j);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 1; i < WNB; ++i) MIXQ_F6_MMA(acc[j][i], wr6[C][i], xf6[j % XR]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + XR < MB) xread(rslot == 0 ? NSTAGE - 1 : rslot - 1, j + XR);
                    else if (refill) xread(rslot, j + XR - MB);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (TL) { if (j < 8) kstamp(3 + j); __builtin_amdgcn_sched_barrier(0); }
                }
#undef MIXQ_F6_MMA
            } else if constexpr (F6) {
                // Same order as the int8 loop: weight load behind the group's first MFMA, the fragment's re-read behind its last.
                // The MFMA is inline asm: the compiler's form of the scaled MFMA is not tied (vdst != srcC, accumulators rotating through
                // VGPRs), which runs at 32 cycles instead of 19.5 (tools/ubench_fp6.hip) and doubles the accumulator footprint.  In asm
                // the accumulators stay in place in the AGPR half.  The hazard recogniser does not look into asm, so: the weight
                // tuples are assembled (v_mov) and pinned BEFORE a scheduling barrier and two wait states; activation tuples come
                // straight from LDS reads (waited for by the compiler: the asm names them as inputs); consecutive MFMAs never share
                // an accumulator (24 MFMAs lie between two uses of one).  Accumulators are VGPRs ("+v", in place): with an AGPR class in
                // play the allocator splits the 256 registers of a wave 128 / 128, and 128 VGPRs do not hold the rings.
                i32x6 w6[WNB];
#pragma unroll
                for (int i = 0; i < WNB; ++i) {
                    w6[i] = i32x6{wq[C][i][0], wq[C][i][1], wq[C][i][2], wq[C][i][3], wq2[C][i][0], wq2[C][i][1]};
#if defined(__HIP_DEVICE_COMPILE__)                  // (the host pass parses kernel bodies too and knows no 192-bit "v" operand)
                    asm volatile("" : "+v"(w6[i]));
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 1" ::: "memory");
                const int unit = 0x7f7f7f7f;                                       // E8M0 block scales of 2^0
#if defined(__HIP_DEVICE_COMPILE__)
#define MIXQ_F6_MMA(ACC, W, X) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:3 blgp:3" \
                                            : "+v"(ACC) : "v"(W), "v"(X), "v"(unit))
#else
#define MIXQ_F6_MMA(ACC, W, X) (void)unit
#endif
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    if constexpr (FULL) xwait(j, true); else xwait_rt(j, refill);
                    MIXQ_F6_MMA(acc[j][0], w6[0], xf6[j % XR]);
                    __builtin_amdgcn_sched_barrier(0);
                    loads_behind(j);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 1; i < WNB; ++i) MIXQ_F6_MMA(acc[j][i], w6[i], xf6[j % XR]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + XR < MB) xread(rslot == 0 ? NSTAGE - 1 : rslot - 1, j + XR);   // (this stage: the slot before stage kt+1's)
                    else if (refill) xread(rslot, j + XR - MB);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef MIXQ_F6_MMA
            } else if constexpr (!I4) {
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    if constexpr (FULL) xwait(j, true); else xwait_rt(j, refill);
                    constexpr bool NEW_ORDER = ABLK != 31;                        // (31, tuning build: round 2's order)
                    if constexpr (P32) {
                        // group g: k-half g >> 2 of row block g & 3 (a 32 x 32 accumulator is touched again four groups = eight MFMAs later)
                        const int kh = j >> 2, mb32 = j & 3, f = mb32 * 2 + kh;
#if defined(__HIP_DEVICE_COMPILE__)                  // (with this builtin in sight the HOST pass drops the kernel's stub without a diagnostic)
                        acc32[mb32][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wq[C][kh], xf[f], acc32[mb32][0], 0, 0, 0);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        loads_behind(j);
                        __builtin_amdgcn_sched_barrier(0);
#if defined(__HIP_DEVICE_COMPILE__)
                        acc32[mb32][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wq[C][2 + kh], xf[f], acc32[mb32][1], 0, 0, 0);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        if (refill) xread(rslot, f);
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                    if constexpr (ABLK == 79) {
                        accs[j][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[C][0], xf[j], accs[j][0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        loads_behind(j);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 1; i < WNB; ++i)
                            accs[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[C][i], xf[j], accs[j][i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (refill) xread(rslot, j);
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                    if constexpr (NEW_ORDER) {
                        // At most ONE memory instruction per MFMA gap: a wave issues in order and a 16-cycle MFMA leaves ~12 cycles in
                        // which one fragment read or one weight load can be issued for free.  The weight load goes behind the group's
                        // FIRST MFMA, the fragment's re-read behind its last use (round 2 put both behind the last).  Measured: -1.6 %
                        // against the old order compiled into the same (tuning) library, -1.8 % / -1.5 % at N = 11008 / 12288 between
                        // two product builds loaded side by side (profiles/r03_gemm_ab_issue_timing.txt, r03_ab_order_builds.txt).
                        acc[j][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[C][0], xf[j], acc[j][0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        loads_behind(j);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 1; i < WNB; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[C][i], xf[j], acc[j][i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (refill) xread(rslot, j);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (TL) { if (j < 8) kstamp(3 + j); __builtin_amdgcn_sched_barrier(0); }
                        continue;
                    }
#pragma unroll
                    for (int i = 0; i < WNB; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[C][i], xf[j], acc[j][i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (refill) xread(rslot, j);
                    loads_behind(j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // nibbles -> int8 values 16 q: even columns (low nibbles) and odd columns (high nibbles) form two fragments,
                // split identically on both operands so the k pairing is preserved; the factor 256 leaves in the epilogue
                i32x4 wl[WNB], wh[WNB];
#pragma unroll
                for (int i = 0; i < WNB; ++i) { wl[i] = lo4(wq[C][i]); wh[i] = hi4(wq[C][i]); }
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    if constexpr (FULL) xwait(j, true); else xwait_rt(j, refill);
                    const i32x4 xl = lo4(xf[j]), xh = hi4(xf[j]);
#pragma unroll
                    for (int i = 0; i < WNB; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wl[i], xl, acc[j][i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < WNB; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wh[i], xh, acc[j][i], 0, 0, 0);
                    if (refill) xread(rslot, j);
                    loads_behind(j);
                }
            }
            wadvance(FULL ? 1 : issue);
        };

        // the epilogue's scales first: the oldest loads of this wave, so they never sit between the hand-counted weight loads
        // (the fat prefill form requests them after its k loop instead: 24 registers held across the loop would be spilled there)
        auto load_scales = [&]() MIXQ_INL {
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const int m = m0 + j * 16 + lm;
                sxh[j] = a.sx[m < a.M ? m : a.M - 1];
            }
#pragma unroll
            for (int i = 0; i < WNB; ++i) {
                const int n = nw0 + i * 16 + lq * 4;                             // N % 4 == 0: 4 columns are all in or all out
                swp[i] = *reinterpret_cast<const u32x2_u*>(a.sw + (n < a.N ? n : a.N - 4));
                // (in the epilogue this load is an exposed round trip: 1.7 us per forward with a bias; tiles without 8 spare registers keep it there)
                if constexpr (PREBIAS) bvp[i] = a.bias ? *reinterpret_cast<const u32x2_u*>(a.bias + (n < a.N ? n : a.N - 4)) : u32x2{0u, 0u};
            }
        };
        if constexpr (SELF) {
            // ---- self-loading k loop (prefill form) -------------------------------------------------------------------------
            // Every k-step, in this program order: MB MFMA groups; behind group i*MB/LOADS this wave's piece i of X stage kt+LOOK
            // (LDS-DMA), behind group ((2i+1) MB)/(2 WNB) weight load i of k-step kt+D; fragment j re-read from stage kt+1 behind its
            // last MFMA.  LOOK = D + 1: stage kt+1 and the weights of k-step kt were requested in the SAME k-step kt-D, pieces first,
            // so ONE counted wait - at most (D-1)(WNB+LOADS) younger requests in flight - retires both (vmcnt retires in order).
            // Requests past the end of K wrap around to the first k-steps (valid addresses, L2-hot, never consumed): every k-step is
            // the same straight-line code, nothing conditional sits between the hand-counted loads.
            const uint8_t* xsrc[LOADS];
            int xdst[LOADS];
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int p = wave + i * CW;
                int rb = (m0 >> 4) + p; rb = rb < a.xblocks ? rb : a.xblocks - 1;
                xsrc[i] = a.qx + static_cast<size_t>(rb) * 1024 + lane * 16;
                xdst[i] = p * 1024;
            }
            const size_t xks = static_cast<size_t>(a.xblocks) * 1024;
            int xkq = 0;
            size_t xoff_g = 0;
            auto xpiece = [&](int slot, int i) MIXQ_INL { wr_glds16(xsrc[i] + xoff_g, lds + slot * STAGE_BYTES + xdst[i]); };
            auto xadvance = [&]() MIXQ_INL { xoff_g += xks; if (++xkq == nk) { xkq = 0; xoff_g = 0; } };
            constexpr int INFLIGHT = (D - 1) * (WNB + LOADS);
            // prologue = the k-steps -D .. -1 of the steady state, preceded by stage 0
#pragma unroll
            for (int i = 0; i < LOADS; ++i) xpiece(0, i);
            xadvance();
            wr_static_for<0, D>([&](auto d_c) MIXQ_INL {
                constexpr int d = decltype(d_c)::value;
#pragma unroll
                for (int i = 0; i < LOADS; ++i) xpiece((d + 1) % NSTAGE, i);
                xadvance();
#pragma unroll
                for (int i = 0; i < WNB; ++i) wload1_always(d_c, i);
                wadvance(1);
            });
            wr_wait_vmcnt<INFLIGHT>();                                           // own pieces of stages 0 and 1, weights of k-step 0
            __builtin_amdgcn_s_barrier();                                        // B0: everybody's pieces of stage 0 (and 1) landed
            stamp(1);
#pragma unroll
            for (int j = 0; j < MB; ++j) xread(0, j);
            int kt = 0, slot1 = 1 % NSTAGE, nxt = LOOK % NSTAGE;                 // ring slots of stage kt+1 and stage kt+LOOK
            auto one = [&](auto c_c) MIXQ_INL {
                constexpr int C = decltype(c_c)::value, L = (C + D) % NSLOT;
                using LC = std::integral_constant<int, L>;
                wwait(c_c, std::integral_constant<int, INFLIGHT>{});             // weights of k-step kt and own pieces of stage kt+1 landed
                __builtin_amdgcn_s_barrier();                                    // everybody's: stage kt+1 readable, stage kt-2's slot free
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    xwait(j, true);                                               // (this form re-reads on every k-step: the wrapped-around stages)
#pragma unroll
                    for (int i = 0; i < WNB; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[C][i], xf[j], acc[j][i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    xread(slot1, j);
#pragma unroll
                    for (int i = 0; i < LOADS; ++i)
                        if ((i * MB) / LOADS == j) xpiece(nxt, i);
#pragma unroll
                    for (int i = 0; i < WNB; ++i)
                        if (((2 * i + 1) * MB) / (2 * WNB) == j) wload1_always(LC{}, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
                wadvance(1);
                xadvance();
                slot1 = (NSTAGE & (NSTAGE - 1)) == 0 ? ((slot1 + 1) & (NSTAGE - 1)) : ((slot1 + 1 == NSTAGE) ? 0 : slot1 + 1);
                nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
                ++kt;
            };
            while (kt + NSLOT <= nk) wr_static_for<0, NSLOT>([&](auto c_c) MIXQ_INL { one(c_c); });
            {
                const int k0 = kt;                                               // (a group is entered at ring slot 0)
                wr_static_for<0, NSLOT>([&](auto c_c) MIXQ_INL { if (k0 + decltype(c_c)::value < nk) one(c_c); });
            }
            wr_wait_vmcnt<0>();                                                  // the wrapped-around requests of the last k-steps: nothing may
            xdrain();                                                            // land in a register the epilogue re-uses
            load_scales();
            stamp(2);
        } else {
        // ---- prologue ------------------------------------------------------------------------------------------------
        auto prologue_w = [&](auto d_c) MIXQ_INL {
#pragma unroll
            for (int i = 0; i < WNB; ++i) wload1(d_c, i, decltype(d_c)::value < nk ? 1 : 0);
            wadvance(decltype(d_c)::value < nk ? 1 : 0);
            // probes: hold the requests of k-steps 1.. back so every CU's first k-step is served first
            if constexpr (ABLK == 25 || ABLK == 26) { if (decltype(d_c)::value == 0) __builtin_amdgcn_s_sleep(8); }
            if constexpr (ABLK == 27) { if (decltype(d_c)::value == 0) __builtin_amdgcn_s_sleep(16); }
        };
        // Request order: the weights of k-step 0 FIRST - they and the loaders' first stage are what the first MFMA waits for - then the
        // epilogue's scales (round 3 asked for those before anything else: ~30 address instructions and 14 requests in front of the first
        // weight request), then the ring's other k-steps, then the accumulators are zeroed under all of it.  The counted waits still hold:
        // vmcnt retires in order and k-step 0's wait leaves WL (D - 1) requests in flight - exactly the k-steps 1 .. D-1 issued behind
        // the scales - so the scales have landed with k-step 0 and never sit among the requests a later wait counts.
        prologue_w(std::integral_constant<int, 0>{});
        if constexpr (!EPI2) load_scales();                                      // (EPI2: the scales travel through LDS, brought by a loader wave)
        wr_static_for<1, D>(prologue_w);
        zero_acc();
        static_assert(D >= 2 && D <= 16 && WL * (D - 1) < 64, "weight ring depth: run-time wait table / vmcnt range");
        __builtin_amdgcn_s_barrier();                                            // B0
        stamp(1);
        // every scalar (kernel-argument) load has long returned; telling the compiler's counter model so keeps the loop
        // header from merging "SMEM pending" with the LDS reads in flight into an lgkmcnt(0) per group
        __builtin_amdgcn_s_waitcnt(0xC07F);                                      // lgkmcnt(0)
#pragma unroll
        for (int j = 0; j < (F6 ? XR : MB); ++j) xread(0, j);

        // ---- k loop: unrolled by NSLOT so ring slots are compile-time registers ----------------------------------------
        int kt = 0, slot1 = 1 % NSTAGE;                                          // ring slot of stage kt+1
        auto one = [&](auto c_c, auto full_c) MIXQ_INL {
            constexpr bool FULL = decltype(full_c)::value;                       // FULL: stage kt+1 and k-step kt+D exist
            if constexpr (FULL) {
                kcur = kt;
                kstamp(0);
                if constexpr (ABLK != 4) wwait(c_c, std::integral_constant<int, WL * (D - 1)>{});    // the D-1 younger k-steps stay in flight
                kstamp(1);
                if constexpr (ABLK != 6) __builtin_amdgcn_s_barrier();            // stage kt+1 landed; stage kt-2's slot is free
                kstamp(2);
                step(c_c, full_c, true, 1, slot1);
            } else {
                wwait_rt(c_c, nk - 1 - kt);
                const bool more = kt + 1 < nk;
                if (more && ABLK != 6) __builtin_amdgcn_s_barrier();
                step(c_c, full_c, more, kt + D < nk ? 1 : 0, slot1);
            }
            slot1 = (NSTAGE & (NSTAGE - 1)) == 0 ? ((slot1 + 1) & (NSTAGE - 1)) : ((slot1 + 1 == NSTAGE) ? 0 : slot1 + 1);
            ++kt;
        };
        auto group = [&](auto full_c) MIXQ_INL { wr_static_for<0, NSLOT>([&](auto c_c) MIXQ_INL { one(c_c, full_c); }); };
        while (kt + NSLOT + D <= nk) group(std::true_type{});                    // every k-step of the group has kt + D < nk
        while (kt < nk) {                                                        // fewer than NSLOT + D k-steps, guarded individually
            // (a group is entered at ring slot 0, so slot c holds k-step kt + c here as well)
            const int k0 = kt;
            auto tail_one = [&](auto c_c) MIXQ_INL { if (k0 + decltype(c_c)::value < nk) one(c_c, std::false_type{}); };
            wr_static_for<0, NSLOT>(tail_one);
        }
        // The wrapped-around requests of the last k-steps are never consumed: to the compiler their destination registers are dead the
        // moment they are issued - free for anything - while the loads are still in flight.  The drain therefore NAMES every ring slot
        // (read-write operands of the wait statements): the ring stays allocated until nothing can land in it any more.
        if constexpr (WRAP) wr_static_for<0, NSLOT>([&](auto d_c) MIXQ_INL { wwait(d_c, std::integral_constant<int, 0>{}); });
        xdrain();                                                                // (every issued fragment read has been consumed; this names the registers once more)
        if constexpr (F6) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");       // asm MFMA -> accumulator reads: wait states the compiler cannot count
#ifdef MIXQ_TUNING
        if constexpr (TL) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kst[0]), "+s"(kst[1]), "+s"(kst[2]), "+s"(kst[3]), "+s"(kst[4]), "+s"(kst[5]), "+s"(kst[6]), "+s"(kst[7]), "+s"(kst[8]), "+s"(kst[9]), "+s"(kst[10]));
            if (a.trace && wave == TLW && lane == 0) {
#pragma unroll
                for (int i = 0; i < 11; ++i) a.trace[2 * 16 * 4096 + blockIdx.x * 16 + i] = kst[i];
            }
        }
#endif
        if constexpr (P32) {
#pragma unroll
#if defined(__HIP_DEVICE_COMPILE__)                  // (512-bit "v" operands: device pass only)
            for (int m = 0; m < 4; ++m) asm volatile("" :: "v"(acc32[m][0]), "v"(acc32[m][1]));
#else
            for (int m = 0; m < 4; ++m) (void)acc32[m][0];
#endif
        }
        if constexpr (ABLK == 79) {
#pragma unroll
            for (int j = 0; j < MB; ++j) asm volatile("" :: "v"(accs[j][0]), "v"(accs[j][1]), "v"(accs[j][2]), "v"(accs[j][3]));
        }
        if constexpr (KS) {
            // ---- pairwise split-K hand-off (see KS above): slot = [wave][MB x WNB fragments][64 lanes x 16 bytes], register order ------
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.ks_slots + static_cast<size_t>(tile) * (BM * BN * 4), 0, BM * BN * 4, 0x00020000);
            const int base = wave * (MB * WNB * 1024) + lane * 16;
            if (kz) {
#pragma unroll
                for (int j = 0; j < MB; ++j)
#pragma unroll
                    for (int i = 0; i < WNB; ++i) {
                        const u32x4 v = {static_cast<uint32_t>(acc[j][i][0]), static_cast<uint32_t>(acc[j][i][1]), static_cast<uint32_t>(acc[j][i][2]), static_cast<uint32_t>(acc[j][i][3])};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + (j * WNB + i) * 1024, 0, 16 /* sc1: write-through */);
                    }
                wr_wait_vmcnt<0>();                                               // every storing wave drains its own stores
                __builtin_amdgcn_s_barrier();                                    // (the epilogue's two barriers: the loader waves arrive here as well)
                if (tid == 0) __hip_atomic_store(a.ks_flags + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_barrier();
                return;
            }
            if (lane == 0) while (__hip_atomic_load(a.ks_flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                u32x4 v[WNB];
#pragma unroll
                for (int i = 0; i < WNB; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (j * WNB + i) * 1024, 0, 16 /* sc1 */);
#pragma unroll
                for (int i = 0; i < WNB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[j][i][r] += static_cast<int>(v[i][r]);
            }
        }
        stamp(2);
        }
    }

    // =================================================================================================================
    // epilogue
    // =================================================================================================================
    if (wave < CW) {
        int n_out = a.n_out;
        if (a.n_out_dev) n_out = n_out_dev_v < n_out ? n_out_dev_v : n_out;
        if (!a.xo || !a.wo) n_out = 0;
        const int kpad = (n_out + 15) & ~15;                                     // readable width of an outlier row
        const int ksteps = (n_out + 31) >> 5;
        constexpr float PRE = I4 ? (1.f / 256.f) : 1.f;
        const bool has_add = a.addend != nullptr, has_bias = a.bias != nullptr, do_silu = a.act != MIXQ_ACT_NONE;
        const bool mul_add = a.act == MIXQ_ACT_SILU_MUL;
        const bool has_amax = a.row_amax != nullptr;
        auto unpack4 = [](u32x2 v, float* o) MIXQ_INL {
            o[0] = h2f(static_cast<uint16_t>(v.x & 0xffffu)); o[1] = h2f(static_cast<uint16_t>(v.x >> 16));
            o[2] = h2f(static_cast<uint16_t>(v.y & 0xffffu)); o[3] = h2f(static_cast<uint16_t>(v.y >> 16));
        };

        if constexpr (EPI2) {
            // ---- EPI2: row panels, software-pipelined, copied out by the loader waves (see the note at its declaration) ----------------
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            constexpr int NB = PJ * WNB;                                         // 16 x 16 blocks of one panel (per wave)
            constexpr int TQ1 = TQ > 0 ? TQ : 1;
            // (the lane's coordinates again, opaque to the compiler: values derived from them at kernel entry - row indices of the scale
            // loads - would otherwise be kept alive across the k loop for the epilogue's addresses, in scratch memory where registers are short)
            int lm = lane & 15, lq = lane >> 4;
            asm volatile("" : "+v"(lm), "+v"(lq));
            float swv[WNB][4], sxv[MB];
            // the weight fragments (W_out rows of this wave, 16 x 32 each) of the first TQ tail k-steps - the ones whose X_out blocks went
            // through LDS - requested before the barrier; k-steps that do not exist and chunks past the padded width are zeros
            u32x4 wo2[TQ1][WNB];
            auto wo_load = [&](int kk, u32x4* dst) MIXQ_INL {
                const int kb = kk * 32 + lq * 8;
                const bool in = kk < ksteps && kb < kpad;
#pragma unroll
                for (int i = 0; i < WNB; ++i) {
                    int wr = nw0 + i * 16 + lm; wr = wr < a.N ? wr : a.N - 1;
                    dst[i] = in ? *reinterpret_cast<const u32x4*>(a.wo + static_cast<size_t>(wr) * a.ldwo + kb) : u32x4{0, 0, 0, 0};
                }
            };
            // columns >= n_out of a k-step (the pad may hold anything, NaN patterns included)
            auto keep_of = [&](int kk) MIXQ_INL {
                const int kb = kk * 32 + lq * 8;
                u32x4 k4;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t k = 0;
                    if (kb + 2 * d < n_out)     k |= 0x0000ffffu;
                    if (kb + 2 * d + 1 < n_out) k |= 0xffff0000u;
                    k4[d] = k;
                }
                return k4;
            };
            // Does tail k-step kk (32 outlier columns) exist?  A k-step past the layer's columns is SKIPPED (its MFMAs are what the epilogue
            // waits for: 0.88 of its 2.5 us in profiles/r04_epilogue_ablations.txt; at 20 columns the skip is worth 0.4 us of the launch).
            // k-step 0 always runs: the dequantisation of the next panel is threaded between ITS MFMAs (absent columns multiply zeros).
            // (A K = 16 MFMA for a k-step holding <= 16 columns - the metric's 41 = 32 + 9 - was built and measured: v_mfma_f32_16x16x16f16
            // issues no faster than the K = 32 form on gfx950, the launch was 0.3 us slower; removed.)
            auto tail_mode = [&](int kk) MIXQ_INL { return (kk == 0 || n_out > 32 * kk) ? 2 : 0; };
            __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): see the first form
#pragma unroll
            for (int kk = 0; kk < TQ; ++kk) wo_load(kk, wo2[kk]);
            copy_setup(tid, std::integral_constant<int, NT>{}, voffA, loffA);     // (this thread's share of the last panel's copy-out)
            __builtin_amdgcn_s_barrier();                                        // every wave is done reading the ring; X_out blocks landed and fixed up
            stamp(6);
            {   // the scales, from the dword slots the loader's LDS-DMA filled (landed: its vmcnt(0) in front of the barrier)
                u32x4 swr[WNB];
                uint32_t sxr[MB];
#pragma unroll
                for (int i = 0; i < WNB; ++i) swr[i] = *reinterpret_cast<const u32x4*>(lds + SW_OFF + (wave * WN + i * 16 + lq * 4) * 4);
#pragma unroll
                for (int j = 0; j < MB; ++j) sxr[j] = *reinterpret_cast<const uint32_t*>(lds + SX_OFF + (j * 16 + lm) * 4);
#pragma unroll
                for (int i = 0; i < WNB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) swv[i][r] = h2f(static_cast<uint16_t>(swr[i][r]));
#pragma unroll
                for (int j = 0; j < MB; ++j) sxv[j] = h2f(static_cast<uint16_t>(sxr[j])) * PRE;
            }
            // columns >= n_out in the WEIGHT fragments (the pad of weight_cache may hold anything); the X_out blocks were cleaned in LDS by the loaders
#pragma unroll
            for (int kk = 0; kk < TQ; ++kk) {
                if (kk * 32 + 32 > n_out) {
                    const u32x4 kp = keep_of(kk);
#pragma unroll
                    for (int i = 0; i < WNB; ++i) wo2[kk][i] &= kp;
                }
            }
            uint32_t rmax[MB];                                                   // running max of |fp16 bits| per tile row (two halves packed)
#pragma unroll
            for (int j = 0; j < MB; ++j) rmax[j] = 0u;
            f32x4 fa[MB][WNB];
            auto deq = [&](int pn, int b) MIXQ_INL {                // block b of panel pn: acc * sx * sw (this order)
                const int j = pn * PJ + b / WNB, i = b % WNB;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (ABLK == 66) fa[j][i][r] = static_cast<float>(acc[j][i][r]);   // probe: no scaling
                    else fa[j][i][r] = static_cast<float>(acc[j][i][r]) * sxv[j] * swv[i][r];
                }
            };
            // optional terms, fp16, staged or stored: the 16-channel block column i of panel pn (PJ blocks) - the first form's finish_tile, cut
            // into pieces small enough to sit between two MFMAs
            auto finish_col = [&](auto opt_c, auto st_c, int pn, int i) MIXQ_INL {
                constexpr bool OPT = decltype(opt_c)::value, ST = decltype(st_c)::value;   // ST: staged through LDS (else: 8-byte stores straight to Y)
                const int nloc = wave * WN + i * 16 + lq * 4, n = n0 + nloc;
                const int nc = n < a.N ? n : a.N - 4;
                float bv[4];
                if (OPT && has_bias) {
                    const u32x4 br = *reinterpret_cast<const u32x4*>(lds + BI_OFF + nloc * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = h2f(static_cast<uint16_t>(br[r]));
                }
                uint32_t keep_lo = 0x7fff7fffu, keep_hi = 0x7fff7fffu;            // |.| of the 4 halves; the next layer's outlier columns drop out
                if (OPT && has_amax) {
                    // (columns past N are computed from zero scales - values that exist nowhere in y - and count for nothing)
                    const uint32_t mb = n >= a.N ? 0xfu : (a.amax_mask ? (a.amax_mask[nc >> 5] >> (nc & 31)) & 0xfu : 0u);
                    if (mb & 1u) keep_lo &= 0xffff0000u;
                    if (mb & 2u) keep_lo &= 0x0000ffffu;
                    if (mb & 4u) keep_hi &= 0xffff0000u;
                    if (mb & 8u) keep_hi &= 0x0000ffffu;
                }
#pragma unroll
                for (int jj = 0; jj < PJ; ++jj) {
                    const int j = pn * PJ + jj;
                    const int m = m0 + j * 16 + lm;
                    f32x4 f = fa[j][i];
                    float av[4];
                    if (OPT && has_add) {
                        unpack4(*reinterpret_cast<const u32x2_u*>(a.addend + static_cast<size_t>(m < a.M ? m : a.M - 1) * a.lda + nc), av);
                        if (!mul_add) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[r] += av[r];
                        }
                    }
                    if (OPT && do_silu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] = wr_silu(f[r]);
                    }
                    if (OPT && has_bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] += bv[r];
                    }
                    if (OPT && mul_add) {                              // (silu(z) + bias) * up: linear.py:372-373, then mlp.py:61
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] *= av[r];
                    }
                    u32x2 o;                                           // round-to-nearest-even, two values per v_cvt_pk_f16_f32
                    o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{f[0], f[1]}, f16x2));
                    o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{f[2], f[3]}, f16x2));
                    if (OPT && has_amax) {
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        const us2 mx = __builtin_elementwise_max(__builtin_bit_cast(us2, o.x & keep_lo), __builtin_bit_cast(us2, o.y & keep_hi));
                        rmax[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, rmax[j]), mx));
                    }
                    if constexpr (ABLK == 65) { asm volatile("" :: "v"(o)); }      // probe: converted, never staged
                    else if constexpr (ST) *reinterpret_cast<u32x2*>(lds + (j * 16 + lm) * OPITCH + nloc * 2) = o;
                    else if (m < a.M && n < a.N) *reinterpret_cast<u32x2_u*>(a.y + static_cast<size_t>(m) * a.ldy + n) = o;
                }
            };
            // PAIR: the same piece for the joint gate / up launch - physical columns n .. n+3 of a lane are up[c], up[c+1], gate[c], gate[c+1]
            // of output channels c = n / 2, c + 1.  up goes through the fp16 rounding it has as up_proj's output tensor (z + bias -> fp16,
            // linear.py:285), the product is taken in fp32 and rounded once: what MIXQ_ACT_SILU_MUL computes from that tensor.
            auto finish_pair = [&](auto st_c, int pn, int i) MIXQ_INL {
                constexpr bool ST = decltype(st_c)::value;
                const int nloc = wave * WN + i * 16 + lq * 4, n = n0 + nloc, oc = n >> 1;
                float bv[4];
                if (has_bias) {
                    const u32x4 br = *reinterpret_cast<const u32x4*>(lds + BI_OFF + nloc * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = h2f(static_cast<uint16_t>(br[r]));
                }
                uint32_t keep = 0x7fff7fffu;
                if (has_amax) {
                    const uint32_t mb = n >= a.N ? 0x3u : (a.amax_mask ? (a.amax_mask[oc >> 5] >> (oc & 31)) & 0x3u : 0u);
                    if (mb & 1u) keep &= 0xffff0000u;
                    if (mb & 2u) keep &= 0x0000ffffu;
                }
#pragma unroll
                for (int jj = 0; jj < PJ; ++jj) {
                    const int j = pn * PJ + jj;
                    const int m = m0 + j * 16 + lm;
                    f32x4 f = fa[j][i];
                    if (has_bias) { f[0] += bv[0]; f[1] += bv[1]; }
                    const f16x2 uh = __builtin_convertvector(f32x2{f[0], f[1]}, f16x2);
                    float g0 = wr_silu(f[2]), g1 = wr_silu(f[3]);
                    if (has_bias) { g0 += bv[2]; g1 += bv[3]; }
                    g0 *= static_cast<float>(uh[0]); g1 *= static_cast<float>(uh[1]);
                    const uint32_t o = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{g0, g1}, f16x2));
                    if (has_amax) {
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        rmax[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, rmax[j]), __builtin_bit_cast(us2, o & keep)));
                    }
                    if constexpr (ST) *reinterpret_cast<uint32_t*>(lds + (j * 16 + lm) * OPITCH + nloc) = o;   // (byte offset of output column nloc / 2)
                    else if (m < a.M && n < a.N) *reinterpret_cast<uint32_t*>(a.y + static_cast<size_t>(m) * a.ldy + oc) = o;
                }
            };
            auto raise_flag = [&](int pn) MIXQ_INL {
                // panel staged by this wave: a flag word written BEHIND the panel's stores (asm with a memory clobber: the compiler may not move
                // it ahead of them; the LDS executes one wave's operations in order)
                const uint32_t one = 1u, fa_ = FLAG_OFF + (pn * CW + wave) * 4;
                asm volatile("ds_write_b32 %0, %1" :: "v"(fa_), "v"(one) : "memory");
            };
            const bool opt = has_add || has_bias || do_silu || has_amax;
            u32x4 xo2[TQ1][PJ];                                                  // X_out fragments of the current panel (from the LDS blocks)
            auto xo_read = [&](int pn, int kk) MIXQ_INL {
                if (tail_mode(kk) == 0) return;
#pragma unroll
                for (int jj = 0; jj < PJ; ++jj)
                    xo2[kk][jj] = *reinterpret_cast<const u32x4*>(lds + TAILX + (kk * MB + pn * PJ + jj) * 1024 + lane * 16);
            };
            auto tail_mma = [&](int j, int i, u32x4 w, u32x4 x) MIXQ_INL {
                if constexpr (ABLK == 64) return;                                 // probe: the epilogue without its tail MFMAs (results are garbage)
                fa[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), fa[j][i], 0, 0, 0);
            };
            // more than 32 TQ outlier columns (rare: the search stops adding beyond 128, linear.py:224): the remaining k-steps with both
            // operands from global memory, one exposed round trip each
            auto extra_ksteps = [&](int pn) MIXQ_INL {
                for (int kk = TQ; kk < ksteps; ++kk) {
                    u32x4 wt[WNB], xt[PJ];
                    wo_load(kk, wt);
                    const int kb = kk * 32 + lq * 8;
#pragma unroll
                    for (int jj = 0; jj < PJ; ++jj) {
                        int xr = m0 + (pn * PJ + jj) * 16 + lm; xr = xr < a.M ? xr : a.M - 1;
                        xt[jj] = kb < kpad ? *reinterpret_cast<const u32x4*>(a.xo + static_cast<size_t>(xr) * a.ldxo + kb) : u32x4{0, 0, 0, 0};
                    }
                    const u32x4 kp = keep_of(kk);
#pragma unroll
                    for (int i = 0; i < WNB; ++i) wt[i] &= kp;
#pragma unroll
                    for (int jj = 0; jj < PJ; ++jj) xt[jj] &= kp;
#pragma unroll
                    for (int b = 0; b < NB; ++b) tail_mma(pn * PJ + b / WNB, b % WNB, wt[b % WNB], xt[b / WNB]);
                }
            };
#pragma unroll
            for (int kk = 0; kk < TQ; ++kk) xo_read(0, kk);
#pragma unroll
            for (int b = 0; b < NB; ++b) deq(0, b);
            // The pipeline (one body for every outlier count, as the first form: k-steps that do not exist multiply zeros - under VALU work
            // that is there anyway).  Iteration pn runs panel pn's tail MFMAs in order (kk-major: NB MFMAs lie between two on one accumulator)
            // and, BETWEEN them, dequantises block d of panel pn + 1 behind every TQ-th MFMA and converts / stages block column i of panel
            // pn - 1 behind some of the others: one wave per SIMD hides ~4 plain VALU instructions behind a 16-cycle MFMA, and a ds_write_b64
            // issued every ~80 cycles costs its issue slot where six in a row cost 24 cycles each (the CU's LDS takes one per 6 cycles from
            // its four waves: tools/ubench_valu.hip).  (The empty asm statements pin a piece HERE: left alone, the compiler sinks the
            // multiplications down to their first use, where nothing overlaps them.)  With optional terms (SiLU, bias, addend, row maxima)
            // the staging stays behind the panel's MFMAs: its global loads and transcendentals do not fit between two MFMAs.
            // (staging one panel BEHIND, its conversions and LDS writes between the next panel's MFMAs, was built and measured in round 4:
            // 24.71 vs 24.50 us, the flags go up a panel later and the loaders start later; the MFMAs of the tail are what the phase waits for,
            // 0.88 us of the launch in the no-tail ablation, not the LDS writes: 0.06 us.  profiles/r04_epilogue_ablations.txt.  Removed.)
            wr_static_for<0, NPAN + 1>([&](auto p_c) MIXQ_INL {
                constexpr int pn = decltype(p_c)::value;                         // 0 .. NPAN: MFMAs of panel pn, dequantisation of pn + 1, staging of pn - 1 (lag) or pn
                if constexpr (pn < NPAN && TQ > 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < TQ; ++kk) {
                        if (tail_mode(kk) == 0) continue;                        // (wave-uniform; k-step 0 always runs)
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            tail_mma(pn * PJ + b / WNB, b % WNB, wo2[kk][b % WNB], xo2[kk][b / WNB]);
                            __builtin_amdgcn_sched_barrier(0);
                            if (pn + 1 < NPAN && kk == 0) {                      // the next panel's blocks, one behind each MFMA of k-step 0
                                const int d = b, j2 = (pn + 1) * PJ + d / WNB, i2 = d % WNB;
                                deq(pn + 1, d);
                                asm volatile("" : "+v"(fa[j2][i2]));
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (pn + 1 < NPAN) xo_read(pn + 1, kk);         // (this k-step's fragments are consumed: the next panel's, in flight under what follows)
                    }
                    extra_ksteps(pn);
                } else if constexpr (pn < NPAN) {
                    if constexpr (pn + 1 < NPAN) {
#pragma unroll
                        for (int b = 0; b < NB; ++b) deq(pn + 1, b);
                    }
                }
                {
                    if constexpr (pn < NPAN) {
                        // (straight-line code per case: the optional terms and the unstaged stores are compile-time switches of finish_col)
                        if constexpr (PAIR) {
                            if (staged) {
#pragma unroll
                                for (int c = 0; c < WNB; ++c) finish_pair(std::true_type{}, pn, c);
                            } else {
#pragma unroll
                                for (int c = 0; c < WNB; ++c) finish_pair(std::false_type{}, pn, c);
                            }
                        } else if (staged && !opt) {
#pragma unroll
                            for (int c = 0; c < WNB; ++c) finish_col(std::false_type{}, std::true_type{}, pn, c);
                        } else if (staged) {
#pragma unroll
                            for (int c = 0; c < WNB; ++c) finish_col(std::true_type{}, std::true_type{}, pn, c);
                        } else {
#pragma unroll
                            for (int c = 0; c < WNB; ++c) finish_col(std::true_type{}, std::false_type{}, pn, c);
                        }
                        stamp2(2 * pn, 0);
                        if constexpr (pn < LP) raise_flag(pn);
                    }
                }
            });
            if (has_amax) {                                                      // one slot per (wave, lane group, row): no atomics, no initialisation
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    const uint32_t lo = rmax[j] & 0xffffu, hi = rmax[j] >> 16;
                    *reinterpret_cast<uint32_t*>(lds + AMAX_OFF + ((wave * 4 + lq) * BM + j * 16 + lm) * 4) = lo > hi ? lo : hi;
                }
            }
            stamp(7);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // ds_write is asynchronous; s_barrier does not wait for it
            __builtin_amdgcn_s_barrier();                                        // every panel staged: what the loaders did not copy is shared by all waves
            stamp(3);
        } else {
        // fp16 outlier tail operands: 16 x 32 fragments, lane = row lm, columns kk*32 + lq*8 .. +8.  TD register sets: with two,
        // the operands of tail k-step kk+1 are in flight behind the MFMAs of kk; the fattest tiles only have room for one.
        constexpr int TD = (SELF || MB * WNB * 4 + 2 * (MB + WNB) * 4 + 40 > 256) ? 1 : 2;
        u32x4 xoq[TD][MB], woq[TD][WNB];
        auto tail_load_w = [&](int P, int kk) MIXQ_INL {            // P: constant after unrolling
            const int kb = kk * 32 + lq * 8;
            const bool in = kb < kpad;                     // chunks past the padded width are not addressable: zeros
#pragma unroll
            for (int i = 0; i < WNB; ++i) {
                int wr = nw0 + i * 16 + lm; wr = wr < a.N ? wr : a.N - 1;
                woq[P][i] = in ? *reinterpret_cast<const u32x4*>(a.wo + static_cast<size_t>(wr) * a.ldwo + kb) : u32x4{0, 0, 0, 0};
            }
        };
        auto tail_load_x = [&](int P, int kk) MIXQ_INL {            // k-steps < TQ: from the blocks the loader staged in LDS (valid after barrier 1)
            const int kb = kk * 32 + lq * 8;
            const bool in = kb < kpad;
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                if (kk < TQ) {
                    xoq[P][j] = *reinterpret_cast<const u32x4*>(lds + TAILX + (kk * MB + j) * 1024 + lane * 16);
                } else {
                    int xr = m0 + j * 16 + lm; xr = xr < a.M ? xr : a.M - 1;
                    xoq[P][j] = in ? *reinterpret_cast<const u32x4*>(a.xo + static_cast<size_t>(xr) * a.ldxo + kb) : u32x4{0, 0, 0, 0};
                }
            }
        };
        auto tail_load = [&](int P, int kk) MIXQ_INL { tail_load_w(P, kk); tail_load_x(P, kk); };
        // both register sets are requested BEFORE the barrier and the dequantisation (the weight ring and the X fragments are dead:
        // their registers hold the tail operands), so the tail's first two k-steps - all of it up to 64 outlier columns - expose
        // no memory round trip of their own
        // (every load of this wave has long landed - the weight ring was consumed, the scales were the first requests of the kernel;
        // saying so through the builtin the compiler's counter model understands keeps it from draining the tail loads issued next
        // with a vmcnt(0) in front of the dequantisation, whose scale operands it would otherwise believe to be still in flight)
        __builtin_amdgcn_s_waitcnt(0x0F70);                                      // vmcnt(0)
        if (ksteps > 0) tail_load_w(0, 0);
        if (TD > 1 && ksteps > 1) tail_load_w(TD - 1, 1);
        __builtin_amdgcn_s_barrier();                                            // every wave is done reading the ring; X_out blocks landed
        if constexpr (KS) { if (tid == 0) __hip_atomic_store(a.ks_flags + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (every wave has passed its poll)
        stamp(6);
        if (ksteps > 0) tail_load_x(0, 0);
        if (TD > 1 && ksteps > 1) tail_load_x(TD - 1, 1);

        // Dequantisation and the first tail k-steps as ONE block-wise pipeline: block b+1 is dequantised (12 VALU) between the
        // tail MFMAs of block b, so the fp16 MFMAs of the first two tail k-steps (all of the tail up to 64 outlier columns) run under
        // the dequantisation's VALU work instead of behind it (round 2 ran 288 VALU, then 48 MFMAs: 1.05 us for 41 columns).
        // Per element the order of operations is unchanged: acc * sx * sw, then k-step 0, 1, 2, ... - results are bit-identical.
        f32x4 fa[MB][WNB];
        float swv[WNB][4], sxv[MB];
#pragma unroll
        for (int i = 0; i < WNB; ++i) unpack4(swp[i], swv[i]);
#pragma unroll
        for (int j = 0; j < MB; ++j) sxv[j] = h2f(sxh[j]) * PRE;
        auto dequant_block = [&](int j, int i) MIXQ_INL {
#pragma unroll
            for (int r = 0; r < 4; ++r) fa[j][i][r] = static_cast<float>(acc[j][i][r]) * sxv[j] * swv[i][r];
        };
        auto tail_mask = [&](int P, int kk) MIXQ_INL {                                    // columns >= n_out of a k-step (the pad may hold anything)
            const int kb = kk * 32 + lq * 8;
            if (kb + 8 > n_out) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t keep = 0;
                    if (kb + 2 * d < n_out)     keep |= 0x0000ffffu;
                    if (kb + 2 * d + 1 < n_out) keep |= 0xffff0000u;
#pragma unroll
                    for (int j = 0; j < MB; ++j) xoq[P][j][d] &= keep;
#pragma unroll
                    for (int i = 0; i < WNB; ++i) woq[P][i][d] &= keep;
                }
            }
        };
        auto tail_mfma1 = [&](int P, int j, int i) MIXQ_INL {
            fa[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, woq[P][i]), __builtin_bit_cast(f16x8, xoq[P][j]),
                                                              fa[j][i], 0, 0, 0);
        };
        // PANELS: the fat prefill tile (SELF) holds 256 accumulator registers per lane; dequantised all at once they would need a
        // second 256 - in VGPRs, since the VALU cannot write AGPRs - next to the tail operands (165 spilled registers, and wrong
        // results under load).  So its epilogue runs one 16-channel column of blocks (PW = 1 of the WNB) at a time: read from the
        // accumulator file, dequantise, tail, optional terms, fp16, staged - then the next.  Every other tiling is ONE panel.
        constexpr int PW = SELF ? 1 : WNB, NP = WNB / PW;
        auto tail_mma = [&](int P, int kk, auto p_c) MIXQ_INL {
            constexpr int p = decltype(p_c)::value;
            tail_mask(P, kk);                                                    // (idempotent: a set shared by several panels is masked again)
#pragma unroll
            for (int j = 0; j < MB; ++j)
#pragma unroll
                for (int i = p * PW; i < (p + 1) * PW; ++i) tail_mfma1(P, j, i);
        };
        // K0 / K1: tail k-step 0 from register set 0 / k-step 1 from set 1 ride along (compile-time: straight-line code)
        auto dequant_with_tail = [&](auto k0_c, auto k1_c, auto p_c) MIXQ_INL {
            constexpr bool K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value;
            constexpr int p = decltype(p_c)::value, NB = MB * PW;
            dequant_block(0, p * PW);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int j = b / PW, i = p * PW + b % PW;
                if (b + 1 < NB) dequant_block((b + 1) / PW, p * PW + (b + 1) % PW);
                if constexpr (K0) tail_mfma1(0, j, i);
                if constexpr (K1) tail_mfma1(TD - 1, j, i);
                // (block order is left to the scheduler)
            }
        };
        uint32_t rmax[MB];                                                       // running max of |fp16 bits| per tile row (two halves packed)
#pragma unroll
        for (int j = 0; j < MB; ++j) rmax[j] = 0u;
        auto finish_tile = [&](auto staged_c, auto opt_c, auto p_c) MIXQ_INL {
            constexpr bool ST = decltype(staged_c)::value, OPT = decltype(opt_c)::value;
            constexpr int p = decltype(p_c)::value;
#pragma unroll
            for (int i = p * PW; i < (p + 1) * PW; ++i) {
                const int nloc = wave * WN + i * 16 + lq * 4, n = n0 + nloc;
                const int nc = n < a.N ? n : a.N - 4;
                float bv[4];
                if (OPT && has_bias) {
                    if constexpr (PREBIAS) unpack4(bvp[i], bv);
                    else unpack4(*reinterpret_cast<const u32x2_u*>(a.bias + nc), bv);
                }
                uint32_t keep_lo = 0x7fff7fffu, keep_hi = 0x7fff7fffu;            // |.| of the 4 halves; the next layer's outlier columns drop out
                if (OPT && has_amax) {
                    // (columns past N are computed from clamped operands - values that exist nowhere in y - and count for nothing)
                    const uint32_t mb = n >= a.N ? 0xfu : (a.amax_mask ? (a.amax_mask[nc >> 5] >> (nc & 31)) & 0xfu : 0u);
                    if (mb & 1u) keep_lo &= 0xffff0000u;
                    if (mb & 2u) keep_lo &= 0x0000ffffu;
                    if (mb & 4u) keep_hi &= 0xffff0000u;
                    if (mb & 8u) keep_hi &= 0x0000ffffu;
                }
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    const int mloc = j * 16 + lm, m = m0 + mloc;
                    f32x4 f = fa[j][i];
                    float av[4];
                    if (OPT && has_add) {
                        unpack4(*reinterpret_cast<const u32x2_u*>(a.addend + static_cast<size_t>(m < a.M ? m : a.M - 1) * a.lda + nc), av);
                        if (!mul_add) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[r] += av[r];
                        }
                    }
                    if (OPT && do_silu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] = wr_silu(f[r]);
                    }
                    if (OPT && has_bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] += bv[r];
                    }
                    if (OPT && mul_add) {                          // (silu(z) + bias) * up: linear.py:372-373, then mlp.py:61
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] *= av[r];
                    }
                    u32x2 o;
                    o.x = static_cast<uint32_t>(f2h(f[0])) | (static_cast<uint32_t>(f2h(f[1])) << 16);
                    o.y = static_cast<uint32_t>(f2h(f[2])) | (static_cast<uint32_t>(f2h(f[3])) << 16);
                    if (OPT && has_amax) {
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        const us2 m = __builtin_elementwise_max(__builtin_bit_cast(us2, o.x & keep_lo), __builtin_bit_cast(us2, o.y & keep_hi));
                        rmax[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, rmax[j]), m));
                    }
                    if constexpr (ST) {
                        *reinterpret_cast<u32x2*>(lds + mloc * OPITCH + nloc * 2) = o;
                    } else {
                        if (m < a.M && n < a.N) *reinterpret_cast<u32x2_u*>(a.y + static_cast<size_t>(m) * a.ldy + n) = o;
                    }
                }
            }
        };
        const bool opt = has_add || has_bias || do_silu || has_amax;
        wr_static_for<0, NP>([&](auto p_c) MIXQ_INL {
            constexpr int p = decltype(p_c)::value;
            if constexpr (TD == 1) {
                // one register set: tail k-step 0 (requested before the barrier) rides along with the dequantisation, the rest follow
                tail_mask(0, 0);
                dequant_with_tail(std::true_type{}, std::false_type{}, p_c);
                for (int kk = 1; kk < ksteps; ++kk) { tail_load(0, kk); tail_mma(0, kk, p_c); }
                if constexpr (p + 1 < NP) { if (ksteps > 1) tail_load(0, 0); }     // the next panel starts from k-step 0 again
            } else {
                // ONE straight-line form for every outlier count: a register set whose k-step does not exist was never loaded (it
                // holds old weight-ring bytes) and tail_mask clears all of it (kb >= n_out), so its MFMAs add zeros - under VALU work
                // that is there anyway.  (Branching between "with" and "without" forms of this phase cost 92-300 spilled registers.)
                tail_mask(0, 0);
                tail_mask(1, 1);
                dequant_with_tail(std::true_type{}, std::true_type{}, p_c);
                if (ksteps > 2) {                                                // > 64 outlier columns: the two sets again, one pair ahead
                    tail_load(0, 2);
                    if (ksteps > 3) tail_load(TD - 1, 3);
                    for (int kk0 = 2; kk0 < ksteps; kk0 += 2) {
                        tail_mma(0, kk0, p_c);
                        if (kk0 + 2 < ksteps) tail_load(0, kk0 + 2);
                        if (kk0 + 1 < ksteps) {
                            tail_mma(TD - 1, kk0 + 1, p_c);
                            if (kk0 + 3 < ksteps) tail_load(TD - 1, kk0 + 3);
                        }
                    }
                    if constexpr (p + 1 < NP) { tail_load(0, 0); tail_load(TD - 1, 1); }   // the next panel starts from k-steps 0 and 1 again
                }
            }
            if (staged) { if (opt) finish_tile(std::true_type{}, std::true_type{}, p_c); else finish_tile(std::true_type{}, std::false_type{}, p_c); }
            else        { if (opt) finish_tile(std::false_type{}, std::true_type{}, p_c); else finish_tile(std::false_type{}, std::false_type{}, p_c); }
        });
        if (has_amax) {                                                          // one slot per (wave, lane group, row): no atomics, no initialisation
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const uint32_t lo = rmax[j] & 0xffffu, hi = rmax[j] >> 16;
                *reinterpret_cast<uint32_t*>(lds + AMAX_OFF + ((wave * 4 + lq) * BM + j * 16 + lm) * 4) = lo > hi ? lo : hi;
            }
        }
        stamp(7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // ds_write is asynchronous; s_barrier does not wait for it
        __builtin_amdgcn_s_barrier();                                            // staging tile complete
        stamp(3);
        }
    }
    if (a.row_amax != nullptr && tid < BM) {                                     // (after the barrier above: every wave's slots are written)
        uint32_t v = 0u;
#pragma unroll
        for (int sl = 0; sl < CW * 4; ++sl) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(lds + AMAX_OFF + (sl * BM + tid) * 4);
            v = w > v ? w : v;
        }
        if (m0 + tid < a.M) atomicMax(a.row_amax + m0 + tid, v);
    }
    if constexpr (EPI2) {
        if (staged) {                                                 // (ABLK 61, a probe: LP = 0 - everything here, as the first form)
#pragma unroll
            for (int p = LP; p < NPAN; ++p) copy_panel(p, std::integral_constant<int, NT>{}, voffA, loffA);   // (the earlier panels left under the arithmetic)
        }
    } else if (staged) {
        // all waves (loader included): 16 bytes per lane, 16 / 24 / 32 consecutive lanes cover one row segment of the tile.  Every
        // LDS read of a lane is issued before its first store, so the copy-out costs ONE LDS round trip, not one per 16 bytes.
        constexpr int ITER = (BM * CPR + NT - 1) / NT;
        u32x4 v[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = tid + it * NT;
            const int r = q / CPR, c = q - r * CPR;
            if (q < BM * CPR) v[it] = *reinterpret_cast<const u32x4*>(lds + r * OPITCH + c * 16);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = tid + it * NT;
            const int r = q / CPR, c = q - r * CPR;
            const int m = m0 + r, n = n0 + c * 8;
            // streaming (nt) store: Y is written once and not re-read by this kernel
            if (q < BM * CPR && m < a.M && n < a.N && (ABLK != 7 || a.act == 12345)) {
                if constexpr (ABLK == 8) *reinterpret_cast<u32x4*>(a.y + static_cast<size_t>(m) * a.ldy + n) = v[it];
                else __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(a.y + static_cast<size_t>(m) * a.ldy + n));
            }
        }
    }
#ifdef MIXQ_TUNING
    if (a.trace) {
        stamp(4);
        wr_wait_vmcnt<0>();
        stamp(5);
    }
#endif
}

// ---- configuration table -------------------------------------------------------------------------------------------
struct WrConfig {
    const char* name;
    int mb, wnb, nstage, loaders;
    void (*k8)(const WrArgs);
    void (*k4)(const WrArgs);
    void (*k6)(const WrArgs);                          // int4 as FP6 codes (MIXQ_FMT_F6X128 operands); nullptr: tiling not built in that form
    int nstage6;                                      // its X ring depth (k-steps of 128 elements, 1.5 KiB blocks)
    void (*k8p)(const WrArgs);                         // int8 with the paired gate / up epilogue (MIXQ_ACT_SILU_PAIR); nullptr: not built for this tiling
    void (*k6p)(const WrArgs);                         // ... the FP6 form with it
};
// (the int4 form expands nibbles in registers: 2 (MB + WNB) more fragment registers, so its weight ring is at most 4 deep)
#define MIXQ_WR(MBv, WNBv, NS, Dv, LD, ABL, TAG)                                                                        \
    { "wr" TAG, MBv, WNBv, NS, LD, gemm_wreg_kernel<MBv, WNBv, NS, Dv, 0, LD, ABL>,                                     \
      gemm_wreg_kernel<MBv, WNBv, NS, ((Dv) > 4 ? 4 : (Dv)) - ((MBv) * (WNBv) >= 32 ? 1 : 0), 1, LD, ABL>, nullptr, 0 }
// ... int8 only: the 128 x 256 tile's nibble form needs 128 accumulators + the ring + the expanded fragments - it never fitted the
// registers (112 bytes of scratch in round 3, some of it inside the hand-counted region: one allocation change away from spilling a ring
// register that is still in flight, which is what round 4's epilogue rewrite then triggered - wrong tiles, caught by the GPU suite).
// 4-bit layers use the FP6 form (tilings below) or, as nibbles, the LDS-staged kernel of gemm.hip; a forced 128 x 256 configuration
// answers MIXQ_EINVAL for nibble-packed operands.
#define MIXQ_WR8(MBv, WNBv, NS, Dv, LD, TAG)                                                                            \
    { "wr" TAG, MBv, WNBv, NS, LD, gemm_wreg_kernel<MBv, WNBv, NS, Dv, 0, LD, 0>, nullptr, nullptr, 0 }
// ... and with the FP6 form (Q = 3: the weight ring held as operand tuples; 20.9 vs 22.1 us at the metric shape against Q = 2, which
// assembles the tuples in front of each k-step's first MFMA and stays as a tuning config): X ring of NS6 stages (12 KiB each at 128 rows), weight ring of D6 k-steps: 2 for the 128-row tiles (a k-step
// is 128 elements, twice the time of an int8 one; 3 deep is 2.7 % slower - 22.03 vs 22.68 us at the metric shape), 3 for the 64-row
// tiles, whose k-steps are half as long (64 x 128 at 11008 -> 4096: 24.7 us with 3 or 4, 28.9 with 2; at 4096 -> 4096 12.6 / 12.9 / 13.5 us with 2 / 3 / 4): 3.  Tried and dropped: assembling the
// tuples of k-step kt+1 DURING k-step kt into a second tuple set (two ring slots + two sets, the 9 moves spread behind MFMAs instead of
// sitting in front of the first one): 24.5 vs 22.5 us, and wrong results on the 64-row tiles.
#define MIXQ_WR6(MBv, WNBv, NS, Dv, LD, NS6, D6, TAG)                                                                   \
    { "wr" TAG, MBv, WNBv, NS, LD, gemm_wreg_kernel<MBv, WNBv, NS, Dv, 0, LD, 0>,                                       \
      gemm_wreg_kernel<MBv, WNBv, NS, ((Dv) > 4 ? 4 : (Dv)) - ((MBv) * (WNBv) >= 32 ? 1 : 0), 1, LD, 0>,                \
      gemm_wreg_kernel<MBv, WNBv, NS6, D6, 3, LD, 0>, NS6 }

// ... and with the paired gate / up epilogue (ABL = 90) of the int8 form: the tilings the joint launch of an MLP block picks from
#define MIXQ_WR6P(MBv, WNBv, NS, Dv, LD, NS6, D6, TAG)                                                                  \
    { "wr" TAG, MBv, WNBv, NS, LD, gemm_wreg_kernel<MBv, WNBv, NS, Dv, 0, LD, 0>,                                       \
      gemm_wreg_kernel<MBv, WNBv, NS, ((Dv) > 4 ? 4 : (Dv)) - ((MBv) * (WNBv) >= 32 ? 1 : 0), 1, LD, 0>,                \
      gemm_wreg_kernel<MBv, WNBv, NS6, D6, 3, LD, 0>, NS6, gemm_wreg_kernel<MBv, WNBv, NS, Dv, 0, LD, 90>,              \
      gemm_wreg_kernel<MBv, WNBv, NS6, D6, 3, LD, 90> }
#define MIXQ_WR8P(MBv, WNBv, NS, Dv, LD, TAG)                                                                           \
    { "wr" TAG, MBv, WNBv, NS, LD, gemm_wreg_kernel<MBv, WNBv, NS, Dv, 0, LD, 0>, nullptr, nullptr, 0,                  \
      gemm_wreg_kernel<MBv, WNBv, NS, Dv, 0, LD, 90>, nullptr }

const WrConfig g_wr[] = {
    // name = tile (activation rows x weight rows) _ X ring depth _ weight ring depth _ loader waves
    MIXQ_WR6P(8, 3, 16, 4, 2, 8, 2, "128x192_s16_d4_l2"), // 0: the metric shape's tile: 232 tiles at 512 x 11008
    MIXQ_WR(8, 3, 16, 3, 2, 0, "128x192_s16_d3_l2"),   // 1
    MIXQ_WR(8, 3, 8, 4, 1, 0, "128x192_s8_d4_l1"),     // 2: the first form of this kernel (8-deep X ring, one loader)
    MIXQ_WR(8, 3, 12, 4, 2, 0, "128x192_s12_d4_l2"),   // 3
    MIXQ_WR6(8, 2, 16, 4, 2, 8, 2, "128x128_s16_d4_l2"),  // 4
    MIXQ_WR8P(8, 4, 16, 3, 2, "128x256_s16_d3_l2"),    // 5 (int8 only)
    MIXQ_WR6(4, 2, 16, 4, 2, 12, 3, "64x128_s16_d4_l2"),  // 6: N = 4096 at M = 512 is exactly 256 such tiles
    MIXQ_WR6P(4, 3, 16, 4, 2, 12, 3, "64x192_s16_d4_l2"), // 7: N = 6144
    MIXQ_WR6P(4, 4, 16, 4, 2, 12, 3, "64x256_s16_d4_l2"), // 8
    MIXQ_WR(8, 1, 8, 4, 1, 0, "128x64_s8_d4_l1"),      // 9
    MIXQ_WR(4, 1, 8, 4, 1, 0, "64x64_s8_d4_l1"),       // 10
    MIXQ_WR(8, 2, 8, 4, 1, 0, "128x128_s8_d4_l1"),     // 11
    MIXQ_WR8(8, 4, 8, 3, 1, "128x256_s8_d3_l1"),       // 12 (int8 only)
    MIXQ_WR(4, 2, 8, 4, 1, 0, "64x128_s8_d4_l1"),      // 13
    // small batches (M <= 32) of wide layers (N >= 8192): a weight stream, one 64-channel panel per workgroup.  12.4 us against
    // the 17.4 us of gemm_skinny.hip's in-workgroup K split at 32 x 4096 -> 11008 with cold weights; deeper weight rings (10, 14
    // k-steps) are slower (13.8, 14.0 us), narrow layers (N = 4096: 64 panels for 256 CUs) stay with gemm_skinny.hip
    // (profiles/r02_decode.txt)
    // (round 4: with an FP6 form too - 12 stages of 3 KiB, weight ring 4 k-steps of 128 elements (10 k-steps: no faster - the loop's barrier per k-step is what a 2-MFMA k-step waits for) - so a 4-bit layer serves its small
    // batches from the ONE FP6 image it keeps; tools/time_w4_small_batch.py)
    MIXQ_WR6P(2, 1, 8, 6, 1, 12, 4, "32x64_s8_d6_l1"), // 14 (WR_SMALL)
#ifdef MIXQ_TUNING                                     // only in the tools build (make tuning):
    // 15 (WR_KSPLIT): two workgroups per tile, half of K each (pairwise split-K).  Correct, tested, and slower than the data-parallel tilings at
    // every shape it was built for (32.6 vs 28.8 us at 11008 -> 4096, profiles/r03_splitk_ab.txt): round 5 moved it out of the product library
    MIXQ_WR(8, 2, 16, 4, 2, 50, "128x128_s16_d4_l2_k2"),
#ifndef MIXQ_TUNING_ONLY_77                            // (-DMIXQ_TUNING_ONLY_77: a quick build for iterating on probe 77 - the split-K entry, whose index is referenced, and that probe only)
    // ablation forms (results are garbage by design)
    { "wr128x192_f6_abl1_noW", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 2, 2, 1>, 8 },   // the FP6 form's feed ablations
    { "wr128x192_f6_abl2_noX", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 2, 2, 2>, 8 },
    { "wr128x192_f6_abl3_mfma", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 2, 2, 3>, 8 },
    { "wr128x192_f6_s10", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 10, 2, 2, 2, 0>, 10 },         // deeper X ring (Q = 2)
    { "wr128x192_f6mov_d2", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 2, 2, 0>, 8 },         // Q = 2: tuples assembled by v_mov in front of each k-step (the first FP6 form)
    { "wr64x128_f6mov_d3", 4, 2, 16, 2, nullptr, nullptr, gemm_wreg_kernel<4, 2, 12, 3, 2, 2, 0>, 12 },
    { "wr128x192_f6r_d3", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 3, 3, 2, 0>, 8 },           // Q = 3 ring variants
    { "wr128x192_f6r_s10", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 10, 2, 3, 2, 0>, 10 },
    { "wr128x192_f6_l4", 8, 3, 16, 4, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 2, 4, 0>, 8 },            // four loader waves
    // round 6: the tuple-ring (shipped) form's feed ablations with pseudo-random stand-ins, ring variants, and the in-k-step timeline
    { "wr128x192_f6r_abl1_noW", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 1>, 8 },
    { "wr128x192_f6r_abl2_noX", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 2>, 8 },
    { "wr128x192_f6r_abl3_mfma", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 3>, 8 },
    { "wr128x192_f6r_abl6_nobar", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 6>, 8 },
    { "wr128x192_f6r_abl32_dma_noreads", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 32>, 8 },
    { "wr128x192_f6r_abl33_reads_nodma", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 33>, 8 },
    { "wr128x192_f6r_t36_kstep_timeline", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 36>, 8 },
#define MIXQ_F6R_FL(NAME, BITS) { "wr128x192_f6r_" NAME, 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 100 + (BITS)>, 8 }
    MIXQ_F6R_FL("oldswz", 128), MIXQ_F6R_FL("pace", 2), MIXQ_F6R_FL("delay", 4), MIXQ_F6R_FL("xr8_pace", 1 + 2), MIXQ_F6R_FL("xr8_delay", 1 + 4), MIXQ_F6R_FL("lprio0", 8),
    MIXQ_F6R_FL("cprio3", 16), MIXQ_F6R_FL("xr8_lprio0", 1 + 8), MIXQ_F6R_FL("pace_lprio0", 2 + 8), MIXQ_F6R_FL("xr8_pace_lprio0", 1 + 2 + 8),
    MIXQ_F6R_FL("xw1", 256), MIXQ_F6R_FL("xw2", 512), MIXQ_F6R_FL("xw4", 768), MIXQ_F6R_FL("t_xw1", 32 + 256), MIXQ_F6R_FL("t_xw4", 32 + 768),
    MIXQ_F6R_FL("t_wave2", 32 + 64), MIXQ_F6R_FL("t_xr8", 32 + 1), MIXQ_F6R_FL("t_pace", 32 + 2), MIXQ_F6R_FL("t_xr8_pace", 32 + 1 + 2),
    MIXQ_WR(8, 3, 16, 4, 2, 36, "128x192_t36_kstep_timeline"), MIXQ_WR(8, 3, 16, 4, 2, 100 + 32 + 64, "128x192_t_wave2"),   // the int8 loop's in-k-step timeline
    { "wr128x192_f6r_xr8", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 38>, 8 },
    { "wr128x192_f6r_xr2", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 2, 39>, 8 },
    { "wr128x192_f6r_l4", 8, 3, 16, 4, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 2, 3, 4, 0>, 8 },
    { "wr128x192_f6r_d4", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 8, 4, 3, 2, 0>, 8 },
    { "wr128x192_f6r_s6", 8, 3, 16, 2, nullptr, nullptr, gemm_wreg_kernel<8, 3, 6, 2, 3, 2, 0>, 6 },
    MIXQ_WR(8, 3, 16, 4, 2, 61, "128x192_p61_epi2_copy_at_end"),
    MIXQ_WR(8, 3, 16, 4, 2, 62, "128x192_p62_epi2_loaders_prio0"),
    MIXQ_WR(8, 3, 16, 4, 2, 6, "128x192_abl6_no_kloop_barriers"),   // (timing probe: what the per-k-step barrier costs; results are garbage)
    MIXQ_WR(8, 3, 16, 4, 2, 64, "128x192_p64_epi2_no_tail_mfma"),
    MIXQ_WR(8, 3, 16, 4, 2, 65, "128x192_p65_epi2_no_staging_writes"),
    MIXQ_WR(8, 3, 16, 4, 2, 66, "128x192_p66_epi2_no_scaling"),
    { "wr128x192_p70_touch", 8, 3, 16, 2, gemm_wreg_kernel<8, 3, 16, 4, 0, 2, 70>, nullptr, nullptr, 0 },   // probe: the loaders pull the panel's weight lines into L2 ahead of the consumers
    MIXQ_WR(8, 3, 16, 4, 2, 67, "128x192_p67_epi2_loaders_copy_3_of_4"),
#endif
    MIXQ_WR(8, 3, 16, 4, 2, 77, "128x192_p77_one_launch"),
#ifndef MIXQ_TUNING_ONLY_77
    { "wr128x256_p78_mfma32x32x32", 8, 4, 16, 2, gemm_wreg_kernel<8, 4, 16, 3, 0, 2, 78>, nullptr, nullptr, 0 }, { "wr128x256_p79_shadow_acc", 8, 4, 16, 2, gemm_wreg_kernel<8, 4, 16, 3, 0, 2, 79>, nullptr, nullptr, 0 },
    MIXQ_WR(8, 3, 8, 4, 2, 0, "128x192_s8_d4_l2"), MIXQ_WR(8, 3, 8, 4, 2, 76, "128x192_s8_p76_weights_through_lds"),
    MIXQ_WR(8, 3, 16, 4, 2, 73, "128x192_p73_xwait_every4"), MIXQ_WR(8, 3, 16, 4, 2, 74, "128x192_p74_xwait_every2"), MIXQ_WR(8, 3, 16, 4, 2, 75, "128x192_p75_xwait_every1"),
    MIXQ_WR(8, 3, 16, 4, 2, 60, "128x192_p60_epi1"),   // cfg 0 with the first form of the epilogue (round 3's): the A/B partner of EPI2
    MIXQ_WR(8, 3, 16, 4, 2, 1, "128x192_abl1_noW"),    // cfg 0 without the weight loads
    MIXQ_WR(8, 3, 16, 4, 2, 2, "128x192_abl2_noX"),    // cfg 0 without X traffic
    MIXQ_WR(8, 3, 16, 4, 2, 3, "128x192_abl3_mfma"),   // cfg 0, MFMA + epilogue only
    MIXQ_WR(8, 3, 16, 4, 2, 16, "128x192_abl16_mfma_rnd"),    // abl3 / abl2 / abl1 with pseudo-random bytes in the never-loaded operands
    MIXQ_WR(8, 3, 16, 4, 2, 17, "128x192_abl17_noX_rnd"),
    MIXQ_WR(8, 3, 16, 4, 2, 18, "128x192_abl18_noW_rnd"),
    MIXQ_WR(8, 3, 16, 4, 2, 7, "128x192_abl7_nostore"),// cfg 0 without the stores of Y
    MIXQ_WR(8, 3, 16, 4, 2, 8, "128x192_abl8_plainst"),// cfg 0 with ordinary (not nt) stores of Y
    MIXQ_WR(8, 3, 16, 4, 2, 9, "128x192_abl9_empty"),  // returns at entry: the launch floor of this grid and LDS footprint
    MIXQ_WR(8, 3, 16, 4, 2, 12, "128x192_abl12_noramp"),// cfg 0 with all LOOK stages requested at once (the form before the ramp)
    MIXQ_WR(8, 3, 16, 4, 2, 20, "128x192_p20_wlate"),  // probes of WHEN requests are issued inside a k-step (results are correct)
    MIXQ_WR(8, 3, 16, 4, 2, 21, "128x192_p21_wearly"),
    MIXQ_WR(8, 3, 16, 4, 2, 22, "128x192_p22_paced"),
    MIXQ_WR(8, 3, 16, 4, 2, 23, "128x192_p23_wlate_paced"),
    MIXQ_WR(8, 3, 16, 4, 2, 24, "128x192_p24_paced2"),
    MIXQ_WR(8, 3, 16, 4, 2, 28, "128x192_p28_prio"),
    MIXQ_WR(8, 3, 16, 4, 2, 41, "128x192_p41_wrapload"),
    MIXQ_WR(8, 3, 16, 4, 2, 31, "128x192_p31_r2order"),   // round 2's order: re-read and weight load in the same gap
    MIXQ_WR(8, 3, 16, 5, 2, 0, "128x192_s16_d5_l2"),
    MIXQ_WR8(8, 3, 16, 6, 2, "128x192_s16_d6_l2"),        // (deeper weight rings for the cold-weights protocol: tools/trace_gemm.py --cold 8)
    MIXQ_WR8(8, 3, 12, 6, 2, "128x192_s12_d6_l2"),
    MIXQ_WR(8, 3, 16, 4, 4, 0, "128x192_s16_d4_l4"),
    MIXQ_WR(8, 3, 16, 4, 2, 25, "128x192_p25_w0first"),
    MIXQ_WR(8, 3, 16, 4, 2, 26, "128x192_p26_w0x0first"),
    MIXQ_WR(8, 3, 16, 4, 2, 27, "128x192_p27_w0x0first_long"),
    // EXPERIMENT (round 3, kept for the record, not shipped): the vendor library's prefill shape - 256 x 256 tiles, four fat
    // self-loading waves with 256 accumulator registers each (SELF form of the kernel), (16 + 16) KB of operands per k-step for
    // 4 x 64 MFMAs, half the L1 / L2 bytes per MFMA of 128 x 256.  Measured at 4096 x 11008 x 4096 without outlier columns: 186.1 us
    // against 191.3 us for 128 x 256 (profiles/r03_prefill_ab.txt) - the loop is not feed-bound there, the part is at its 1.4 kW
    // power cap either way - and its one-register-set tail costs 33 us with 41 outlier columns.  Known fault: memory access
    // violation at M = 8192 and with > 64 outlier columns.  int8 only.
    { "wr256x256_s6_d3_self", 16, 4, 6, 0, gemm_wreg_kernel<16, 4, 6, 3, 0, 0, 0>, nullptr, nullptr, 0 },
#endif
#endif
};
constexpr int WR_SMALL = 14;
constexpr int WR_KSPLIT = 15;
constexpr int NUM_WR = sizeof(g_wr) / sizeof(g_wr[0]);
#ifdef MIXQ_TUNING
int g_wr_krot = 0;
#else
constexpr int g_wr_krot = 0;
#endif

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

int mixq_wr_num_configs() { return NUM_WR; }
#ifdef MIXQ_TUNING
int mixq_wr_set_krot(int v) { if (v < 0) return MIXQ_EINVAL; g_wr_krot = v; return MIXQ_OK; }
#else
int mixq_wr_set_krot(int v) { return v ? MIXQ_EINVAL : MIXQ_OK; }
#endif
const char* mixq_wr_config_name(int c) { return (c >= 0 && c < NUM_WR) ? g_wr[c].name : "?"; }

// Estimated time = rounds over the device's CUs x (k-steps x time per k-step of one tile + the tile's fixed prologue / epilogue), both
// fitted on an MI355X at M = 512 over the Llama-2-7b / 70b and Llama-3-8B shapes (tools/sweep_gemm.py, profiles/r02_sweep_shapes.txt):
// per shape this picks the measured winner (64x128 at N = 4096, 64x192 at 6144, 128x128 at 8192, 128x192 at 10-12 k, 128x256 at
// 14 k and 28 k).  ONE table for every function below (round 5 had four copies); the CU count is the device's (mixq_num_cus: a
// partitioned part - CPX mode - has fewer than 256, and whole rounds of ITS compute units are what a launch pays for).
namespace {
struct WrCost { int cfg; float tk, fixed; };
// int8 (and, where a nibble form exists, nibble-packed int4) tilings: us per 64-byte k-step of one tile, fixed us per tile
const WrCost g_cost8[] = {{0, 0.248f, 8.8f}, {4, 0.218f, 5.5f}, {5, 0.341f, 7.1f}, {6, 0.144f, 4.5f}, {7, 0.19f, 4.4f}, {8, 0.235f, 4.4f},
                          {9, 0.16f, 3.9f}, {10, 0.089f, 1.65f}};
// the FP6 form (k-steps of 128 elements, priced per 64 bytes of nibbles like the others): the tilings that exist in it (128 x 256 does not fit the registers)
const WrCost g_cost6[] = {{0, 0.37f, 10.0f}, {4, 0.304f, 7.5f}, {6, 0.23f, 5.1f}, {7, 0.29f, 5.9f}, {8, 0.35f, 6.5f}};
template <class F> void wr_for_costs(int bit, F&& f) {
    if (bit == 6) { for (const auto& c : g_cost6) f(c); } else { for (const auto& c : g_cost8) f(c); }
}
double wr_tile_us(int bit, int cfg, int KB) {
    double t = -1.0;
    wr_for_costs(bit, [&](const WrCost& c) { if (c.cfg == cfg) t = (KB >> 6) * static_cast<double>(c.tk) + c.fixed; });
    return t;
}
// cheapest tiling among those `ok` admits for (M, N) output tiles over `cus` compute units; *best_us: its estimate
template <class F> int wr_cheapest(int bit, int M, int N, int KB, int cus, F&& ok, double* best_us = nullptr, double margin = 0.999) {
    double best = 1e30; int bi = -1;
    wr_for_costs(bit, [&](const WrCost& c) {
        const WrConfig& g = g_wr[c.cfg];
        if (!ok(c.cfg, g)) return;
        const int tiles = cdiv(M, g.mb * 16) * cdiv(N, g.wnb * 64);
        const double t = cdiv(tiles, cus) * wr_tile_us(bit, c.cfg, KB);           // (the busiest CU's tiles decide: whole rounds)
        if (t < best * margin) { best = t; bi = c.cfg; }                          // (margin < 1: a later tiling must be clearly cheaper to displace an earlier one)
    });
    if (best_us) *best_us = best;
    return bi;
}
}  // namespace

int mixq_wr_pick(int bit, int M, int N, int KB)
{
    if (M <= 32) return WR_SMALL;                        // a weight stream: one 64-channel panel per workgroup (narrow int8 layers run gemm_skinny.hip instead)
    const int cus = mixq_num_cus();
    if (bit == 6) return wr_cheapest(6, M, N, KB, cus, [](int, const WrConfig& g) { return g.k6 != nullptr; });
    // few tiles (narrow layer, small batch): when 64-row tiles would occupy at most half the CUs and 32-row tiles still fit one round,
    // the 32 x 64 tiling wins (64 x 4096 -> 4096: 9.9 vs 11.1 us, 128 x 4096 -> 4096: 10.2 vs 11.3 us; profiles/r02_gemm_ab_mid_batch.txt)
    if (cdiv(M, 32) * cdiv(N, 64) <= cus && cdiv(M, 64) * cdiv(N, 64) <= cus / 2) return WR_SMALL;
    return wr_cheapest(bit, M, N, KB, cus, [bit](int, const WrConfig& g) { return bit != 4 || g.k4 != nullptr; });   // (bit 4: tilings with a nibble form)
}

// The joint gate / up launch (N = the interleaved rows of both layers): the same model over the tilings that have the paired epilogue.
int mixq_wr_pick_pair(int bit, int M, int N, int KB)
{
    if (M <= 32) return WR_SMALL;
    return wr_cheapest(bit == 6 ? 6 : 8, M, N, KB, mixq_num_cus(), [bit](int, const WrConfig& g) { return (bit == 6 ? g.k6p : g.k8p) != nullptr; });
}
bool mixq_wr_has_pair(int bit, int c) { return c >= 0 && c < NUM_WR && (bit == 6 ? g_wr[c].k6p : g_wr[c].k8p) != nullptr; }
int mixq_wr_ksplit_config() { return WR_KSPLIT; }
// Can the pairwise split-K form (WR_KSPLIT) run (M, N, KB) on the current device?  MIXQ_OK, or why not: both halves of every tile must be
// resident at once (2 x tiles <= CUs), the workspace registered with mixq_gemm_set_workspace must hold a flag word and an int32 slot per tile.
int mixq_wr_ksplit_ok(int M, int N, int KB)
{
    if (WR_KSPLIT >= NUM_WR) return MIXQ_EINVAL;                            // (product library: the form is not compiled in)
    const WrConfig& g = g_wr[WR_KSPLIT];
    const int tiles = cdiv(M, g.mb * 16) * cdiv(N, g.wnb * 64);
    if ((KB >> 6) < 2) return MIXQ_ESHAPE;
    if (2 * tiles > mixq_num_cus()) return MIXQ_ESHAPE;
    void* ws; size_t bytes, fb;
    if (!mixq_ws_get(&ws, &bytes, &fb)) return MIXQ_EINVAL;
    if (static_cast<size_t>(tiles) * 4 > fb || fb + static_cast<size_t>(tiles) * (g.mb * 16) * (g.wnb * 64) * 4 > bytes) return MIXQ_EINVAL;
    return MIXQ_OK;
}
// ... and is it the faster choice?  The model of mixq_wr_pick with the MEASURED cost of a k-step when all 256 CUs run 128 x 128 tiles
// (0.28 us - not the 0.218 us fitted on 128 such tiles over 256 CUs: the part is power-bound, time follows the MFMA work, not the bytes)
// plus the hand-off (2.9 us, r03_handoff_ubench.txt).  At the shapes it was built for it does not pay: 32.6 vs 28.8 us at 11008 ->
// 4096, 38.6 vs 35.8 us at 14336 -> 4096 against 64 x 128 (profiles/r03_splitk_ab.txt; re-measured in round 6: profiles/r06_splitk_ab.txt).
// The form stays selectable by configuration (tuning build).
bool mixq_wr_ksplit_pays(int M, int N, int KB)
{
    if (M <= 32 || mixq_wr_ksplit_ok(M, N, KB) != MIXQ_OK) return false;
    const int nk = KB >> 6;
    const double t_split = (nk - nk / 2) * 0.28 + 5.5 + 2.9;
    double best = 1e30;
    wr_cheapest(8, M, N, KB, mixq_num_cus(), [](int, const WrConfig&) { return true; }, &best, 1.0);
    return t_split < 0.97 * best;
}

// N split for the last, partial round of tiles.  A launch of T tiles on C compute units takes ceil(T / C) rounds of the tile's time whatever the
// last round holds: 4096 x 11008 x 4096 on 128 x 256 tiles is 1376 tiles = 5.4 rounds of 256 and pays for 6 (profiles/r05_prefill_sweep.txt: 179 us
// where the vendor's stream-K schedule takes 176).  The same kernels cover it in TWO launches over disjoint ranges of the weight rows: the
// picked tiling over the columns of its FULL rounds, then whatever tiling the model prices cheapest over the rest (WrArgs::n_begin: the
// tile map of a launch starts at that row; same image, same Y, no partial tile through memory, results bit-identical).  It pays when the
// remainder's cheaper tiles save more than the second launch's floor (2.2 us).
bool mixq_wr_split(int bit, int M, int N, int KB, int c, int* n1, int* c2)
{
    if (c < 0 || c >= NUM_WR || c == WR_SMALL || c == WR_KSPLIT || (bit != 8 && bit != 6) || M <= 32) return false;
    const int cus = mixq_num_cus();
    const WrConfig& g = g_wr[c];
    const int bm = g.mb * 16, bn = g.wnb * 64, tiles_m = cdiv(M, bm), tiles_n = cdiv(N, bn), tiles = tiles_m * tiles_n;
    const int full = tiles / cus;
    if (full < 1 || tiles % cus == 0) return false;
    const int tn1 = full * cus / tiles_m;                                    // column tiles whose tiles fit the full rounds
    if (tn1 <= 0 || tn1 >= tiles_n) return false;
    const double T = wr_tile_us(bit, c, KB);
    if (T <= 0) return false;
    const int nrest = N - tn1 * bn;
    double best = 1e30;
    const int bc = wr_cheapest(bit, M, nrest, KB, cus, [bit](int, const WrConfig& h) { return bit != 6 || h.k6 != nullptr; }, &best, 1.0);
    if (bc < 0) return false;
    const double t_single = (full + 1) * T, t_split = cdiv(tn1 * tiles_m, cus) * T + best + 2.2;
    if (t_split >= 0.975 * t_single) return false;
    *n1 = tn1 * bn; *c2 = bc;
    return true;
}

#ifdef MIXQ_TUNING
// mixq_gemm_set_fuse_probe (timing probe 77): the stand-in quantise phase's input rows, scratch output and counters (2 x tiles_m words, zero), until cleared
static thread_local const uint16_t* t_fx = nullptr;
static thread_local uint8_t* t_fq = nullptr;
static thread_local unsigned int* t_fcnt = nullptr;
static thread_local int t_fK = 0;
void mixq_wr_fuse_probe(const void* x, void* scratch, void* counters, int K) { t_fx = static_cast<const uint16_t*>(x); t_fq = static_cast<uint8_t*>(scratch); t_fcnt = static_cast<unsigned int*>(counters); t_fK = K; }
// mixq_gemm_hint_next_weights (experiment): the image the launch AFTER the next one will stream, for the next launch's loader waves to touch
// (see the loader).  Per host thread, consumed by the next weights-in-registers launch from it.
static thread_local const uint8_t* t_next_w = nullptr;
static thread_local long long t_next_bytes = 0;
void mixq_wr_hint_next(const void* w, long long bytes) { t_next_w = static_cast<const uint8_t*>(w); t_next_bytes = w && bytes > 0 ? bytes : 0; if (!t_next_bytes) t_next_w = nullptr; }
#endif

// bit: 8, 4 (nibble-packed operands) or 6 (int4 as FP6 codes, MIXQ_FMT_F6X128 operands; KB is K / 2 as for bit 4)
int mixq_wr_launch(int c, int bit, const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                   const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                   const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int KB, int act,
                   unsigned long long* trace, hipStream_t st, uint32_t* row_amax, const uint32_t* amax_mask, int n_begin, int n_cols)
{
    if (c < 0 || c >= NUM_WR) return MIXQ_EINVAL;
    if (n_cols < 0) n_cols = N - n_begin;                                   // (the default: every weight row from n_begin on)
    if (n_begin < 0 || (n_begin & 63) || n_cols <= 0 || n_begin + n_cols > N) return MIXQ_EINVAL;
    const bool pair = act == MIXQ_ACT_SILU_PAIR;                            // (the paired epilogue is a kernel form, not a run-time switch)
    if (pair && ((bit != 8 && bit != 6) || (N & 3) || addend)) return MIXQ_EINVAL;
    const WrConfig& g = g_wr[c];
    WrArgs a;
    memset(&a, 0, sizeof(a));
    a.qx = static_cast<const uint8_t*>(q_x); a.qw = static_cast<const uint8_t*>(q_w);
    a.sx = x_scale; a.sw = scale_col; a.xo = x_out; a.wo = w_out; a.n_out_dev = n_out_dev;
    a.addend = addend; a.bias = bias; a.y = y;
    a.M = M; a.N = N; a.KB = KB; a.ldxo = ldxo; a.ldwo = ldwo; a.n_out = n_out; a.lda = lda; a.ldy = ldy; a.act = act;
    const int bm = g.mb * 16, bn = g.wnb * 64;
    a.tiles_m = cdiv(M, bm); a.tiles_n = cdiv(n_cols, bn); a.n_begin = n_begin;
    if (n_begin + n_cols < N && n_cols % bn) return MIXQ_EINVAL;           // (only the launch that ends at N may hold a partial tile)
    a.xblocks = (M + 15) >> 4; a.wblocks = (N + 15) >> 4;
    {
        const int forced_gm = g_wr_krot >> 16;                               // (tuning build: mixq_gemm_set_krot(gm << 16 | krot); 0 = automatic)
        a.gm = forced_gm > 0 ? forced_gm : (a.tiles_m <= 8 ? a.tiles_m : 8);
        if (a.gm > a.tiles_m) a.gm = a.tiles_m;
    }
    {
        auto magic = [](int d) { return d <= 1 ? 0u : static_cast<unsigned int>((1ull << 32) / static_cast<unsigned long long>(d)) + 1u; };
        a.mg_group = magic(a.gm * a.tiles_n); a.mg_gm = magic(a.gm);
        a.mg_last = magic(a.tiles_m % a.gm ? a.tiles_m % a.gm : a.gm);
        if (static_cast<long long>(a.tiles_m) * a.tiles_n * (a.gm * static_cast<long long>(a.tiles_n)) >= (1ll << 32)) return MIXQ_ESHAPE;   // (exactness bound of the reciprocals)
    }
    a.trace = trace;
    a.row_amax = row_amax; a.amax_mask = amax_mask;
#ifdef MIXQ_TUNING
    if (t_next_w && g.loaders && c != WR_KSPLIT) {
        static const int mode = [] { const char* e = getenv("MIXQ_PF_MODE"); return e ? atoi(e) : 2; }();
        a.pf = t_next_w; a.pf_lines = static_cast<unsigned int>(t_next_bytes >> 7); a.pf_mode = mode;
    }
    t_next_w = nullptr; t_next_bytes = 0;                                    // (a hint serves ONE launch)
    if (t_fx && t_fq && t_fcnt && t_fK > 0 && (t_fK & 7) == 0 && n_begin == 0 && n_cols == N) { static const int ff = [] { const char* e = getenv("MIXQ_FUSE_FLAGS"); return e ? atoi(e) : 0; }(); a.fx = t_fx; a.fq = t_fq; a.fcnt = t_fcnt; a.fK = t_fK; a.fflags = ff; }
#endif
    void (*k)(const WrArgs) = pair ? (bit == 8 ? g.k8p : g.k6p) : (bit == 8 ? g.k8 : (bit == 6 ? g.k6 : g.k4));
    if (!k) return MIXQ_EINVAL;
    int units = a.tiles_m * a.tiles_n;
    if (c == WR_KSPLIT) {
        if (n_begin || n_cols != N) return MIXQ_EINVAL;
        if (int rc = mixq_wr_ksplit_ok(M, N, KB)) return rc;
        void* ws; size_t bytes, fb;
        mixq_ws_get(&ws, &bytes, &fb);
        a.ks_flags = static_cast<unsigned int*>(ws);
        a.ks_slots = static_cast<uint8_t*>(ws) + fb;
        units *= 2;
    }                                                  // (the prefill tiles have no nibble form, few tilings an FP6 form)
    const size_t ring = (bit == 6 ? static_cast<size_t>(g.nstage6) * g.mb * 1536 + 4 * g.mb * 1024 : static_cast<size_t>(g.nstage + (g.loaders ? 2 : 0)) * g.mb * 1024) + 64 /* panel flags */ + static_cast<size_t>(bm) * 4 + static_cast<size_t>(bn) * 8 /* scales */, stg = ((static_cast<size_t>(bm) * (bn * 2 + 16) + 15) & ~static_cast<size_t>(15)) + 16 * bm * 4;   // ring + the tail's X_out blocks | staging tile + row-maximum slots
#ifdef MIXQ_TUNING
    const size_t shm0 = ring > stg ? ring : stg, shm = shm0 + static_cast<size_t>(bn) * 64 <= 160 * 1024 ? shm0 + static_cast<size_t>(bn) * 64 : shm0;   // (room for probe 76's weight blocks where it fits)
#else
    const size_t shm = ring > stg ? ring : stg;
#endif
    if (int rc = mixq_ensure_dynamic_lds(reinterpret_cast<const void*>(k), shm)) return rc;
    hipLaunchKernelGGL(k, dim3(units), dim3((WR_CW + g.loaders) * 64), shm, st, a);
    return mixq_launch_status();
}
