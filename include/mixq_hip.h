/*
 * mixq_hip.h — C ABI of libmixq_hip.so: MI355X (gfx950 / CDNA4) kernels for MixQ's mixed-precision
 * quantized Linear (W8A8O16 / W4A4O16).
 *
 * This is the drop-in boundary.  In the reference the hot path sits behind the un-vendored
 * Python extension module `mixlib` (call sites /root/reference/mixquant/modules/linear.py:22,189-193,205,221,
 * 235-283,321-366 and mixquant/modules/fused/norm.py:21-33).  Each entry point below names the `mixlib`
 * function / torch expression it replaces.  The Python shim mixq_amd/mixlib.py maps the reference's names and
 * positional argument orders onto these functions; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers are DEVICE pointers unless the name ends in _host; no torch types cross this boundary.
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     returns immediately; 0 = success, otherwise a hipError_t (>0) or one of the MIXQ_E* codes (<0).
 *   - outputs are caller-allocated.  Nothing here allocates or frees device memory or synchronises the device,
 *     so every entry point is hipGraph-capturable.
 *   - fp16 = IEEE binary16 (uint16_t storage).  "row-major [R,C] ld" = element (r,c) at base[r*ld + c].
 *   - quantisation convention (fixed by decision because the reference cannot pin it):
 *       x_scale[m] = fp16( max_k |x[m,k]| / qmax ),  qmax = 2^(bit-1)-1   (fp32 divide, RNE to fp16)
 *       q[m,k]     = clamp( rint( x[m,k] / float(x_scale[m]) ), -qmax, qmax )  (fp32 IEEE divide, RNE);
 *                    all-zero row -> scale 0, q = 0
 *       int4 pairs are nibble-packed low-nibble = even column, two's complement (linear.py:12-18).
 *       y[m,n]     = fp16( float(acc_i32) * float(x_scale[m]) * float(scale_col[n])
 *                          + sum_j x_out[m,j]*w_out[n,j] (fp32 accumulate)  + addend[m,n] )  -> optional SiLU
 *                          -> + bias[n]; ONE rounding to fp16 at the end.
 */
#ifndef MIXQ_HIP_H
#define MIXQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIXQ_OK 0
#define MIXQ_EINVAL (-1)     /* bad argument (null pointer, bit not in {4,8}, negative size ...) */
#define MIXQ_ESHAPE (-2)     /* shape not supported by the kernels (see each function) */
#define MIXQ_ENODEV (-3)     /* no gfx950 device / kernel image not loadable on the current device */

#define MIXQ_ACT_NONE 0
#define MIXQ_ACT_SILU 1      /* SiLU on (dequant + outlier + addend), bias added afterwards (linear.py:324-373) */
#define MIXQ_ACT_SILU_MUL 2  /* (SiLU(dequant + outlier) + bias) TIMES addend[m,n] (required): gate_proj's epilogue with up_proj's
                              * output as the multiplier - linear.py:372-373 (`y1 += self.bias`) followed by the
                              * `gate_output *= up_output` pass of modules/fused/mlp.py:61-63, folded into the GEMM
                              *  */
#define MIXQ_ACT_SILU_PAIR 3 /* gate_proj and up_proj of an MLP block as ONE GEMM (modules/fused/mlp.py:57-63 in one launch).  The N weight
                              * rows (and scale_col, bias, w_out rows) are the two layers' rows INTERLEAVED in groups of four:
                              * row 4g+0 = up[2g], 4g+1 = up[2g+1], 4g+2 = gate[2g], 4g+3 = gate[2g+1].  y has N/2 columns:
                              * y[m,c] = fp16((SiLU(z_gate[m,c]) + bias_gate[c]) * fp16(z_up[m,c] + bias_up[c])), z = dequant + outlier -
                              * bit for bit what up_proj's launch followed by gate_proj's MIXQ_ACT_SILU_MUL launch leave.  int8,
                              * MIXQ_FMT_F16X64 weights with MIXQ_FMT_P16X64 activations, N % 16 == 0, ldy >= N/2, no addend; other
                              * operands answer MIXQ_ESHAPE (the caller then runs the two launches) */

/* Quantised-operand storage formats.
 * MIXQ_FMT_PLAIN  : row-major [R, KB] bytes (KB = K for int8, K/2 for nibble-packed int4) - the reference's layout.
 * MIXQ_FMT_P16X64 : tile-major "P16x64": [KB/64][rows16/16] blocks of 16 rows x 64 bytes (1 KiB each), rows16 =
 *                   roundup(R,16); inside a block row r (0..15) stores its four 16-byte chunks c at position
 *                   c ^ (-(r>>2) & 3).  A block is byte-for-byte the LDS image the GEMMs' fragment reads expect (conflict-free for
 *                   the 32x32x32 and the 16x16x64 int8 MFMA operand shapes alike), a row's 64 bytes stay contiguous, so one
 *                   LDS-DMA instruction moves one contiguous KiB and a k-step of a tile is one contiguous run
 *                   (1.7x the operand-feed rate of 64-byte row segments on MI355X).  KB % 64 == 0. */
#define MIXQ_FMT_PLAIN  0
#define MIXQ_FMT_P16X64 1
/* MIXQ_FMT_F16X64 : "fragment order", the WEIGHT format of the weights-in-registers kernels (gemm_wreg.hip): the same
 *                   [KB/64][rows16/16] grid of 1 KiB blocks, but inside a block the four 16-byte k-chunks c are the slow index:
 *                   byte c*256 + r*16 + b.  Lane l of a wave that loads 16 bytes at block + 16 l holds row l & 15, k-chunk
 *                   l >> 4 - one v_mfma_i32_16x16x64_i8 operand - so a weight fragment is ONE fully coalesced
 *                   global_load_dwordx4 with no LDS round trip.  (Activations stay in P16X64: the quantise kernels write a
 *                   row's 64 bytes contiguously.) */
#define MIXQ_FMT_F16X64 2
/* MIXQ_FMT_F6X128 : int4 operands (bit = 4 only) stored as FP6 E3M2 codes in fragment order, for BOTH operands of the W4A4 GEMM on
 *                   the FP6 matrix pipe: gfx950 has no int4 MFMA, but every integer of [-8, 8] is an E3M2 value (sign | 3-bit exponent,
 *                   bias 3 | 2-bit mantissa: 0 1 2 3 4 5 6 7 8 = 0x00 0x0c 0x10 0x12 0x14 0x15 0x16 0x17 0x18; the reference's 4-bit
 *                   weights are clamp(round(w / scale), -8, 7), linear.py:139), the products are integers and the fp32 accumulator is
 *                   exact below 2^24 (K < 262144), so v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales returns the
 *                   int4 x int4 -> int32 contraction bit for bit at 1.6x the rate of the int8 MFMA (profiles/r03_ubench_fp6.txt).
 *                   [K/128][rows16/16] blocks of 16 rows x 128 elements = 1536 bytes.  Lane l of a wave owns row l & 15, elements
 *                   32 (l >> 4) .. + 31 of the block: 32 six-bit codes as a little-endian bit stream (element e at bit 6 e) = 24
 *                   bytes, stored as a 16-byte piece at block + 16 l and an 8-byte piece at block + 1024 + 8 l (one dwordx4 + one
 *                   dwordx2 per lane, both fully coalesced).  The image of an [R, K] operand takes rows16 x 3K/4 bytes.  K % 128 == 0.
 *                   mixq_pack_operand / mixq_unpack_operand (fmt = MIXQ_FMT_F6X128) convert from / to the plain nibble-packed
 *                   [R, K/2] matrix (KB = K/2 in their signature). */
#define MIXQ_FMT_F6X128 3
/* MIXQ_FMT_R6X128 : the ACTIVATION side of the same GEMM: the same blocks and the same 24-byte lane fragments, ROW-MAJOR inside a block: row r
 *                   (= row % 16) owns the 96 bytes at 96 r - its four 16-byte pieces (fragment g at + 16 g), then its four 8-byte pieces
 *                   (+ 64 + 8 g) - because a quantise kernel owns ONE row: it writes whole 96-byte runs here, against 16-byte and 8-byte
 *                   pieces at 256- and 128-byte strides in fragment order.  The GEMM's LDS-DMA copies whole KiBs and swaps the two
 *                   16-byte units of every aligned 32-byte pair in rows 8 .. 15 on the way (the source address of a DMA lane is free):
 *                   that image serves its b128 and b64 fragment reads without bank conflicts. */
#define MIXQ_FMT_R6X128 4
/* `layout` bits of the GEMM entry points */
#define MIXQ_X_PACKED 1      /* q_x is MIXQ_FMT_P16X64 */
#define MIXQ_W_PACKED 2      /* q_w is MIXQ_FMT_P16X64 */
#define MIXQ_W_F16X64 8      /* q_w is MIXQ_FMT_F16X64 (requires MIXQ_X_PACKED) */
#define MIXQ_XW_F6X128 16    /* mixq_gemm_i4_fused only: q_x is MIXQ_FMT_R6X128 and q_w MIXQ_FMT_F6X128 (no other layout bit) */

typedef void* mixq_stream_t; /* hipStream_t */

/* Library / device identification.  mixq_version: ABI version (major*1000 + minor). */
int mixq_version(void);
/* Writes up to `cap` bytes of a NUL-terminated description ("gfx950 cu=256 ...") ; returns MIXQ_ENODEV without GPU. */
int mixq_device_info(char* buf_host, int cap);

/* ---- (i) per-token absmax + quantise ------------------------------------------------------------------
 * Replaces mixlib.FindRowScale(x, x_scale, M, K, bit) -> q_x   (linear.py:190-193, :221).
 *   x        fp16 [M,K] row-major, ldx elements between rows (read only); values must be finite (the result for
 *            inf / nan inputs is not specified - the reference does not define it either)
 *   x_scale  fp16 [M]   written in place (the reference passes its cache.x_scale[inputdim,1] buffer)
 *   q        bit=8: int8 [M,K];  bit=4: uint8 [M,K/2] nibble-packed.  K % 8 == 0 (bit 8) / K % 16 == 0 (bit 4).
 *   qfmt     MIXQ_FMT_PLAIN, or MIXQ_FMT_P16X64 (what the GEMMs take) / MIXQ_FMT_F16X64 to emit q directly in a tile-major layout (buffer of
 *            roundup(M,16) * KB bytes; rows >= M are left untouched).
 */
int mixq_find_row_scale(const uint16_t* x, uint16_t* x_scale, void* q,
                        int M, int K, int ldx, int bit, int qfmt, mixq_stream_t stream);

/* ---- outlier extraction -------------------------------------------------------------------------------
 * Replaces mixlib.ExtractOutliersAndSetToZeros(ind, x) -> x_out   (linear.py:189, :205).
 *   ind    int32 [n] column ids in [0,K) (distinct)
 *   x      fp16 [M,K] ldx : columns `ind` are ZEROED IN PLACE (the reference mutates the caller's tensor)
 *   x_out  fp16 [M,ldo]   : x_out[m,j] = x[m,ind[j]] for j < n; columns n..ldo-1 are written with zeros
 */
int mixq_extract_outliers_zero(uint16_t* x, const int32_t* ind, int n, uint16_t* x_out,
                               int M, int K, int ldx, int ldo, mixq_stream_t stream);

/* ---- (i)+(ii) fused: extract + zero + absmax + quantise + misprediction flag, one pass over X ---------
 * Replaces the k2+k1 pair of linear.py:187-193 and the device-side part of the check linear.py:201.
 *   n_dev    optional int32* device scalar holding the live outlier count (overrides n when non-NULL;
 *            must be <= n, which then is the capacity of `ind`)
 *   flag     optional int32* device scalar: set to 1 (atomic OR) when any x_scale[m] > fp16(sigma/qmax),
 *            i.e. the reference's `cache.x_scale[0:M].max() > self.sigma / qmax` evaluated on device.
 *   sigma    outlier threshold (reference: 6)
 */
int mixq_quant_fused(uint16_t* x, const int32_t* ind, int n, const int32_t* n_dev,
                     uint16_t* x_scale, void* q, uint16_t* x_out, int32_t* flag,
                     int M, int K, int ldx, int ldo, int bit, float sigma, int qfmt, mixq_stream_t stream);
/* The same pass for a caller that KEEPS an OUTLIER MAP of its columns in device memory (a layer whose prediction is frozen: `ind` never
 * changes again).  col_mask = int32 words:
 *     [W = ceil(K / 32) little-endian words : bit c set <=> column c is one of ind[0 .. live count)]
 *     [1 word                               : the number of columns the map marks]
 *     [pad to a multiple of 4 words]
 *     [K uint16, two per word, low half first : keep[c] = 0x0000 for an outlier column, 0xffff for every other column]
 * (required when n > 0).  Same bytes out.  With the map every byte the pass needs arrives in ONE memory round trip: the eight AND-masks of
 * a 16-byte chunk are requested beside the chunk; a chunk that holds a marked column goes to an LDS image of the row as a whole and is
 * zeroed by four ANDs (it goes back to x), and lane j reads element ind[j] of that image for x_out[row][j] behind the barrier of the row
 * maximum - the row maximum waits for neither the device-resident count, nor `ind`, nor a mask build in LDS, there is no dependent
 * gather ind[j] -> x[row][ind[j]] behind the row load, and no per-element branch in an issue-bound pass (profiles/r05_quant_pmc.txt).  A map built for another count than the live one (n, or
 * *n_dev when given - device code may lower it without the host knowing) is ignored: the kernel then builds its own mask, as
 * mixq_quant_fused does; so is a map when the row's fp16 image does not fit the default 64 KB of LDS (K > 30000).  A producing GEMM reads only the bit words
 * (mixq_gemm_i8_fused_amax).  mixq_linear_forward takes this route when args->col_mask is set. */
/* map_words: the number of int32 words the caller's map holds.  It must be at least mixq_kept_map_words(K) and the map 16-byte aligned,
 * else MIXQ_EINVAL: the layout grew in round 5 (the per-column AND-masks behind the bit words), and a buffer in the older, shorter layout
 * handed to these kernels would be read up to 2 K bytes past its end (the entry points that take a map all carry this argument -
 * mixq_quant_known_amax, mixq_rmsnorm_quant_fused_masked, mixq_linear_args::col_mask_words). */
int mixq_kept_map_words(int K);
int mixq_quant_fused_masked(uint16_t* x, const int32_t* ind, int n, const int32_t* n_dev, const uint32_t* col_mask, int map_words,
                            uint16_t* x_scale, void* q, uint16_t* x_out, int32_t* flag,
                            int M, int K, int ldx, int ldo, int bit, float sigma, int qfmt, mixq_stream_t stream);

/* ---- online outlier-column detection ------------------------------------------------------------------
 * Replaces torch.unique(torch.where(abs(X) > sigma)[1]).int32  (linear.py:157-161, FindOutliers).
 *   colflags  uint8 [K] scratch (overwritten)
 *   ind_out   int32 [K] : ascending distinct column ids with any |x[m,k]| > fp16(sigma)
 *   count     int32 [1] : number of ids written
 */
int mixq_detect_outlier_cols(const uint16_t* x, float sigma, uint8_t* colflags, int32_t* ind_out,
                             int32_t* count, int M, int K, int ldx, mixq_stream_t stream);

/* ---- weight-column dequantisation for newly found outliers --------------------------------------------
 * Replaces `q_weight[:,ind].to(fp16) * scale_col.T` (linear.py:207,305) and, for bit=4,
 * mixlib.unpack_int4_to_fp16(q_weight, ind) * scale_col.T (linear.py:20-22,209-210).
 *   w          bit=8 int8 [N,K]; bit=4 uint8 [N,K/2]
 *   scale_col  fp16 [N], or NULL to get the raw integer values as fp16 (= unpack_int4_to_fp16 alone)
 *   out        fp16 [N,ldo], columns 0..n-1 written: out[r,j] = fp16( fp16(w[r,ind[j]]) * scale_col[r] )
 */
int mixq_dequant_weight_cols(const void* w, const uint16_t* scale_col, const int32_t* ind, int n,
                             uint16_t* out, int N, int K, int ldo, int bit, mixq_stream_t stream);

/* ---- (iii)+(iv) int8 MFMA GEMM with fused dequant / outlier / addend / act / bias epilogue ------------
 * Replaces mixlib.int8FusedDequantize[Silu](q_x, q_w, x_scale, scale_col, addend, M, N, K) (linear.py:251-273,
 * 337-351), the cuBLAS outlier GEMM torch.mm(X_out, weight_cache.T) feeding it (linear.py:248,334) and the
 * separate `y1 += bias` (linear.py:284-285).
 *   q_x [M,K] int8, q_w [N,K] int8 (both K-contiguous), x_scale fp16 [M], scale_col fp16 [N]
 *   x_out fp16 [M,ldxo], w_out fp16 [N,ldwo]: outlier operands (NULL / n_out = 0 for none);
 *         n_out_dev: optional device int32 overriding n_out (<= n_out); ldxo, ldwo >= roundup(n_out,16), % 8 == 0
 *   addend fp16 [M,lda] or NULL;  bias fp16 [N] or NULL;  y fp16 [M,ldy]
 *   layout  MIXQ_X_PACKED | MIXQ_W_PACKED bits (0 = both operands plain row-major as in the reference), or
 *           MIXQ_X_PACKED | MIXQ_W_F16X64 (the weights-in-registers kernels: fragment-order weights, P16X64 activations)
 *   K % 64 == 0, N % 4 == 0, ldy % 4 == 0.
 */
int mixq_gemm_i8_fused(const int8_t* q_x, const int8_t* q_w, const uint16_t* x_scale,
                       const uint16_t* scale_col, const uint16_t* x_out, int ldxo,
                       const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                       const uint16_t* addend, int lda, const uint16_t* bias,
                       uint16_t* y, int ldy, int M, int N, int K, int act, int layout, mixq_stream_t stream);

/* Same for nibble-packed operands: replaces mixlib.int4FusedDequantize[Silu](..., M, N, K/2)
 * (linear.py:259-265, 278-283, 360-366).  q_x uint8 [M,K/2], q_w uint8 [N,K/2]; K is the LOGICAL depth
 * (number of int4 columns), K % 128 == 0. */
int mixq_gemm_i4_fused(const uint8_t* q_x, const uint8_t* q_w, const uint16_t* x_scale,
                       const uint16_t* scale_col, const uint16_t* x_out, int ldxo,
                       const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                       const uint16_t* addend, int lda, const uint16_t* bias,
                       uint16_t* y, int ldy, int M, int N, int K, int act, int layout, mixq_stream_t stream);

/* ---- the next layer's row maximum as a side output of the producing GEMM ------------------------------------------------
 * down_proj's pre-pass (linear.py:187-193) needs max_k |x[m,k]| over the columns it does not extract - a full pass over the
 * [M, 11008] activation that gate_proj's GEMM has just written (mixquant/modules/fused/mlp.py:57-70).  mixq_gemm_i8_fused_amax is
 * mixq_gemm_i8_fused that ALSO maximises, per output row m, the fp16 bit patterns |y[m,n]| over the columns n whose bit in col_mask
 * (bit n of a uint32 array, NULL = no column excluded) is clear, into row_amax[m] with integer atomic max: order-independent, so the
 * value is exactly the maximum the quantiser would have found.  row_amax must be zero when the GEMM starts.
 * mixq_quant_known_amax is mixq_quant_fused for rows whose maximum is known: one pass, no reduction; it reads row_amax[m], CLEARS it
 * (ready for the producer's next run), and writes exactly the bytes mixq_quant_fused writes.  Its col_mask is the layer's KEPT OUTLIER MAP
 * of `ind` (the full layout of mixq_quant_fused_masked: bit words, count word, pad, per-column AND-masks - required when n > 0; the GEMM reads only
 * the bit words of the same buffer).
 * mixq_gemm_amax_supported: 1 when the side output is available for (M, N, K, layout) - int8, MIXQ_X_PACKED | MIXQ_W_F16X64, the
 * batches the weights-in-registers kernels serve - else 0 (mixq_gemm_i8_fused_amax then returns MIXQ_ESHAPE). */
int mixq_gemm_i8_fused_amax(const int8_t* q_x, const int8_t* q_w, const uint16_t* x_scale,
                            const uint16_t* scale_col, const uint16_t* x_out, int ldxo,
                            const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                            const uint16_t* addend, int lda, const uint16_t* bias,
                            uint16_t* y, int ldy, int M, int N, int K, int act, int layout,
                            uint32_t* row_amax, const uint32_t* col_mask, mixq_stream_t stream);
int mixq_gemm_amax_supported(int M, int N, int K, int layout);
int mixq_quant_known_amax(uint16_t* x, const int32_t* ind, int n, const int32_t* n_dev, uint32_t* row_amax,
                          const uint32_t* col_mask, int map_words, uint16_t* x_scale, void* q, uint16_t* x_out, int32_t* flag,
                          int M, int K, int ldx, int ldo, int bit, float sigma, int qfmt, mixq_stream_t stream);

/* ---- the whole forward of a frozen layer in ONE call ---------------------------------------------------------
 * mixq_linear_forward = mixq_quant_fused followed by mixq_gemm_i8_fused / mixq_gemm_i4_fused, both launched from C on `stream`:
 * the steady state of MixLinear_GEMM.forward(x, cache, unfused=True) once outlier prediction has frozen
 * (/root/reference/mixquant/modules/linear.py:187-193 then :244-285; with act / addend the SiLU twin :330-373).  One foreign call
 * and one argument block per forward instead of two calls with 16 + 21 marshalled arguments: what an eager (un-graphed) decoder
 * loop pays per layer (/root/reference/examples/benchbitsand.py:538-550 times exactly that loop).  The block is caller-owned and
 * may be kept and re-used between calls (only x, q_x, x_out, y change from forward to forward).  Results are bit-identical to the
 * two-call sequence.  Field meanings are those of the two entry points above:
 *   x [M,K] fp16 ldx (outlier columns zeroed in place), ind / n_cap / n_dev, x_scale [M] (written), q_x (written, format qfmt),
 *   x_out [M,ldxo] (written; NULL when n_cap = 0), flag (optional), q_w in format wfmt (MIXQ_FMT_*), scale_col [N],
 *   w_out [N,ldwo], addend / lda, bias, y [M,ldy], act (MIXQ_ACT_*), row_amax / col_mask (optional: a producer left the row maxima,
 *   see mixq_gemm_i8_fused_amax - the quantise pass is then mixq_quant_known_amax; col_mask alone: the caller's kept outlier map of
 *   `ind`, the pass is mixq_quant_fused_masked).  qfmt must be what the GEMM takes for wfmt:
 *   PLAIN / P16X64 with wfmt PLAIN / P16X64 in any combination, P16X64 with wfmt F16X64. */
typedef struct mixq_linear_args {
    uint16_t* x; int ldx;
    const int32_t* ind; int n_cap; const int32_t* n_dev;
    uint16_t* x_scale; void* q_x; uint16_t* x_out; int ldxo; int32_t* flag;
    const void* q_w; const uint16_t* scale_col; const uint16_t* w_out; int ldwo;
    const uint16_t* addend; int lda; const uint16_t* bias;
    uint16_t* y; int ldy;
    int M, N, K, bit; float sigma; int act, qfmt, wfmt;
    uint32_t* row_amax; const uint32_t* col_mask;   /* row_amax non-NULL: the rows' maxima are known (mixq_quant_known_amax runs instead) */
    int col_mask_words;                             /* int32 words behind col_mask: >= mixq_kept_map_words(K) (checked whenever col_mask is used) */
} mixq_linear_args;
int mixq_linear_forward(const mixq_linear_args* args, mixq_stream_t stream);

/* ---- operand re-tiling ------------------------------------------------------------------------------------
 * Copy a plain [R,KB] byte matrix (int8 weights, or nibble-packed int4) into MIXQ_FMT_P16X64.
 * dst holds roundup(R,16) * KB bytes; rows >= R are zero-filled.  KB % 64 == 0.  Done once per weight at load. */
int mixq_pack_p16x64(const void* src, void* dst, int R, int KB, mixq_stream_t stream);
/* The same for either packed format (fmt = MIXQ_FMT_P16X64 or MIXQ_FMT_F16X64), and the inverse: mixq_unpack_operand
 * writes the plain [R,KB] matrix back from a packed image (rows >= R of the image are ignored) - what lets an operator keep
 * ONLY the packed weights in HBM and still serve `q_weight` / state_dict in the reference's layout. */
int mixq_pack_operand(const void* src, void* dst, int R, int KB, int fmt, mixq_stream_t stream);
int mixq_unpack_operand(const void* src, void* dst, int R, int KB, int fmt, mixq_stream_t stream);

/* ---- unfused debug pair ---------------------------------------------------------------------------------
 * mixq_gemm_i8 replaces mixlib.gemm(q_x, q_w, M, N, K) -> int32 [M,N]        (linear.py:235,321)
 * mixq_dequant replaces mixlib.dequantizeInt8[Silu](y32, x_scale, scale_col, addend, bit, M, N) (linear.py:238,241,324,327)
 */
int mixq_gemm_i8(const int8_t* q_x, const int8_t* q_w, int32_t* y32, int ldy, int M, int N, int K,
                 mixq_stream_t stream);
int mixq_dequant(const int32_t* y32, int ldy32, const uint16_t* x_scale, const uint16_t* scale_col,
                 const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy,
                 int M, int N, int act, mixq_stream_t stream);

/* ---- RMSNorm in front of the path -----------------------------------------------------
 * mixq_rmsnorm replaces mixlib.layernorm_forward_cuda(x, weight, out, eps)                         (norm.py:21)
 * mixq_rmsnorm_quant_fused replaces mixlib.layernorm_forward_cuda_extract_outliers[_int4](x, weight, out, eps, ind,
 *   x_scale) -> (X_out, q_x)  (norm.py:24-33): out = RMSNorm(x) with the NEXT linear's known outlier columns `ind`
 *   zeroed, x_out[m,j] = the normalised value of column ind[j], x_scale/q of the zeroed row, misprediction flag, q
 *   plain or P16x64 - the activation is read once.  x is not modified.
 * Arithmetic: y = fp16((float(x) * inv_rms) * float(weight)), inv_rms = 1/sqrt(mean(x^2) + eps) in fp32 with the fixed
 * summation order documented in mixq_amd/csrc/norm.hip (restated by the oracle, so results are bit-reproducible).
 *   x fp16 [M,K] ldx, weight fp16 [K], out fp16 [M,K] ldout; K % 8 == 0, K <= 32768. */
int mixq_rmsnorm(const uint16_t* x, const uint16_t* weight, uint16_t* out, int M, int K, int ldx, int ldout, float eps,
                 mixq_stream_t stream);
int mixq_rmsnorm_quant_fused(const uint16_t* x, const uint16_t* weight, uint16_t* out, const int32_t* ind, int n,
                             const int32_t* n_dev, uint16_t* x_scale, void* q, uint16_t* x_out, int32_t* flag,
                             int M, int K, int ldx, int ldout, int ldxo, float eps, int bit, float sigma, int qfmt,
                             mixq_stream_t stream);
/* ... with the next layer's kept outlier map (layout: mixq_quant_fused_masked): same bytes out */
int mixq_rmsnorm_quant_fused_masked(const uint16_t* x, const uint16_t* weight, uint16_t* out, const int32_t* ind, int n,
                                    const int32_t* n_dev, const uint32_t* col_mask, int map_words, uint16_t* x_scale, void* q, uint16_t* x_out,
                                    int32_t* flag, int M, int K, int ldx, int ldout, int ldxo, float eps, int bit, float sigma,
                                    int qfmt, mixq_stream_t stream);

/* ---- weight-only W8A16 Linear -----------------------------------------------------
 * mixq_gemm_w8a16 replaces EETQ's w8_a16_gemm(x, q_weight, scale_col) as called at modules/linear.py:178-184:
 *   y[M,N] fp16 = x[M,K] fp16 * (q[K,N] int8 * scale_col[N] fp16) (+ bias[N] fp16), fp32 accumulation, one fp16 rounding.
 * The kernel streams the weights from a re-tiled copy: mixq_pack_w8a16 turns the checkpoint's row-major [K,N] int8
 * matrix (modules/linear.py:70-71; EETQ's *unprocessed* quantised tensor) into round16(N) * K bytes of offset-binary
 * (q + 128) MIXQ_FMT_F16X64 (rows = output channels) once per layer: the image the kernel's lanes load straight into their
 * MFMA operand registers.
 *   K % 64 == 0, N % 4 == 0, ldx % 8 == 0, x 16-byte aligned, ldy % 4 == 0, y 8-byte aligned. */
int mixq_pack_w8a16(const int8_t* q_weight_kn, uint8_t* packed, int K, int N, mixq_stream_t stream);
int mixq_gemm_w8a16(const uint16_t* x, int ldx, const uint8_t* w_packed, const uint16_t* scale_col, const uint16_t* bias,
                    uint16_t* y, int ldy, int M, int N, int K, mixq_stream_t stream);
int mixq_gemm_w8a16_set_config(int cfg);             /* tuning: force a tile config, -1 = automatic */
int mixq_gemm_w8a16_num_configs(void);
int mixq_gemm_w8a16_config_name(int cfg, char* buf, int cap);

/* ---- stream-K workspace ---------------------------------------------------------------------------------
 * The stream-K form of the GEMM (opt-in through mixq_gemm_set_config) hands
 * int32 partial tiles between workgroups through a device buffer the HOST provides once:
 *   mixq_gemm_workspace_bytes()  size to allocate; mixq_gemm_set_workspace(ptr, bytes) registers it (bytes = 0
 *   unregisters).  The buffer must be zero-filled when registered; every launch leaves its flag words zero.
 * Without a registered workspace the data-parallel kernels are used.  Launches that use the workspace must be
 * serialised on one stream (the operator, like the reference's, is not re-entrant: SURVEY.md §8b Threading). */
long long mixq_gemm_workspace_bytes(void);
int mixq_gemm_set_workspace(void* ws, long long bytes);

/* ---- tuning / introspection (not part of the reference surface) ----------------------------------------
 * Force a GEMM tile configuration id (>= 0) for subsequent mixq_gemm_* calls ON THE CURRENT DEVICE (the state is kept per HIP
 * device), -1 = automatic shape-aware choice.  Returns MIXQ_EINVAL for an unknown id.  mixq_gemm_config_name writes the config's
 * description.  Every configuration of the product library computes the same (correct) result; the ablation forms used for
 * tuning, the in-kernel timeline stamps (mixq_gemm_set_trace), the forced tile-order group (mixq_gemm_set_krot) and the quantise kernel's
 * timing probes (mixq_quant_set_config >= 100) exist only in the -DMIXQ_TUNING build (`make -C mixq_amd/csrc tuning` ->
 * libmixq_hip_tuning.so, loaded by the tools/ scripts): the product library does not export the first two and returns MIXQ_EINVAL for the probes.
 * cfg = -2: automatic, but always ONE launch - without the N split described at mixq_gemm_pick_split (its A/B partner). */
int mixq_gemm_set_config(int cfg);
/* The N split of the automatic choice.  A launch of T tiles on the device's 256 CUs takes ceil(T / 256) rounds of the tile's time
 * whatever the last round holds (4096 x 11008 x 4096 on 128 x 256 tiles: 1376 tiles = 5.4 rounds, paid as 6).  With fragment-order
 * weights (MIXQ_FMT_F16X64 / MIXQ_FMT_F6X128) and no forced configuration the fused GEMMs then run TWO launches over disjoint ranges of
 * the weight rows (output columns): the picked tiling over the columns of its full rounds, the tiling priced cheapest over the rest -
 * same weight image, same Y, no partial tile through memory, bit-identical results (profiles/r05_prefill_sweep.txt: 4-8 % at the
 * shapes it applies to).  Returns 1 and writes the first range's width and the second tiling's id when (M, N, K, bit, fmt) is split,
 * else 0. */
int mixq_gemm_pick_split(int M, int N, int K, int bit, int fmt, int* n1, int* cfg2);
/* Launch geometry of the extract + scale + quantise pass (mixq_quant_fused / mixq_find_row_scale): -1 automatic, 0 the
 * one-row-per-256-thread-workgroup kernel, 1..9 (threads per row, rows per workgroup) = (64,1) (64,2) (64,4) (128,1)
 * (128,2) (256,1) (256,2) (512,1) (512,2).  Every geometry produces identical bytes.  Per device, like mixq_gemm_set_config. */
int mixq_quant_set_config(int cfg);
#ifdef MIXQ_TUNING   /* exported by libmixq_hip_tuning.so only (make -C mixq_amd/csrc tuning): the product library does not have them */
/* Diagnostics: when buf is non-null every workgroup of the data-parallel fused GEMM writes 16 x u64 to
 * buf[16 * workgroup + i]: i in 0..7 = the 100 MHz device wall clock at 0 entry, 1 first stage landed, 2 k loop
 * done, 3 epilogue arithmetic done, 4 stores issued, 5 stores retired, 6/7 inside the epilogue; 8 + i = s_memtime
 * at the same points.  The weights-in-registers kernel also writes a second set 16 x 4096 entries further on (the panel pipeline of its
 * epilogue: tools/trace_gemm.py --panels), so buf needs 2 x 16 x 4096 x 8 bytes. */
int mixq_gemm_set_trace(unsigned long long* buf);
/* Tuning (tools build): bits 16.. = M tiles per group of the weights-in-registers kernels' tile order (0: automatic).  The low 16 bits
 * used to rotate the K walk between neighbouring N tiles (rounds 1-2; it never changed a timing and cost two scalar counters per wave):
 * removed in round 4, ignored. */
int mixq_gemm_set_krot(int krot);
/* Experiment (round 6, measured negative - NOTEBOOK.md, profiles/r06_cold_prefetch_ab.txt): names the weight image (`bytes` long, 128-byte aligned)
 * that the launch AFTER the next weights-in-registers GEMM of this host thread will stream; that next launch's loader waves then touch one
 * dword per 128-byte line of it while its own k loop runs, so that the image is in the 256 MB memory-side cache when its GEMM starts
 * (a model's layers: mixquant/modules/fused/mlp.py:57-70, benchflops.py:112-128).  Forms: environment MIXQ_PF_MODE = 1 scalar-cache loads,
 * 2 vector loads, 6 vector loads with the nt hint; results never depend on it.  One launch per hint; NULL / 0 clears it. */
int mixq_gemm_hint_next_weights(const void* w, long long bytes);
/* Timing probe (round 6, tuning configuration wr128x192_p77_one_launch; NOTEBOOK.md "the one-launch form"): until cleared with NULLs, launches of that configuration from this
 * host thread run a STAND-IN quantise phase in front of the GEMM - x: fp16 rows [M, K], scratch: [M, K] bytes written with write-through stores, counters: 2 x ceil(M / 128)
 * zeroed 32-bit words (one agent-scope add per row, polled by the loader waves, reset by the last poller).  The GEMM's own operands and results are untouched. */
int mixq_gemm_set_fuse_probe(const void* x, void* scratch, void* counters, int K);
#endif
/* Diagnostics: exhaustive self-test of the division-free quantiser used by the quantise kernels (q = rint(x / s) for all
 * finite fp16 x and all finite fp16 s > 0, ~2e9 pairs, ~1 s): adds the number of disagreements with the IEEE-division form
 * to *mismatches_dev (a zeroed device counter).  bit = 8 or 4. */
int mixq_selftest_quant_exact(unsigned long long* mismatches_dev, int bit, mixq_stream_t stream);
int mixq_gemm_num_configs(void);
int mixq_gemm_config_name(int cfg, char* buf_host, int cap);
/* The config the automatic choice picks for (M,N,K,bit) with MIXQ_FMT_P16X64 operands / with operands in `fmt`. */
int mixq_gemm_pick_config(int M, int N, int K, int bit);
int mixq_gemm_pick_config_fmt(int M, int N, int K, int bit, int fmt);

#ifdef __cplusplus
}
#endif
#endif /* MIXQ_HIP_H */
