/*
 * mixq_oracle.c — CPU restatement of the MixQ quantized-Linear hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file.  The
 * product (mixq_amd/) never does: it has no CPU path and fails loudly without the HIP library.
 *
 * PARITY STATUS: "partially pinned".
 *   - The Python-level arithmetic of the reference (pack_to_i4, from_linear 8/4-bit, FindOutliers, the forward
 *     state machine) is pinned: oracle/gen_golden.py imports /root/reference/mixquant/modules/linear.py and
 *     Cache.py unmodified (natives stubbed), runs them on seeded inputs and commits inputs+outputs under
 *     tests/golden/; tests/test_oracle_golden.py checks this file against those vectors bit-exactly.
 *   - The arithmetic INSIDE the reference's native module `mixlib` is UNPINNED: its source (github
 *     Qcompiler/QComplier, quantkernel/, no version/commit recorded anywhere in the reference: README.md:39-49,
 *     requirements.txt:1-3) is absent from /root/reference and the reference holds no test vectors for it.
 *     Those kernels are restated from their call sites and the algebra of mixquant/models/sample.py:4-12;
 *     rounding conventions the reference cannot pin are fixed by decision and documented at each function.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference/).
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; no reference sources are compiled or copied).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- IEEE binary16 <-> binary32, round-to-nearest-even ------------------------------------------------------- */
static float h2f(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (exp + 112u) << 23 | man << 13;
    float f; memcpy(&f, &bits, 4); return f;
}

static uint16_t f2h(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);            /* >= 65520 rounds to inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                         /* <= 2^-25 rounds to 0 (tie -> even 0) */
    int e = (int)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                       /* subnormal halves lose more bits */
    uint32_t half_man = man >> shift, rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) ++half_man;
    uint32_t out = (e < -14) ? half_man : (((uint32_t)(e + 15) << 10) + (half_man - 0x400u));
    return (uint16_t)(sign | out);                                       /* mantissa carry propagates into exp */
}

void orc_half_to_float(const uint16_t* h, float* f, long n) { for (long i = 0; i < n; ++i) f[i] = h2f(h[i]); }
void orc_float_to_half(const float* f, uint16_t* h, long n) { for (long i = 0; i < n; ++i) h[i] = f2h(f[i]); }

/* ---- a1: two_compl + pack_to_i4 (mixquant/modules/linear.py:12-18) ------------------------------------------
 * byte[j] = u(x[2j]) | (u(x[2j+1]) << 4), u(v) = v < 0 ? v + 16 : v.   x int8 in [-8,7], K even. */
void orc_pack_i4(const int8_t* x, uint8_t* out, int R, int K)
{
    for (int r = 0; r < R; ++r)
        for (int j = 0; j < K / 2; ++j) {
            int lo = x[(size_t)r * K + 2 * j], hi = x[(size_t)r * K + 2 * j + 1];
            if (lo < 0) lo += 16;
            if (hi < 0) hi += 16;
            out[(size_t)r * (K / 2) + j] = (uint8_t)((lo | (hi << 4)) & 0xff);
        }
}

static int nib_at(const uint8_t* row, int c)
{
    int b = row[c >> 1], nib = (c & 1) ? (b >> 4) : (b & 0xf);
    return nib >= 8 ? nib - 16 : nib;
}

/* a2: unpack_int8_to_int4 -> mixlib.unpack_int4_to_fp16(weight, ind) (linear.py:20-22): sign-extended int4
 * weight columns `ind` as fp16 [N,n]. */
void orc_unpack_i4_cols(const uint8_t* w, const int32_t* ind, int n, uint16_t* out, int N, int K)
{
    for (int r = 0; r < N; ++r)
        for (int j = 0; j < n; ++j)
            out[(size_t)r * n + j] = f2h((float)nib_at(w + (size_t)r * (K / 2), ind[j]));
}

/* ---- a4: from_linear 8-bit (linear.py:111-119) ---------------------------------------------------------------
 * scale = fp16( rowabsmax(W) / 127 )   [torch: max over dim 1 in W's dtype, `/127` in that dtype, .to(fp16)]
 * q     = int8( round( W / scale ) )   [torch: `tmp /= scale_col.T` in W's dtype (fp16 here: one rounding to
 *                                       fp16 per element), .round() = half-to-even, .to(int8)]
 * W fp16 [N,K].  The reference's weights are fp16 when quantising checkpoints (quantize/mixquant.py:177). */
void orc_quant_weight_w8(const uint16_t* w, int N, int K, int8_t* q, uint16_t* scale_col)
{
    for (int r = 0; r < N; ++r) {
        float amax = 0.f;
        for (int k = 0; k < K; ++k) { float a = fabsf(h2f(w[(size_t)r * K + k])); if (a > amax) amax = a; }
        uint16_t sh = f2h(h2f(f2h(amax / 127.0f)));          /* fp16 division result (amax is an fp16 value) */
        scale_col[r] = sh;
        float s = h2f(sh);
        for (int k = 0; k < K; ++k) {
            float d = h2f(f2h(h2f(w[(size_t)r * K + k]) / s)); /* fp16 division: exact quotient rounded once to fp16 */
            q[(size_t)r * K + k] = (int8_t)rintf(d);
        }
    }
}

/* a5: from_linear 4-bit (linear.py:123-143).  ind = the fp_features input channels with the largest
 * layer_scales, in ascending-scale order (torch.sort(...)[1][-fp:]; computed by the caller, ties are the caller's
 * business); weight_cache = W[:,ind]; W[:,ind] = 0; scale = fp16(rowabsmax/10); q = clamp(round(W/scale),-8,7);
 * nibble-pack. */
void orc_quant_weight_w4(const uint16_t* w, int N, int K, const int32_t* ind, int nf, uint8_t* q_packed,
                         uint16_t* scale_col, uint16_t* weight_cache)
{
    int8_t* row = (int8_t*)malloc((size_t)K);
    uint8_t* mask = (uint8_t*)calloc((size_t)K, 1);
    for (int j = 0; j < nf; ++j) mask[ind[j]] = 1;
    for (int r = 0; r < N; ++r) {
        const uint16_t* wr = w + (size_t)r * K;
        for (int j = 0; j < nf; ++j) weight_cache[(size_t)r * nf + j] = wr[ind[j]];
        float amax = 0.f;
        for (int k = 0; k < K; ++k) if (!mask[k]) { float a = fabsf(h2f(wr[k])); if (a > amax) amax = a; }
        uint16_t sh = f2h(h2f(f2h(amax / 10.0f)));
        scale_col[r] = sh;
        float s = h2f(sh);
        for (int k = 0; k < K; ++k) {
            float v = mask[k] ? 0.f : h2f(wr[k]);
            float d = rintf(h2f(f2h(v / s)));                 /* 0/0 -> NaN exactly as torch would; callers avoid zero rows */
            if (d < -8.f) d = -8.f;
            if (d > 7.f) d = 7.f;
            row[k] = (int8_t)d;
        }
        orc_pack_i4(row, q_packed + (size_t)r * (K / 2), 1, K);
    }
    free(row); free(mask);
}

/* ---- a6: FindOutliers (linear.py:157-161): sorted distinct columns with any |x| > sigma (fp16 compare) --------
 * returns the count; ind_out must hold K entries. */
int orc_find_outliers(const uint16_t* x, int M, int K, int ldx, float sigma, int32_t* ind_out)
{
    float thr = h2f(f2h(sigma));                              /* self.sigma is an fp16 [1,1] tensor (linear.py:82-84) */
    int n = 0;
    for (int k = 0; k < K; ++k) {
        int hit = 0;
        for (int m = 0; m < M && !hit; ++m) hit = fabsf(h2f(x[(size_t)m * ldx + k])) > thr;
        if (hit) ind_out[n++] = k;
    }
    return n;
}

/* ---- k2: mixlib.ExtractOutliersAndSetToZeros(ind, x) (call sites linear.py:189,205) ---------------------------
 * x_out[m,j] = x[m,ind[j]]; x[m,ind[j]] = 0 in place (the caller's tensor is mutated: probe in SURVEY.md §8c). */
void orc_extract_outliers_zero(uint16_t* x, const int32_t* ind, int n, uint16_t* x_out, int M, int K, int ldx, int ldo)
{
    (void)K;
    for (int m = 0; m < M; ++m) {
        for (int j = 0; j < n; ++j) { x_out[(size_t)m * ldo + j] = x[(size_t)m * ldx + ind[j]]; x[(size_t)m * ldx + ind[j]] = 0; }
        for (int j = n; j < ldo; ++j) x_out[(size_t)m * ldo + j] = 0;
    }
}

/* ---- k1: mixlib.FindRowScale(x, x_scale, M, K, bit) -> q (call sites linear.py:190-193,221) -------------------
 * Convention fixed by decision (unpinned in the reference; the consumer check linear.py:201 fixes only "/qmax"):
 *   x_scale[m] = fp16(rowabsmax / qmax) (fp32 divide, one rounding); q = clamp(rint(x / float(x_scale)), +-qmax)
 *   with an fp32 IEEE divide and round-half-even; an all-zero row gives scale 0 and q = 0.
 * bit 8: q int8 [M,K]; bit 4: q uint8 [M,K/2] nibble-packed like the weights. */
void orc_find_row_scale(const uint16_t* x, uint16_t* x_scale, void* q, int M, int K, int ldx, int bit)
{
    const float qmax = (float)((1 << (bit - 1)) - 1);
    int8_t* row = (int8_t*)malloc((size_t)K);
    for (int m = 0; m < M; ++m) {
        const uint16_t* xr = x + (size_t)m * ldx;
        float amax = 0.f;
        for (int k = 0; k < K; ++k) { float a = fabsf(h2f(xr[k])); if (a > amax) amax = a; }
        uint16_t sh = f2h(amax / qmax);
        x_scale[m] = sh;
        float s = h2f(sh);
        for (int k = 0; k < K; ++k) {
            float v = (s > 0.f) ? rintf(h2f(xr[k]) / s) : 0.f;
            if (v > qmax) v = qmax;
            if (v < -qmax) v = -qmax;
            row[k] = (int8_t)v;
        }
        if (bit == 8) memcpy((int8_t*)q + (size_t)m * K, row, (size_t)K);
        else orc_pack_i4(row, (uint8_t*)q + (size_t)m * (K / 2), 1, K);
    }
    free(row);
}

/* The device-side form of the check `cache.x_scale[0:M].max() > self.sigma / qmax` (linear.py:201):
 * both sides fp16 tensors in torch -> compare fp16(max scale) with fp16(fp16(sigma)/qmax). */
int orc_mispredicted(const uint16_t* x_scale, int M, float sigma, int bit)
{
    const float qmax = (float)((1 << (bit - 1)) - 1);
    float thr = h2f(f2h(h2f(f2h(sigma)) / qmax)), mx = 0.f;
    for (int m = 0; m < M; ++m) { float s = h2f(x_scale[m]); if (s > mx) mx = s; }
    return mx > thr;
}

/* ---- k10: q_weight[:,ind].to(fp16) * scale_col.T (linear.py:207) and the int4 twin (linear.py:209-210) -------- */
void orc_dequant_weight_cols(const void* w, const uint16_t* scale_col, const int32_t* ind, int n, uint16_t* out, int N,
                             int K, int ldo, int bit)
{
    for (int r = 0; r < N; ++r)
        for (int j = 0; j < n; ++j) {
            int v = (bit == 8) ? ((const int8_t*)w)[(size_t)r * K + ind[j]] : nib_at((const uint8_t*)w + (size_t)r * (K / 2), ind[j]);
            float f = h2f(f2h((float)v));
            if (scale_col) f = f * h2f(scale_col[r]);           /* fp16 x fp16 -> rounded once to fp16 (torch half mul) */
            out[(size_t)r * ldo + j] = f2h(f);
        }
}

/* ---- k5: mixlib.gemm(q_x, q_w, M, N, K) -> int32 (linear.py:235): exact integer contraction ------------------- */
void orc_gemm_i8(const int8_t* qx, const int8_t* qw, int32_t* y, int M, int N, int K)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const int8_t* a = qx + (size_t)m * K; const int8_t* b = qw + (size_t)n * K;
            int32_t acc = 0;
            for (int k = 0; k < K; ++k) acc += (int32_t)a[k] * (int32_t)b[k];
            y[(size_t)m * N + n] = acc;
        }
}

static float silu_f(float v) { return v / (1.f + expf(-v)); }

/* ---- k3/k4/k6/k7/k9 + bias: the whole compute step of linear.py:244-285 (and :320-373 for SiLU) ---------------
 *   y[m,n] = fp16( act( float(acc) * sx[m] * sw[n] + sum_j xo[m,j]*wo[n,j] + addend[m,n] ) + bias[n] )
 * acc exact int32; everything after in fp32 with ONE final rounding (decision (2) of SURVEY.md §8c; the reference
 * rounds the outlier GEMM to fp16 first (torch.mm, linear.py:248) and adds bias in fp16 (linear.py:285): the two
 * differ by at most ~2 fp16 ulp of the result, inside the 1e-2 gate).
 * bit 8: qx int8 [M,K], qw int8 [N,K];  bit 4: nibble-packed uint8 [M,K/2], [N,K/2]. */
void orc_linear_fused(const void* qx, const void* qw, const uint16_t* sx, const uint16_t* sw, const uint16_t* xo, int ldxo,
                      const uint16_t* wo, int ldwo, int n_out, const uint16_t* addend, int lda, const uint16_t* bias,
                      uint16_t* y, int ldy, int M, int N, int K, int act, int bit)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        int8_t* arow = NULL;
        if (bit == 4) {
            arow = (int8_t*)malloc((size_t)K);
            for (int k = 0; k < K; ++k) arow[k] = (int8_t)nib_at((const uint8_t*)qx + (size_t)m * (K / 2), k);
        }
        for (int n = 0; n < N; ++n) {
            int32_t acc = 0;
            if (bit == 8) {
                const int8_t* a = (const int8_t*)qx + (size_t)m * K; const int8_t* b = (const int8_t*)qw + (size_t)n * K;
                for (int k = 0; k < K; ++k) acc += (int32_t)a[k] * (int32_t)b[k];
            } else {
                const uint8_t* b = (const uint8_t*)qw + (size_t)n * (K / 2);
                for (int k = 0; k < K; ++k) acc += (int32_t)arow[k] * nib_at(b, k);
            }
            float v = (float)acc * h2f(sx[m]) * h2f(sw[n]);
            float o = 0.f;
            for (int j = 0; j < n_out; ++j) o += h2f(xo[(size_t)m * ldxo + j]) * h2f(wo[(size_t)n * ldwo + j]);
            v += o;
            if (act == 2) {   /* MIXQ_ACT_SILU_MUL: (silu(z) + bias) * multiplier - the bias lands BEFORE the product, as in
                               * linear.py:372-373 (`y1 += self.bias`) followed by mlp.py:61 (`gate_output *= up_output`) */
                v = silu_f(v);
                if (bias) v += h2f(bias[n]);
                v *= h2f(addend[(size_t)m * lda + n]);
            } else {
                if (addend) v += h2f(addend[(size_t)m * lda + n]);
                if (act == 1) v = silu_f(v);
                if (bias) v += h2f(bias[n]);
            }
            y[(size_t)m * ldy + n] = f2h(v);
        }
        free(arow);
    }
}

/* The north_star's parity reference: a CPU Linear over the SAME dequantised operands
 *   X^[m,k] = q[m,k]*sx[m] with the outlier columns restored exactly, W^[n,k] = qw[n,k]*sw[n] - except on the outlier
 *   columns, where the operand the reference multiplies is its fp16 `weight_cache` (linear.py:207/:129), passed as wo;
 *   with wo == NULL those columns use qw*sw too.  fp64 accumulate.
 * Independent of the integer path above (no int32 accumulator, no factoring of the scales). */
void orc_linear_dequant_ref(const int8_t* qx, const int8_t* qw, const uint16_t* sx, const uint16_t* sw, const uint16_t* xo,
                            int ldxo, const int32_t* ind, int n_out, const uint16_t* wo, int ldwo, const uint16_t* bias,
                            double* y, int M, int N, int K)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        double* xr = (double*)malloc(sizeof(double) * (size_t)K);
        for (int k = 0; k < K; ++k) xr[k] = (double)qx[(size_t)m * K + k] * (double)h2f(sx[m]);
        for (int j = 0; j < n_out; ++j) xr[ind[j]] = wo ? 0.0 : (double)h2f(xo[(size_t)m * ldxo + j]);
        for (int n = 0; n < N; ++n) {
            double acc = 0.0, s = (double)h2f(sw[n]);
            for (int k = 0; k < K; ++k) acc += xr[k] * ((double)qw[(size_t)n * K + k] * s);
            if (wo)
                for (int j = 0; j < n_out; ++j)
                    acc += (double)h2f(xo[(size_t)m * ldxo + j]) * (double)h2f(wo[(size_t)n * ldwo + j]);
            if (bias) acc += (double)h2f(bias[n]);
            y[(size_t)m * N + n] = acc;
        }
        free(xr);
    }
}

/* ---- §8f row 1: FasterTransformer RMSNorm (+ fused extract / quantise for the next linear) ----------------------
 * Call sites mixquant/modules/fused/norm.py:21-33 (mixlib.layernorm_forward_cuda and
 * mixlib.layernorm_forward_cuda_extract_outliers[_int4]); the kernels themselves are not in /root/reference, so the
 * arithmetic is fixed by decision (see mixq_amd/csrc/norm.hip) and restated here INCLUDING the summation order of the
 * 256-thread kernel, which makes the GPU result reproducible bit for bit:
 *   thread t accumulates chunks t, t+256, ... (8 elements each, in order) with fmaf; per 64-lane wave a butterfly
 *   (xor 32,16,8,4,2,1); then (w0+w1)+(w2+w3);  inv = 1/sqrt(ss/K + eps);  y = fp16((x*inv)*w). */
static float rms_inv(const uint16_t* xr, int K, float eps)
{
    float part[256];
    const int nchunk = K / 8;
    for (int t = 0; t < 256; ++t) {
        float s = 0.f;
        for (int c = t; c < nchunk; c += 256)
            for (int e = 0; e < 8; ++e) { float v = h2f(xr[c * 8 + e]); s = fmaf(v, v, s); }
        part[t] = s;
    }
    float wsum[4];
    for (int w = 0; w < 4; ++w) {
        float a[64], b[64];
        for (int l = 0; l < 64; ++l) a[l] = part[w * 64 + l];
        for (int o = 32; o > 0; o >>= 1) {
            for (int l = 0; l < 64; ++l) b[l] = a[l] + a[l ^ o];
            memcpy(a, b, sizeof(a));
        }
        wsum[w] = a[0];
    }
    float total = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    return 1.0f / sqrtf(total / (float)K + eps);
}

void orc_rmsnorm(const uint16_t* x, const uint16_t* w, uint16_t* out, int M, int K, int ldx, int ldout, float eps)
{
    for (int m = 0; m < M; ++m) {
        const uint16_t* xr = x + (size_t)m * ldx;
        float inv = rms_inv(xr, K, eps);
        for (int k = 0; k < K; ++k) out[(size_t)m * ldout + k] = f2h((h2f(xr[k]) * inv) * h2f(w[k]));
    }
}

/* out = RMSNorm(x) with the columns `ind` zeroed; x_out[m,j] = the normalised value of column ind[j]; x_scale / q from
 * the zeroed row as orc_find_row_scale does.  x itself is not modified. */
void orc_rmsnorm_quant(const uint16_t* x, const uint16_t* w, uint16_t* out, const int32_t* ind, int n, uint16_t* x_scale,
                       void* q, uint16_t* x_out, int M, int K, int ldx, int ldout, int ldxo, float eps, int bit)
{
    orc_rmsnorm(x, w, out, M, K, ldx, ldout, eps);
    orc_extract_outliers_zero(out, ind, n, x_out, M, K, ldout, ldxo);
    orc_find_row_scale(out, x_scale, q, M, K, ldout, bit);
}

/* ---- weight-only W8A16 (modules/linear.py:102-106, :178-184) --------------------------------------
 * The arithmetic lives in EETQ (github.com/NetEase-FuXi/EETQ: `quant_weights`, `w8_a16_gemm`), a third-party CUDA
 * extension that is absent from the reference tree and not version-pinned: PARITY UNPINNED for this row.  Restated from
 * its published algorithm (FasterTransformer's symmetric_quantize_last_axis_of_batched_matrix, which EETQ wraps):
 *   per output column n of W^T [K,N]:  s = colabsmax / 128  (float; the reference stores scales.half(), linear.py:106)
 *   q[k,n] = clip(round(w / s), -128, 127),  round = C round(), half away from zero
 * An all-zero column has s = 0 there (0/0); it is given q = 0 here (the dequantised weight is 0 either way).
 * wkn: fp16 [K,N] (the transposed Linear weight), q: int8 [K,N], scale: fp16 [N]. */
void orc_quant_weight_w8a16(const uint16_t* wkn, int K, int N, int8_t* q, uint16_t* scale)
{
    for (int n = 0; n < N; ++n) {
        float amax = 0.f;
        for (int k = 0; k < K; ++k) { float a = fabsf(h2f(wkn[(size_t)k * N + n])); if (a > amax) amax = a; }
        const float s = amax * (1.0f / 128.0f);
        scale[n] = f2h(s);
        for (int k = 0; k < K; ++k) {
            float v = 0.f;
            if (s > 0.f) { v = roundf(h2f(wkn[(size_t)k * N + n]) / s); v = v < -128.f ? -128.f : (v > 127.f ? 127.f : v); }
            q[(size_t)k * N + n] = (int8_t)v;
        }
    }
}

/* y[m,n] = fp16( float(scale[n]) * sum_k x[m,k] * q[k,n]  (+ bias[n]) ): exact products, double accumulation, one rounding. */
void orc_w8a16_linear(const uint16_t* x, int ldx, const int8_t* q, const uint16_t* scale, const uint16_t* bias, uint16_t* y,
                      int ldy, int M, int N, int K)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        double* accv = (double*)calloc((size_t)N, sizeof(double));
        for (int k = 0; k < K; ++k) {
            const double xv = (double)h2f(x[(size_t)m * ldx + k]);
            if (xv == 0.0) continue;
            const int8_t* qr = q + (size_t)k * N;
            for (int n = 0; n < N; ++n) accv[n] += xv * (double)qr[n];
        }
        for (int n = 0; n < N; ++n) {
            float v = (float)(accv[n] * (double)h2f(scale[n]));
            if (bias) v += h2f(bias[n]);
            y[(size_t)m * ldy + n] = f2h(v);
        }
        free(accv);
    }
}
