// The whole forward of a frozen MixQ layer behind ONE C-ABI call: (i)+(ii) the fused extract / zero / absmax / quantise pass, then
// (iii)+(iv) the int8 / int4 MFMA GEMM with its fused epilogue, both enqueued from here on the caller's stream.  Nothing new is
// computed - it is the two entry points of quant.hip and gemm.hip back to back - so results are bit-identical to calling them one
// after the other; what it removes is the second foreign call and the marshalling of 37 arguments per forward on the host, which
// is what an eager decoder loop pays per layer.  Replaces the steady state of /root/reference/mixquant/modules/linear.py:187-193 + :244-285.
#include "common.h"

extern "C" int mixq_linear_forward(const mixq_linear_args* a, mixq_stream_t stream)
{
    if (!a) return MIXQ_EINVAL;
    if (a->bit != 8 && a->bit != 4) return MIXQ_EINVAL;
    int layout;
    switch (a->wfmt) {
        case MIXQ_FMT_PLAIN:  layout = 0; break;
        case MIXQ_FMT_P16X64: layout = MIXQ_W_PACKED; break;
        case MIXQ_FMT_F16X64: layout = MIXQ_W_F16X64; break;
        case MIXQ_FMT_F6X128: layout = MIXQ_XW_F6X128; break;       // W4A4 on the FP6 matrix pipe: both operands as FP6 codes (activations: R6X128)
        default: return MIXQ_EINVAL;
    }
    if ((a->wfmt == MIXQ_FMT_F6X128) != (a->qfmt == MIXQ_FMT_R6X128)) return MIXQ_EINVAL;    // FP6 weights go with row-contiguous FP6 activations
    if (a->wfmt == MIXQ_FMT_F6X128) { if (a->bit != 4) return MIXQ_EINVAL; }
    else if (a->qfmt == MIXQ_FMT_P16X64) layout |= MIXQ_X_PACKED;
    else if (a->qfmt != MIXQ_FMT_PLAIN) return MIXQ_EINVAL;         // (fragment-order activations are a producer-side experiment only)
    if (a->wfmt == MIXQ_FMT_F16X64 && a->qfmt != MIXQ_FMT_P16X64) return MIXQ_EINVAL;
    const bool outl = a->n_cap > 0 && a->x_out && a->w_out;
    if (a->n_cap > 0 && !outl) return MIXQ_EINVAL;                  // known outlier columns need both operands of the tail
    int rc = a->row_amax
        ? mixq_quant_known_amax(a->x, a->ind, a->n_cap, a->n_dev, a->row_amax, a->col_mask, a->col_mask_words, a->x_scale, a->q_x, a->x_out, a->flag, a->M,
                                a->K, a->ldx, a->ldxo, a->bit, a->sigma, a->qfmt, stream)
        : (a->col_mask && a->n_cap > 0         // a frozen layer keeps the mask of its outlier columns: no mask build in front of the row maximum
           ? mixq_quant_fused_masked(a->x, a->ind, a->n_cap, a->n_dev, a->col_mask, a->col_mask_words, a->x_scale, a->q_x, a->x_out, a->flag, a->M, a->K, a->ldx,
                                     a->ldxo, a->bit, a->sigma, a->qfmt, stream)
           : mixq_quant_fused(a->x, a->ind, a->n_cap, a->n_dev, a->x_scale, a->q_x, a->x_out, a->flag, a->M, a->K, a->ldx, a->ldxo,
                              a->bit, a->sigma, a->qfmt, stream));
    if (rc) return rc;
    if (a->bit == 8)
        return mixq_gemm_i8_fused(static_cast<const int8_t*>(a->q_x), static_cast<const int8_t*>(a->q_w), a->x_scale, a->scale_col,
                                  a->x_out, a->ldxo, a->w_out, a->ldwo, a->n_cap, a->n_dev, a->addend, a->lda, a->bias, a->y, a->ldy,
                                  a->M, a->N, a->K, a->act, layout, stream);
    return mixq_gemm_i4_fused(static_cast<const uint8_t*>(a->q_x), static_cast<const uint8_t*>(a->q_w), a->x_scale, a->scale_col,
                              a->x_out, a->ldxo, a->w_out, a->ldwo, a->n_cap, a->n_dev, a->addend, a->lda, a->bias, a->y, a->ldy,
                              a->M, a->N, a->K, a->act, layout, stream);
}
