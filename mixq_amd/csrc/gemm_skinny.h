// Internal interface between gemm.hip (dispatch) and gemm_skinny.hip (small-batch kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

bool mixq_skinny_applies(int bit, int M, int N, int KB, bool x_packed, bool w_packed);
int mixq_skinny_launch(int bit, const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col, const uint16_t* x_out,
                       int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev, const uint16_t* addend,
                       int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int KB, int act, int wf16, hipStream_t st);
