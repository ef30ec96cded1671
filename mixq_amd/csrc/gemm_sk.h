// Internal interface between gemm.hip (dispatch) and gemm_sk.hip (stream-K kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

int mixq_sk_num_configs();
const char* mixq_sk_config_name(int c);
size_t mixq_sk_workspace_need(int c, int G);
bool mixq_sk_usable(int c);                  // workspace registered and large enough for config c on this device
int mixq_sk_launch(int c, int bit, const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                   const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                   const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int KB, int act,
                   hipStream_t st);

// the registered workspace of the current device (mixq_gemm_set_workspace): flags region of `*flag_bytes` bytes first (zero between launches),
// slots behind it.  False when none is registered.  Also used by the pairwise split-K form of gemm_wreg.hip.
bool mixq_ws_get(void** ws, size_t* bytes, size_t* flag_bytes);
