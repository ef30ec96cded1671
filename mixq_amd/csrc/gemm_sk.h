// Internal interface between gemm.hip (dispatch, per-device registry) and gemm_sk.hip (stream-K kernels: TUNING build only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// per-device state (one process may drive several GPUs), gemm.hip: the workspace registered with mixq_gemm_set_workspace and the CU count
struct MixqDevState { void* ws; size_t bytes; int num_cu; };
MixqDevState* mixq_dev_state();              // of the CURRENT device; nullptr when it cannot be asked
int mixq_num_cus();                          // compute units of the current device (cached per device; 256 when it cannot be asked)
// Workspace layout: MIXQ_WS_FLAG_BYTES of int32 flags (zero on entry; every launch leaves them zero), then the slots.  The flags sit at a
// FIXED place: behind the slots their offset would depend on the configuration's tile size, and one configuration's partial tiles would
// land on another one's flag words.
constexpr size_t MIXQ_WS_FLAG_BYTES = 4096;  // up to 1024 workgroups
// the registered workspace of the current device: flags region of `*flag_bytes` bytes first (zero between launches), slots behind it.
// False when none is registered.  Used by the stream-K kernels and by the pairwise split-K form of gemm_wreg.hip (both: tuning build).
bool mixq_ws_get(void** ws, size_t* bytes, size_t* flag_bytes);

#ifdef MIXQ_TUNING
int mixq_sk_num_configs();
const char* mixq_sk_config_name(int c);
size_t mixq_sk_workspace_need(int c, int G);
bool mixq_sk_usable(int c);                  // workspace registered and large enough for config c on this device
long long mixq_sk_workspace_bytes();
int mixq_sk_launch(int c, int bit, const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                   const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                   const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int KB, int act,
                   hipStream_t st);
#else                                        // the product library holds no stream-K kernel: the dispatcher's questions answer themselves
inline int mixq_sk_num_configs() { return 0; }
inline const char* mixq_sk_config_name(int) { return ""; }
inline bool mixq_sk_usable(int) { return false; }
inline long long mixq_sk_workspace_bytes() { return 0; }
inline int mixq_sk_launch(int, int, const void*, const void*, const uint16_t*, const uint16_t*, const uint16_t*, int, const uint16_t*, int, int,
                          const int32_t*, const uint16_t*, int, const uint16_t*, uint16_t*, int, int, int, int, int, hipStream_t) { return -1; }
#endif
