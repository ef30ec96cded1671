// Internal interface between gemm.hip (dispatch) and gemm_wreg.hip (weights-in-registers kernels, MIXQ_FMT_F16X64 operands).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int mixq_wr_num_configs();
int mixq_wr_set_krot(int v);                 // k-step rotation between neighbouring N tiles (tuning; results do not depend on it)
const char* mixq_wr_config_name(int c);
// config for (M, N, KB) or -1 when the data-parallel kernels of gemm.hip should run instead
int mixq_wr_pick(int bit, int M, int N, int KB);
int mixq_wr_launch(int c, int bit, const void* q_x, const void* q_w, const uint16_t* x_scale, const uint16_t* scale_col,
                   const uint16_t* x_out, int ldxo, const uint16_t* w_out, int ldwo, int n_out, const int32_t* n_out_dev,
                   const uint16_t* addend, int lda, const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int KB, int act,
                   unsigned long long* trace, hipStream_t st, uint32_t* row_amax = nullptr, const uint32_t* amax_mask = nullptr,
                   int n_begin = 0, int n_cols = -1);     // the launch covers the weight rows [n_begin, n_begin + n_cols) (default: all from n_begin)
#ifdef MIXQ_TUNING
// (experiment) the weight image of the launch after the next one: the next launch's loader waves touch it (one launch, this host thread)
void mixq_wr_hint_next(const void* w, long long bytes);
// (timing probe 77) the stand-in quantise phase of the one-launch form: fp16 rows [M, K], scratch [M, K] bytes, 2 x tiles_m zeroed counter words; NULL clears
void mixq_wr_fuse_probe(const void* x, void* scratch, void* counters, int K);
#endif
// N split for a partial last round of tiles: true when running tiling c over the first *n1 weight rows and tiling *c2 over the rest is priced
// cheaper than one launch of c (gemm_wreg.hip)
bool mixq_wr_split(int bit, int M, int N, int KB, int c, int* n1, int* c2);
// pairwise split-K form (two workgroups per 128 x 128 tile, half of K each, int32 hand-off through the registered workspace)
int mixq_wr_ksplit_config();                 // its configuration index
int mixq_wr_ksplit_ok(int M, int N, int KB); // MIXQ_OK when it can run this problem on the current device (workspace, residency), else the reason
bool mixq_wr_ksplit_pays(int M, int N, int KB);   // ... and the tile model says it is the faster choice
// the joint gate_proj / up_proj form (MIXQ_ACT_SILU_PAIR: interleaved weight rows, N / 2 output columns): the tiling for a problem, and
// whether a (forced) configuration has the paired epilogue
int mixq_wr_pick_pair(int bit, int M, int N, int KB);   // bit: 8, or 6 (int4 as FP6 codes)
bool mixq_wr_has_pair(int bit, int c);
