#pragma once
#include "dfh_common.h"
#include <algorithm>
#include <cmath>

namespace dfh {
int pack_matrix_t_launch(const float* w, bf16_t* out, int N, int K, int ldt, int t_row_off, int t_col_off, int geglu, hipStream_t s);
int pack_conv3x3_t_launch(const float* w, bf16_t* out, int Cout, int Cin, int ldt, int t_col_off, int o_pad, hipStream_t s);
int unpack_matrix_launch(const float* g, float* grad, int N, int K, int ldw, int row_off, int col_off, int geglu, hipStream_t s);
int unpack_conv3x3_launch(const float* g, float* grad, int Cout, int Cin, int ldw, int col_off, int cin_pad, hipStream_t s);
int unpack_vector_launch(const float* g, float* grad, int N, int off, int geglu, hipStream_t s);
int pool2x2_sum_launch(const bf16_t* in, bf16_t* out, int B, int H, int W, int C, hipStream_t s);
// backward of an upsample conv through its four phase planes (gemm.h GemmArgs::phase2x): dY [B][2H][2W][C] -> [4][B][H][W][C];
// dx (=|+=) sum of four planes; dW3 [N][9 C] (= | +=) from dWp [4][N][4 C]
int phase_gather_launch(const bf16_t* in, bf16_t* out, int B, int H, int W, int C, hipStream_t s);
int phase_sum4_launch(const bf16_t* planes, bf16_t* dx, long n, int accumulate, hipStream_t s);
int ups_phase_unfold_launch(const float* dwp, float* dw3, int N, int C, int ldw, int overwrite, hipStream_t s);
int add_bf16_launch(bf16_t* dst, const bf16_t* src, long n, int accumulate, hipStream_t s);
int geglu_bwd_launch(const bf16_t* pre, const bf16_t* dy, bf16_t* dpre, long M, int N2, hipStream_t s);
int geglu_fwd_launch(const bf16_t* pre, bf16_t* y, long M, int N2, hipStream_t s);
int act_fwd_launch(const bf16_t* pre, bf16_t* y, long n, int kind, hipStream_t s);
int act_bwd_launch(const bf16_t* ref, const float* ref_f32, const bf16_t* dy, const float* dy_f32, bf16_t* dpre, long n, int kind,
                   float scale, hipStream_t s);
int nhwc_to_nchw_f32_launch(const bf16_t* src, float* dst, int B, int HW, int Cp, int C, float scale, int accumulate, hipStream_t s);
int transpose_bf16_launch(const bf16_t* in, bf16_t* out, int B, int R, int C, int ld_in, int ld_out, long in_bstride, long out_bstride,
                          hipStream_t s);
int mse_bwd_launch(const float* pred, const float* target, const float* w, float* dpred, int rows, int L, float loss_scale,
                   const float* scale_dev, hipStream_t s);
int assemble_bwd_launch(const float* dx, const unsigned char* mutual_real, float* dmutual, int rows, int CL, float eta, hipStream_t s);
int sumsq_launch(const float* g, long n, float* out, hipStream_t s);
int adamw_launch(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
                 int step, const float* sumsq, float max_norm, hipStream_t s, float* shadow = nullptr, float ema_decay = 0.f);
// bf16 wire format of the gradient exchange (dist.py exchange_bf16)
int wire_pack_launch(const float* g, bf16_t* wire, long n, long n_pad, hipStream_t s);
int wire_shard_mean_launch(const bf16_t* recv, bf16_t* shard, int world, long per, hipStream_t s);
int wire_unpack_launch(const bf16_t* wire, float* g, long n, hipStream_t s);
int ema_launch(float* shadow, const float* p, long n, float decay, hipStream_t s);
}  // namespace dfh
