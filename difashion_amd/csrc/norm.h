#pragma once
#include "dfh_common.h"
#include <algorithm>

enum { GN_MAX_CHUNKS = 64 };

struct GnArgs {
  const bf16_t* src0; const bf16_t* src1;  // [B][HW][C0], [B][HW][C1] (src1 optional: fused channel concat)
  int C0, C1;
  int B, HW, G;
  const float* gamma; const float* beta;    // [C0+C1]
  float eps; int silu;
  bf16_t* out;                               // [B][HW][C0+C1]
  // fp8 proj_in (BASELINE configs[4]): when out8 is set the kernels write e4m3(clamp((x - mean) * rstd * q_mul, +-448)) [B][HW][C]
  // INSTEAD of the bf16 tensor -- the normalised value WITHOUT gamma / beta (the consumer's weights carry gamma, its bias W . beta:
  // unet_model.h), under a static scale: |z| <= 448 / q_mul is representable, typical |z| ~ 1 sits 4-5 binades inside the e4m3 range.
  // One source, no SiLU.
  uint8_t* out8; float q_mul;
  float* partial;                            // >= B * GN_MAX_CHUNKS * G * 2 floats
  float* stats_out;                          // optional [B][G][2] (mean, rstd), kept for the backward pass
  const float* pre; int pre_chunks;          // optional: statistics partials written by the producing GEMM's epilogue
                                             // ([B][G][pre_chunks][2], gemm.h GemmArgs::gstat): gn_stats_kernel is skipped
  // filled by the launcher
  int C, PL, chunks, pix_per_chunk, apix_per_chunk;
};

// GroupNorm(+SiLU) backward.  x = concat(src0, src1) as in the forward; dy [B][HW][C]; stats [B][G][2] from the forward.
struct GnBwdArgs {
  const bf16_t* src0; const bf16_t* src1; int C0, C1;
  const bf16_t* dy;
  int B, HW, G;
  const float* gamma; const float* beta; const float* stats; int silu;
  bf16_t* dx0; bf16_t* dx1; int acc0, acc1;  // gradient wrt each source; acc: add into existing contents
  float* dgamma; float* dbeta;               // fp32 [C], accumulated with atomics
  float* partial;                            // >= B * GN_MAX_CHUNKS * G * 2 floats
  int C, PL, chunks, pix_per_chunk, apix_per_chunk;
};

// GroupNorm FOLDED into the 1x1 projection that consumes it (transformer entry: norm -> proj_in, difashion.py:249-253 through diffusers
// Transformer2DModel): y[t][n] = sum_c W[n][c] ((x[t][c] - mean) rstd gamma_c + beta_c) + bias[n]
//                              = sum_c Wimg[i][n][c] x[t][c] + rv[i][n]         for the pixels t of image i,
// Wimg[i][n][c] = bf16(W[n][c] gamma_c rstd_{i, g(c)}),  rv[i][n] = bias[n] + sum_c (W[n][c] beta_c - Wimg[i][n][c] mean_{i, g(c)})
// (the mean term uses the ROUNDED per-image weight, the one the GEMM multiplies x by, so a constant channel cancels exactly).  The
// normalised tensor is never written: the GEMM reads x itself with per-image weights (GemmArgs::w_img_bs) and rv as its row vector.
struct GnFoldArgs {
  const bf16_t* x; int B, HW, C, G; float eps;
  const float* gamma; const float* beta;
  const float* pre; int pre_chunks;          // statistics partials of x from its producer ([B][G][pre_chunks][2]) or nullptr:
  float* partial;                            // ... then gn_stats_kernel sums x into `partial` (>= B * GN_MAX_CHUNKS * G * 2 floats)
  const bf16_t* W; int ldw; int N; const float* bias;   // the projection: bf16 [N][ldw] (K = C), optional fp32 bias
  bf16_t* Wimg;                              // out: [B][N][C]
  float* rv;                                 // out: [B][N]
};

namespace dfh {
int groupnorm_fold_launch(GnFoldArgs a, hipStream_t stream);
int groupnorm_bwd_launch(GnBwdArgs a, hipStream_t stream);
// LayerNorm backward over rows [M][C]; dx (=|+=); dgamma/dbeta fp32 atomics
int layernorm_bwd_launch(const bf16_t* x, const bf16_t* dy, const float* gamma, bf16_t* dx, int accumulate, float* dgamma,
                         float* dbeta, int M, int C, float eps, hipStream_t stream);
int groupnorm_launch(GnArgs a, hipStream_t stream);
int layernorm_launch(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int M, int C, float eps,
                     hipStream_t stream);
}  // namespace dfh
