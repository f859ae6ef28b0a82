// Fused GEGLU feed-forward + proj_out of a transformer block at C = 320 (mlp_fused.hip).
#pragma once
#include "gemm.h"

struct MlpArgs {
  const bf16_t* x;              // [M][320] raw rows of the feed-forward's input (LayerNorm 3 is folded into the weights)
  const bf16_t* resid;          // [M][320] the transformer block's input: residual of proj_out
  const unsigned char* img;     // weight image of the layer (mlp_pack_launch)
  const float* ln_stat;         // per-row statistics of x from its producer: [ln_parts][M][2] (gemm.h GemmArgs::rowstat)
  int ln_parts, ln_cnt; float ln_eps;
  const float* bias;            // [320] pout . ff2b + poutb
  bf16_t* out;                  // [M][320]
  int M;
};

namespace dfh {
size_t mlp_fused_image_bytes();
bool mlp_fused_eligible(int C, long M);
// w1 / s1 / b1: the LayerNorm-folded GEGLU projection ([8 C][C] bf16, packed rows) and its fold vectors; w2p: [C][5 C] = [pout . ff2 | pout]
int mlp_pack_launch(const bf16_t* w1, const float* s1, const float* b1, const bf16_t* w2p, void* img, hipStream_t stream);
int mlp_fused_launch(const MlpArgs& a, hipStream_t stream);
}  // namespace dfh
