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
  // optional (form 2): GroupNorm statistics of `out` for its consumer -- per (image, group, 128-token chunk) the sum and the sum of squares of
  // the bf16-rounded outputs, layout [image][group][gstat_hw / 128][2] (what gn_apply_kernel reads as producer statistics, norm.h GnArgs::pre)
  float* gstat; int gstat_cpg, gstat_hw;
};

// K = N = 320 token linear on the fused-MLP machinery (mlp_fused2.hip token_linear_kernel)
struct TokLinArgs {
  const bf16_t* x;              // [M][320] bf16 rows
  const unsigned char* img;     // weight image (token_linear_pack_launch)
  const float* bias;            // [320] or null (b' for a folded-LayerNorm consumer)
  const bf16_t* resid;          // optional [M][320]
  const float* ln_stat; int ln_parts, ln_cnt; float ln_eps; const float* ln_s;      // folded-LayerNorm consumer (gemm.h), optional
  float* rowstat;               // optional: [M][2] = (mean, centred sum of squares) of each ROUNDED output row over its 320 columns
  bf16_t* out;                  // [M][320]
  int M;
};

namespace dfh {
size_t token_linear_image_bytes();
bool token_linear_eligible(int N, int K, long M);
int token_linear_pack_launch(const bf16_t* W, int ldw, void* img, hipStream_t stream);
int token_linear_launch(const TokLinArgs& a, hipStream_t stream);
size_t mlp_fused_image_bytes();
bool mlp_fused_eligible(int C, long M);
// w1 / s1 / b1: the LayerNorm-folded GEGLU projection ([8 C][C] bf16, packed rows) and its fold vectors; w2p: [C][5 C] = [pout . ff2 | pout]
int mlp_pack_launch(const bf16_t* w1, const float* s1, const float* b1, const bf16_t* w2p, void* img, hipStream_t stream);
int mlp_fused_launch(const MlpArgs& a, hipStream_t stream);
// second form (mlp_fused2.hip): eight waves of 16 tokens, two per SIMD; same arguments, its own image layout (same size)
int mlp2_pack_launch(const bf16_t* w1, const float* s1, const float* b1, const bf16_t* w2p, void* img, hipStream_t stream);
int mlp2_fused_launch(const MlpArgs& a, hipStream_t stream);
// which form the walk uses: DFH_MLP_FUSED = 0 (the two-launch walk), 1 (32-token waves), 2 (16-token waves, default)
int mlp_fused_form();
}  // namespace dfh
