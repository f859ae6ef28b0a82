// Fused GEGLU feed-forward + proj_out of a transformer block at C = 320 (mlp_fused2.hip; form 1, a probe kernel: scripts/probes/kernels/mlp_fused_v1.hip).
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

namespace dfh {
size_t mlp_fused_image_bytes();
bool mlp_fused_eligible(int C, long M);
// w1 / s1 / b1: the LayerNorm-folded GEGLU projection ([8 C][C] bf16, packed rows) and its fold vectors; w2p: [C][5 C] = [pout . ff2 | pout]
int mlp_pack_launch(const bf16_t* w1, const float* s1, const float* b1, const bf16_t* w2p, void* img, hipStream_t stream);
int mlp_fused_launch(const MlpArgs& a, hipStream_t stream);
// the kernel (mlp_fused2.hip): eight waves of 16 tokens, two per SIMD.  (mlp_pack_launch / mlp_fused_launch above: the first form, four waves
// of 32 tokens -- scripts/probes/kernels/mlp_fused_v1.hip, probe builds only; same arguments, its own image layout of the same size)
int mlp2_pack_launch(const bf16_t* w1, const float* s1, const float* b1, const bf16_t* w2p, void* img, hipStream_t stream);
int mlp2_fused_launch(const MlpArgs& a, hipStream_t stream);
// which path the walk uses: DFH_MLP_FUSED = 0 (the two-launch walk, A/B), 2 (the kernel, default), 1 (the first form: probe builds only)
int mlp_fused_form();
}  // namespace dfh
