// PROBE header: K = N = 320 token linear on the fused-MLP machinery (scripts/probes/kernels/token_linear.hip)
#pragma once
#include "gemm.h"

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
int token_linear_from_gemm(const GemmArgs& g, hipStream_t stream);      // dfh_gemm tile id 30
}  // namespace dfh
