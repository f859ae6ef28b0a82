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
  float* partial;                            // >= B * GN_MAX_CHUNKS * G * 2 floats
  // filled by the launcher
  int C, PL, chunks, pix_per_chunk, apix_per_chunk;
};

namespace dfh {
int groupnorm_launch(GnArgs a, hipStream_t stream);
int layernorm_launch(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int M, int C, float eps,
                     hipStream_t stream);
}  // namespace dfh
