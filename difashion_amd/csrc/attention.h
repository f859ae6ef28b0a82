#pragma once
#include "dfh_common.h"

struct AttnArgs {
  const bf16_t* Q; int ldq;     // [B][Nq][ldq]; head h occupies columns h*D .. h*D+D
  const bf16_t* K; int ldk;     // [B][Nk][ldk]
  const bf16_t* Vt; int ldvt;   // [B][H*D][ldvt]  V transposed (key index contiguous), ldvt >= roundup8(Nk)
  long vt_bstride;              // elements between batches of Vt (0 = H*D*ldvt): lets a layer read a slice of a batched V^T
  bf16_t* O; int ldo;           // [B][Nq][ldo]
  int B, H, D, Nq, Nk;
  float scale;                  // D^-0.5
};

namespace dfh {
int attention_launch(const AttnArgs& a, hipStream_t stream);
}
