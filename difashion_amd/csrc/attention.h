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
  float* lse;                   // optional [B][H][Nq] fp32: log2-domain log-sum-exp (m + log2 l) of the scaled scores, for backward
  unsigned long long* prof;     // diagnosis only (DFH_ATTN_PROF=1): s_memtime stamps of workgroup 0 / wave 0, 8 per key tile
};

// Attention backward (training): P is recomputed from Q, K and the forward LSE; two passes of one kernel.
struct AttnBwdArgs {
  const bf16_t* Q; int ldq;      // [B][Nq][ldq]
  const bf16_t* K; int ldk;      // [B][Nk][ldk]
  const bf16_t* V; int ldv;      // [B][Nk][ldv]   ROW-major V (not V^T)
  const bf16_t* dO; int ldo;     // [B][Nq][ldo]
  const float* lse;              // [B][H][Nq]
  const float* delta;            // [B][H][Nq]  rowsum(dO * O)
  bf16_t* dQ; int lddq;          // [B][Nq][lddq]
  bf16_t* dK; int lddk;          // [B][Nk][lddk]
  bf16_t* dV; int lddv;          // [B][Nk][lddv]
  int B, H, D, Nq, Nk;
  float scale;
};

namespace dfh {
int attention_launch(const AttnArgs& a, hipStream_t stream);
// delta[b][h][q] = sum_d dO[b][q][h*D+d] * O[b][q][h*D+d]
int attention_delta_launch(const bf16_t* O, const bf16_t* dO, int ld, float* delta, int B, int H, int D, int Nq, hipStream_t stream);
int attention_bwd_launch(const AttnBwdArgs& a, hipStream_t stream);
}
