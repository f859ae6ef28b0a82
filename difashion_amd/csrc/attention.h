#pragma once
#include "dfh_common.h"

struct AttnArgs {
  const bf16_t* Q; int ldq;     // [B][Nq][ldq]; head h occupies columns h*D .. h*D+D
  const bf16_t* K; int ldk;     // [B][Nk][ldk]
  const bf16_t* Vt; int ldvt;   // [B][H*D][ldvt]  V transposed (key index contiguous), ldvt >= roundup8(Nk)
  long vt_bstride;              // elements between batches of Vt (0 = H*D*ldvt): lets a layer read a slice of a batched V^T
  bf16_t* O; int ldo;           // [B][Nq][ldo]
  // fp8 to_out (BASELINE configs[4]): O8 != null -> the output leaves as e4m3 [B][Nq][ldo] (bytes) instead of bf16, scaled by
  // 448 / o_amax[b]: the attention output is a convex combination of the rows of V, so |O| <= max |V| of the batch element, which
  // the V projection's epilogue tracked (gemm.h Fp8GemmArgs::amax).  The consumer multiplies o_amax[b] / 448 back in.
  uint8_t* O8; const float* o_amax;
  // fp8 attention products (attention_fp8.hip; BASELINE configs[4]): static per-channel operand factors [H * D] -- Q is quantised as
  // q * f8_rq, K as k * f8_rk (f8_rq * f8_rk = 1 / f8_hs[h] for every channel of head h), V as v * f8_rv -- and the per-head factor the
  // softmax scale absorbs, all derived from the projection weights (attn_scales_launch).  Null: the bf16 kernels.
  const float* f8_rq; const float* f8_rk; const float* f8_rv; const float* f8_hs;
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

#ifdef __HIPCC__
// four consecutive channels d0 .. d0 + 3 of output row `row` (elements from the start of O), already normalised (and, for O8, scaled)
DFH_DEVICE void attn_store4(const AttnArgs& a, long row, int d0, float v0, float v1, float v2, float v3) {
  if (a.O8) {
    const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v0, -448.f, 448.f), __builtin_amdgcn_fmed3f(v1, -448.f, 448.f), 0, false);
    *(unsigned*)(a.O8 + row + d0) =
        (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v2, -448.f, 448.f), __builtin_amdgcn_fmed3f(v3, -448.f, 448.f), lo, true);
  } else {
    uint2 w;
    w.x = pack2bf(v0, v1); w.y = pack2bf(v2, v3);
    *(uint2*)(a.O + row + d0) = w;
  }
}
// multiplier that takes a normalised output of batch element b to its stored form
DFH_DEVICE float attn_qmul(const AttnArgs& a, int b) {
  if (!a.O8) return 1.0f;
  const float am = a.o_amax[b];
  return am > 0.f ? 448.0f / am : 0.f;
}
#endif

namespace dfh {
int attention_launch(const AttnArgs& a, hipStream_t stream);
// operand factors of the fp8 attention from the LayerNorm-folded projection weights wf [3C][C] (rows q | k | v) and biases bf [3C]
int attn_scales_launch(const bf16_t* wf, const float* bf, int C, int heads, float* rq, float* rk, float* rv, float* hs, hipStream_t stream);
// delta[b][h][q] = sum_d dO[b][q][h*D+d] * O[b][q][h*D+d]
int attention_delta_launch(const bf16_t* O, const bf16_t* dO, int ld, float* delta, int B, int H, int D, int Nq, hipStream_t stream);
int attention_bwd_launch(const AttnBwdArgs& a, hipStream_t stream);
}
