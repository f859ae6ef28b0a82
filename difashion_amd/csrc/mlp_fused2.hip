// The GEGLU feed-forward of a transformer block as ONE kernel (C = 320: the 64x64 level of the SD U-Nets):
//
//     out = proj_out( ff.net.2( GEGLU( ff.net.0( LayerNorm3(x) ) ) ) + x ) + resid
//
// Reference call site: DiFashion/models/difashion.py:518-523 -> diffusers BasicTransformerBlock.ff (GEGLU FeedForward) + Transformer2DModel.proj_out.
// As two launches (the folded-LayerNorm GEGLU projection, then [proj_out . ff2 | proj_out] over [hidden | h2]) the hidden tensor
// ([tokens][4 C] bf16: 168 MB at batch 16) makes a round trip through HBM and the short-K GEMMs around it run behind their epilogues
// (250-262 us per block).  Here a wave owns 16 tokens for the whole chain and the hidden units never leave its registers:
//
//   * X (raw rows; LayerNorm folded into the weights: lnfold.hip) sits in 40 VGPRs as the B operand of v_mfma_f32_16x16x32_bf16 for all ten
//     k-steps, read once from HBM;
//   * per chunk of 32 hidden units: D1[64 packed rows][16 tokens] = W1'[chunk] . X^T (40 MFMAs), the LayerNorm fix-up and the GEGLU gate on the
//     accumulators (the packed rows put value and gate of a hidden unit into the same lane), and the rounded product IS the B operand of the
//     second GEMM -- the C layout of two 16-row tiles is the B layout of a 32-deep k-step once the weight columns are permuted to match
//     (mlp2_pack_kernel) -- D2[320][16 tokens] += W2[:, chunk] . H (20 MFMAs);
//   * the [hidden | h2] K-segment trick of the unfused walk (AttL::fffp: ff.net.2 and proj_out as one matrix) carries over: after the last
//     chunk D2 += Wp . X^T (100 MFMAs on the same X registers), then bias + residual + store through the dead weight ring, and the
//     GroupNorm statistics of the output for the next resnet.
//
// Eight waves of one workgroup (two per SIMD) share the weight stream: every chunk's W1' / W2 slices are stored in HBM as the LDS image the
// fragment reads want (1-KB blocks, lane-linear), staged through registers one chunk ahead into a double-buffered ring, one barrier per
// chunk.  The shape of a wave is the result of a measurement: the first form of this kernel (scripts/probes/kernels/mlp_fused_v1.hip) gave a
// wave 32 tokens on v_mfma_f32_32x32x16_bf16, one wave per SIMD, and showed that ONE wave does not overlap its own MFMAs with its own
// VALU / LDS / VMEM issue: an iteration took the SUM of its 1920 matrix-pipe cycles and its ~3000 cycles of other instructions (257-265 us
// per launch, profiles/r05/mlp_fused_v1_microbench.txt).  With 16 tokens per wave -- X 40 + D2 80 + D1 2 x 16 registers of ~250, ALL of
// them VGPRs (a function that touches AGPRs gets its 256-register budget split 128 / 128) -- a 512-thread workgroup puts TWO waves on
// every SIMD and one's MFMAs run under the other's GEGLU slices, fragment reads and staging: 239-250 us.  The price: a 16-row weight
// fragment (1 KB from LDS) feeds one 16-cycle MFMA, so at full matrix-pipe rate the four SIMDs would ask the LDS for its whole 256 B/clk
// -- the kernel is LDS-read bound at ~480 KB per iteration, not MFMA bound.
//
// Layouts (v_mfma_f32_16x16x32_bf16; lane = (c = lane & 15, g = lane >> 4)): A row c, k = 8 g .. 8 g + 7 of the 32-deep step; B column c,
// same k; D column c, rows 4 g + r.  Packed W1' rows come as [16 values | 16 gates] per 16 hidden units, so a (value tile, gate tile)
// pair leaves value and gate of hidden units 4 g + r in the same lane; two such pairs = 32 hidden units = one k-step of the second
// GEMM, whose B operand (k-slot 8 g + q) is {units 4 g + q of pair 0, q < 4; units 16 + 4 g + (q - 4) of pair 1}: W2's columns are
#include "mlp_fused.h"

#include <cstdlib>
#include <cstring>

namespace {

#include "mlp_fused2_core.h"

// ---------------------------------------------------------------------------------------------------------------- weight image
__global__ __launch_bounds__(256) void mlp2_pack_kernel(const bf16_t* __restrict__ w1, const float* __restrict__ s1, const float* __restrict__ b1,
                                                        const bf16_t* __restrict__ w2p, unsigned char* __restrict__ img) {
  const long slot = (long)blockIdx.x * 256 + threadIdx.x;
  const long byte = slot * 16;
  if (byte >= IMG_BYTES) return;
  uint4 v = uint4{0u, 0u, 0u, 0u};
  if (byte < G3_OFF) {
    const int c = (int)(byte / CHUNK_BYTES), off = (int)(byte - (long)c * CHUNK_BYTES);
    if (off < W1_BYTES) {
      const int blk = off >> 10, lane = (off & 1023) >> 4;
      const int t = blk / KS, ks = blk - t * KS;               // tile t: packed rows 64 c + 16 t .. + 15 (v0, g0, v1, g1)
      const int row = c * 64 + t * 16 + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
      v = *(const uint4*)(w1 + (long)row * MC + k0);
    } else if (off < W1_BYTES + W2_BYTES) {
      const int o2 = off - W1_BYTES, ct = o2 >> 10, lane = (o2 & 1023) >> 4;
      const int n = 16 * ct + (lane & 15), g = lane >> 4;
      const bf16_t* src = w2p + (long)n * (5 * MC) + c * 32;
      const uint2 lo = *(const uint2*)(src + 4 * g), hi = *(const uint2*)(src + 16 + 4 * g);      // units 4 g .. + 3 | 16 + 4 g .. + 3
      v = uint4{lo.x, lo.y, hi.x, hi.y};
    } else {
      const int p = (off - W1_BYTES - W2_BYTES) >> 4;          // row pair: (s_r, s_r+1, b_r, b_r+1) of packed rows 64 c + 2 p, + 1
      if (p < 32) {
        const int r = c * 64 + 2 * p;
        v.x = __float_as_uint(s1[r]); v.y = __float_as_uint(s1[r + 1]); v.z = __float_as_uint(b1[r]); v.w = __float_as_uint(b1[r + 1]);
      }
    }
  } else {
    const int o3 = (int)(byte - G3_OFF), q = o3 / G3_BYTES, o4 = o3 - q * G3_BYTES;
    const int blk = o4 >> 10, lane = (o4 & 1023) >> 4;
    const int ct = blk >> 1, kk = blk & 1;
    const int n = 16 * ct + (lane & 15), k0 = 32 * (2 * q + kk) + 8 * (lane >> 4);
    v = *(const uint4*)(w2p + (long)n * (5 * MC) + 4 * MC + k0);
  }
  *(uint4*)(img + byte) = v;
}

__global__ __launch_bounds__(512, 2) void mlp2_fused_kernel(const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int m = blockIdx.x * 128 + wave * 16 + (lane & 15);          // this lane's token (M % 128 == 0)
  const unsigned char* img = a.img;

  Mlp2State st;
  // the seven GELU coefficients live in SGPR pairs (one scalar source per v_pk_fma_f32): eight VGPRs the two-wave budget does not have
  st.gk.k65 = f32x2_t{1.917432119e-05f, -6.586021110e-04f}; st.gk.k43 = f32x2_t{7.754402186e-03f, -5.296538429e-02f};
  st.gk.k21 = f32x2_t{-4.590602584e-01f, -1.151122051e+00f}; st.gk.k0 = f32x2_t{3.063254510e-07f - 1.0f, 0.0f};
  asm volatile("" : "+s"(st.gk.k65), "+s"(st.gk.k43), "+s"(st.gk.k21), "+s"(st.gk.k0));
#pragma unroll
  for (int k = 0; k < 5; ++k)                                        // W1 of chunk 0 -> W1 slot 0
    *(u32x4_t*)(smem + LDS_W1 + (wave + 8 * k) * 1024 + lane * 16) = *(gptr16_t)(img + (long)(wave + 8 * k) * 1024 + lane * 16);
  {
    const bf16_t* xr = a.x + (long)m * MC + 8 * g;                   // X fragments: token m, k = 32 ks + 8 g .. + 7
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) st.xf[ks] = *(const bf16x8_t*)(xr + 32 * ks);
  }
  {
    GemmArgs gg; gg.ln_stat = a.ln_stat; gg.ln_parts = a.ln_parts; gg.ln_cnt = a.ln_cnt; gg.ln_eps = a.ln_eps; gg.M = a.M;
    const float2 mr = ln_row_stats(gg, m);
    st.rstd = mr.y; st.ms = -mr.x * mr.y;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) st.d2[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  mlp2_iter<0, 0, false, 0>(st, smem, img, (long)CHUNK_BYTES, (long)W1_BYTES, wave, lane);
  for (int c = 1; c < NCHUNK - 1; c += 2) {
    mlp2_iter<0, 1, true, 0>(st, smem, img, (long)(c + 1) * CHUNK_BYTES, (long)c * CHUNK_BYTES + W1_BYTES, wave, lane);
    mlp2_iter<0, 0, true, 0>(st, smem, img, (long)(c + 2) * CHUNK_BYTES, (long)(c + 1) * CHUNK_BYTES + W1_BYTES, wave, lane);
  }
  mlp2_iter<0, 1, true, 0>(st, smem, img, G3_OFF, (long)(NCHUNK - 1) * CHUNK_BYTES + W1_BYTES, wave, lane);
  mlp2_iter<1, 0, true, 0>(st, smem, img, G3_OFF + 1 * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 1, false, 1>(st, smem, img, G3_OFF + 2 * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 0, false, 2>(st, smem, img, G3_OFF + 3 * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 1, false, 3>(st, smem, img, G3_OFF + 4 * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 0, false, 4>(st, smem, img, -1, -1, wave, lane);

  // ---- epilogue: out[m][n] = D2 + bias[n] + resid[m][n], n = 16 ct + 4 g + r.  The wait for the last MFMAs is tied to the accumulator
  //      tuples as operands (their AGPR -> VGPR copies are plain moves the register allocator would otherwise place right behind the
  //      last MFMA of each tuple), and the addresses are derived from an opaque copy of the thread id so that they are computed HERE.
  asm volatile("s_nop 15\n\ts_nop 7"
               : "+v"(st.d2[0]), "+v"(st.d2[1]), "+v"(st.d2[2]), "+v"(st.d2[3]), "+v"(st.d2[4]), "+v"(st.d2[5]), "+v"(st.d2[6]), "+v"(st.d2[7]),
                 "+v"(st.d2[8]), "+v"(st.d2[9]), "+v"(st.d2[10]), "+v"(st.d2[11]), "+v"(st.d2[12]), "+v"(st.d2[13]), "+v"(st.d2[14]), "+v"(st.d2[15]),
                 "+v"(st.d2[16]), "+v"(st.d2[17]), "+v"(st.d2[18]), "+v"(st.d2[19])
               :: "memory");
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int m2 = blockIdx.x * 128 + (tid2 >> 6) * 16 + (tid2 & 15), g2 = (tid2 >> 4) & 3;
  const long row = (long)m2 * MC;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int n = 16 * ct + 4 * g2;
    const float4 b4 = *(const float4*)(a.bias + n);
    const uint2 rr = *(const uint2*)(a.resid + row + n);
    const float v0 = st.d2[ct][0] + b4.x + __uint_as_float(rr.x << 16), v1 = st.d2[ct][1] + b4.y + __uint_as_float(rr.x & 0xffff0000u);
    const float v2 = st.d2[ct][2] + b4.z + __uint_as_float(rr.y << 16), v3 = st.d2[ct][3] + b4.w + __uint_as_float(rr.y & 0xffff0000u);
    uint2 o; o.x = pack2bf(v0, v1); o.y = pack2bf(v2, v3);
    *(uint2*)(a.out + row + n) = o;
    // the rounded tile also goes to LDS (the weight ring is dead: every wave passed the last iteration's barrier) for the statistics below
    if (a.gstat) *(uint2*)(smem + ((tid2 >> 6) * 16 + (tid2 & 15)) * (MC * 2) + n * 2) = o;
  }
  if (a.gstat) {
    // GroupNorm statistics of the output tile for the consuming GroupNorm (unet_model.h: the next resnet's norm1): thread (group j, token
    // slice sl) sums its cpg channels over 8 tokens of the staged tile, the 16 slices are folded in order -- fixed orders, no atomics
    __syncthreads();
    const int cpg = a.gstat_cpg, G = MC / cpg;
    float* red = (float*)(smem + 128 * MC * 2);                       // [16 slices][G][2] behind the tile
    const int j = tid2 % G, sl = tid2 / G;
    if (sl < 16) {
      float ss = 0.f, qq = 0.f;
      for (int t = sl * 8; t < sl * 8 + 8; ++t) {
        const bf16_t* src = (const bf16_t*)(smem + t * (MC * 2)) + j * cpg;
        for (int c = 0; c < cpg; ++c) { const float f = bf2f(src[c]); ss += f; qq = fmaf(f, f, qq); }
      }
      red[(sl * G + j) * 2] = ss; red[(sl * G + j) * 2 + 1] = qq;
    }
    __syncthreads();
    if (tid2 < G) {
      float ss = 0.f, qq = 0.f;
      for (int q = 0; q < 16; ++q) { ss += red[(q * G + tid2) * 2]; qq += red[(q * G + tid2) * 2 + 1]; }
      const int m0 = blockIdx.x * 128, b = m0 / a.gstat_hw, chunk = (m0 - b * a.gstat_hw) / 128, chunks = a.gstat_hw / 128;
      float* dst = a.gstat + (((long)b * G + tid2) * chunks + chunk) * 2;
      dst[0] = ss; dst[1] = qq;
    }
  }
}


}  // namespace

namespace dfh {

size_t mlp_fused_image_bytes() { return (size_t)IMG_BYTES; }
bool mlp_fused_eligible(int C, long M) { return C == MC && M > 0 && M % 128 == 0; }
// DFH_MLP_FUSED: 0 = the two-launch walk (A/B), 2 = this kernel (default); 1 = the first form (probe builds only, scripts/probes/kernels/mlp_fused_v1.hip)
int mlp_fused_form() {
  static const int form = [] { const char* e = getenv("DFH_MLP_FUSED"); return e ? atoi(e) : 2; }();
#ifndef DFH_PROBES
  return form == 1 ? 2 : form;
#else
  return form;
#endif
}


int mlp2_pack_launch(const bf16_t* w1, const float* s1, const float* b1, const bf16_t* w2p, void* img, hipStream_t stream) {
  DFH_REQUIRE(w1 && s1 && b1 && w2p && img, "null argument");
  static_assert(IMG_BYTES == 2703360, "both forms of the fused MLP share one image size (mlp_fused_image_bytes)");
  const long slots = IMG_BYTES / 16;
  hipLaunchKernelGGL(mlp2_pack_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, w1, s1, b1, w2p, (unsigned char*)img);
  return check_launch("mlp2_pack_kernel");
}

int mlp2_fused_launch(const MlpArgs& a, hipStream_t stream) {
  DFH_REQUIRE(a.x && a.resid && a.img && a.ln_stat && a.bias && a.out, "null argument");
  DFH_REQUIRE(a.M > 0 && a.M % 128 == 0, "fused MLP: whole 128-token tiles");
  DFH_REQUIRE(a.ln_parts > 0 && a.ln_cnt > 0 && a.ln_parts * a.ln_cnt == MC, "fused MLP: row statistics of a 320-channel producer");
  if (a.gstat) DFH_REQUIRE(a.gstat_cpg > 0 && MC % a.gstat_cpg == 0 && MC / a.gstat_cpg <= 32 && a.gstat_hw % 128 == 0 && a.M % a.gstat_hw == 0,
                           "fused MLP statistics: groups of channels dividing 320 (at most 32), whole 128-token chunks per image");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)mlp2_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    attr_set = true;
  }
  ProfScope ps(PC_LINEAR, 2.0 * a.M * (8.0 * MC * MC + 5.0 * MC * MC), 3.0 * a.M * MC * 2.0 + 13.0 * MC * MC * 2.0, stream);
  census(CK_MLP_FUSED);
  hipLaunchKernelGGL(mlp2_fused_kernel, dim3(a.M / 128), dim3(512), LDS_TOTAL, stream, a);
  return check_launch("mlp2_fused_kernel");
}



}  // namespace dfh
