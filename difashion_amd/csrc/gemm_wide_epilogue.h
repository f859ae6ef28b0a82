// The epilogue of the 256-row GEMM tiles (gemm_wide.hip, gemm_halo.hip): four 64-row passes through an fp32 LDS tile, so that bias /
// time-embedding row / activation / GEGLU / residual are applied on full rows (16-byte coalesced residual reads and output writes,
// one rounding).  4 or 8 waves as 2 x WN, each holding (BM/2) x (BN/WN) of accumulators in the 16x16 MFMA layout
// (lane (fr = lane & 15, fg = lane >> 4): pixel row i*16 + fr, channels j*16 + fg*4 .. +3).  LDS_BYTES = what the caller allocated
// (the pass tile and the bias slices reuse the pipeline buffers; the caller's last reads of them must be complete: the first
// statement of every pass is a barrier).
#pragma once
#include "gemm.h"

namespace {

template <int BM, int BN, int WN, int LDS_BYTES>
DFH_DEVICE void wide_epilogue(const GemmArgs& a, f32x4_t (&acc)[BM / 2 / 16][BN / WN / 16], unsigned char* smem, const int tid,
                              const int wm, const int wn, const int fr, const int fg, const int m0, const int n0) {
  constexpr int NWV = 2 * WN;
  constexpr int TM = BM / 2, TN = BN / WN;
  constexpr int FN = TN / 16;
  // ---------------------------------------------------------------- epilogue: four 64-row passes through fp32 LDS
  constexpr int RSF = BN * 4 + 16;                 // fp32 row stride (bytes); 64 rows = 42 KB
  constexpr int CPR = BN / 8;                      // 8-column chunks per row
  constexpr int EPI = (64 * CPR) / (NWV * 64);     // chunks per thread and pass
  static_assert(64 * RSF <= LDS_BYTES, "epilogue tile must fit the pipeline buffers");
  static_assert((64 * CPR) % (NWV * 64) == 0, "whole chunks per thread");
  // Residual rows are fetched one pass AHEAD (the loads of pass q+1 fly while pass q is rounded and stored): read inside the
  // pass they cost EPI dependent global-load latencies per pass -- 14 of the 37 us of a 65536 x 320 x 320 linear.
  // ONE register set: chunk e of pass q + 1 is requested at the top of chunk e of pass q, right after that chunk's value has been copied
  // out -- a second set (all of pass q + 1 requested before pass q is processed) cost 20 more VGPRs next to the 160 accumulator registers
  // of the 256 x 320 tile and spilled 6-26 of them.
  uint4 rnext[EPI];
  auto fetch_one = [&](int q, int e) {
    const int c = tid + e * NWV * 64;
    const int row = c / CPR, cchunk = c - row * CPR;
    const int m = m0 + q * 64 + row, n = n0 + cchunk * 8;
    rnext[e] = make_uint4(0u, 0u, 0u, 0u);
    if (m < a.M && n < a.N) rnext[e] = *(const uint4*)(a.resid + (long)m * a.ld_res + n);
  };
  auto fetch_resid = [&](int q) {
#pragma unroll
    for (int e = 0; e < EPI; ++e) fetch_one(q, e);
  };
  const bool has_resid = a.resid != nullptr && a.act != ACT_GEGLU;
  if (has_resid) fetch_resid(0);
  // GroupNorm statistics of the output (a.gstat, see gemm.h): the rounded values of a pass go back into the fp32 pass tile, thread
  // t < BN sums column t over the 64 rows in row order, and after the last pass the columns are folded into groups: fixed orders
  // throughout, no atomics.  The launcher guarantees full tiles inside one image and BN % cpg == 0.
  const bool gst = a.gstat != nullptr;
  const bool rst = a.rowstat != nullptr;             // per-row statistics of the output for a folded LayerNorm (gemm.h): full column tiles only
  constexpr int CQ = BN / 4;                        // column quads; thread t < 4 * CQ sums quad t % CQ over rows (t / CQ) * 16 .. + 15
  static_assert(4 * CQ <= NWV * 64, "one thread per (column quad, row quarter)");
  float4 col_s = float4{0.f, 0.f, 0.f, 0.f}, col_q = col_s;
  // the tile's bias slice goes to LDS once (behind the fp32 pass tile): per-chunk global reads cost 2.5 us per launch
  constexpr int BIAS_OFF = 64 * RSF;
  // ... and so does the time-embedding row (rowvec) when the whole tile lies inside one image (every level but 8x8): read per
  // chunk it was EPI dependent L2 round trips in each of the four passes of every resnet conv1
  constexpr int RV_OFF = BIAS_OFF + BN * 4;
  static_assert(RV_OFF + BN * 4 <= LDS_BYTES, "bias + rowvec slices must fit behind the pass tile");
  const bool rv_lds = a.rowvec != nullptr && (a.rv_ld == 0 || (m0 / a.rows_per_b) == ((min(m0 + BM, a.M) - 1) / a.rows_per_b));   // rv_ld == 0: one row for every image (cached timestep row)
  float4 bias_reg = make_float4(0.f, 0.f, 0.f, 0.f), rv_reg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias && tid < BN / 4 && n0 + tid * 4 < a.N) bias_reg = *(const float4*)(a.bias + n0 + tid * 4);
  if (rv_lds && tid < BN / 4 && n0 + tid * 4 < a.N)
    rv_reg = *(const float4*)(a.rowvec + (long)(m0 / a.rows_per_b) * a.rv_ld + a.rv_off + n0 + tid * 4);
#pragma unroll
  for (int q = 0; q < BM / 64; ++q) {
    __syncthreads();                               // pipeline buffers / previous pass no longer read
    if (wm == (q * 64) / TM) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = ((q * 64) % TM) / 16 + ii;
        const int row = ii * 16 + fr;              // row inside the 64-row pass
#pragma unroll
        for (int j = 0; j < FN; ++j)
          *(f32x4_t*)(smem + row * RSF + (wn * TN + j * 16 + fg * 4) * 4) = acc[i][j];
      }
    }
    if (q == 0 && tid < BN / 4) {
      *(float4*)(smem + BIAS_OFF + tid * 16) = bias_reg;
      *(float4*)(smem + RV_OFF + tid * 16) = rv_reg;
    }
    __syncthreads();
    if (a.act == ACT_GEGLU) {
      // packed columns come in 32-blocks (16 values | 16 gates): out[m][16 k + i] = (v_i + bv_i) * gelu(g_i + bg_i)
      constexpr int GPR = BN / 16;                 // 8-output chunks per row: two per 32-block
      for (int c = tid; c < 64 * GPR; c += NWV * 64) {
        const int row = c / GPR, gc = c - row * GPR;
        const int colv = (gc >> 1) * 32 + (gc & 1) * 8;        // value columns of this chunk; gates at +16
        const int m = m0 + q * 64 + row, n = n0 + colv;
        if (m >= a.M || n + 16 >= a.N) continue;
        float v[8], g[8];
        {
          const float* src = (const float*)(smem + row * RSF) + colv;
          const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
          const float4 g0 = *(const float4*)(src + 16), g1 = *(const float4*)(src + 20);
          v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
          g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        }
        if (a.bias) {
          const float* bl = (const float*)(smem + BIAS_OFF) + colv;
          const float4 bv0 = *(const float4*)bl, bv1 = *(const float4*)(bl + 4);
          const float4 bg0 = *(const float4*)(bl + 16), bg1 = *(const float4*)(bl + 20);
          v[0] += bv0.x; v[1] += bv0.y; v[2] += bv0.z; v[3] += bv0.w; v[4] += bv1.x; v[5] += bv1.y; v[6] += bv1.z; v[7] += bv1.w;
          g[0] += bg0.x; g[1] += bg0.y; g[2] += bg0.z; g[3] += bg0.w; g[4] += bg1.x; g[5] += bg1.y; g[6] += bg1.z; g[7] += bg1.w;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= gelu_erf_f(g[r]);
        *(uint4*)((bf16_t*)a.out + (long)m * a.ld_out + ((n0 + (gc >> 1) * 32) >> 1) + (gc & 1) * 8) = pack8(v);
      }
      continue;
    }
#pragma unroll
    for (int e = 0; e < EPI; ++e) {
      const int c = tid + e * NWV * 64;
      const int row = c / CPR, cchunk = c - row * CPR;
      const int m = m0 + q * 64 + row, n = n0 + cchunk * 8;
      const uint4 rcur = rnext[e];
      if (has_resid && q + 1 < BM / 64) fetch_one(q + 1, e);
      if (m >= a.M || n >= a.N) continue;
      float v[8];
      {
        const float4 lo = *(const float4*)(smem + row * RSF + cchunk * 32);
        const float4 hi = *(const float4*)(smem + row * RSF + cchunk * 32 + 16);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
      }
      if (a.bias) {
        const float4 b0 = *(const float4*)(smem + BIAS_OFF + cchunk * 32), b1 = *(const float4*)(smem + BIAS_OFF + cchunk * 32 + 16);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (a.rowvec) {
        float4 r0, r1;
        if (rv_lds) {
          r0 = *(const float4*)(smem + RV_OFF + cchunk * 32); r1 = *(const float4*)(smem + RV_OFF + cchunk * 32 + 16);
        } else {
          const float* rv = a.rowvec + (long)(m / a.rows_per_b) * a.rv_ld + a.rv_off + n;
          r0 = *(const float4*)rv; r1 = *(const float4*)(rv + 4);
        }
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      }
      if (a.act == ACT_SILU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = silu_f(v[r]);
      } else if (a.act == ACT_LEAKY) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r];
      } else if (a.act == ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = tanhf(v[r]);
      }
      if (has_resid) {
        float f[8];
        unpack8(rcur, f);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += f[r];
      }
      const uint4 packed = pack8(v);
      long orow = m;
      if (a.phase2x == 1) {                            // phase plane of an upsample conv (gemm.h): source pixel -> its pixel of the 2H x 2W image
        const int hw = a.Hout * a.Wout, pb = m / hw, rem = m - pb * hw;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        orow = (long)pb * 4 * hw + (long)(2 * oy + (int)(blockIdx.y >> 1)) * (2 * a.Wout) + 2 * ox + (int)(blockIdx.y & 1);
      }
      // batched launch (gemm.h GemmArgs::nbatch): plane blockIdx.y of the output (o_bs = 0 otherwise)
      *(uint4*)((bf16_t*)a.out + (long)blockIdx.y * a.o_bs + orow * a.ld_out + n) = packed;
      if (gst || rst) {                                // this thread read the slot, nobody else touches it in this pass
        float f[8];
        unpack8(packed, f);
        *(float4*)(smem + row * RSF + cchunk * 32) = float4{f[0], f[1], f[2], f[3]};
        *(float4*)(smem + row * RSF + cchunk * 32 + 16) = float4{f[4], f[5], f[6], f[7]};
      }
    }
    if (rst) {
      // four adjacent lanes per row of the 64-row pass, two passes over the rounded values (mean, then centred squares), fixed orders
      static_assert(NWV * 64 >= 256, "four threads per row of a 64-row pass");
      __syncthreads();
      if (tid < 256) {
        const int row = tid >> 2, part = tid & 3;
        const unsigned char* src = smem + row * RSF;
        float sum = 0.f;
        for (int c = part; c < BN / 4; c += 4) { const float4 x = *(const float4*)(src + c * 16); sum += (x.x + x.y) + (x.z + x.w); }
        sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
        const float mean = sum * (1.0f / BN);
        float m2 = 0.f;
        for (int c = part; c < BN / 4; c += 4) {
          const float4 x = *(const float4*)(src + c * 16);
          const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
          m2 = fmaf(d0, d0, m2); m2 = fmaf(d1, d1, m2); m2 = fmaf(d2, d2, m2); m2 = fmaf(d3, d3, m2);
        }
        m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64);
        const int m = m0 + q * 64 + row;
        if (part == 0 && m < a.M) *(float2*)(a.rowstat + ((long)(n0 / BN) * a.M + m) * 2) = float2{mean, m2};
      }
    }
    if (gst) {
      __syncthreads();
      if (tid < 4 * CQ) {
        const int cq = tid % CQ, r0 = (tid / CQ) * 16;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
          const float4 x = *(const float4*)(smem + (r0 + r) * RSF + cq * 16);
          col_s.x += x.x; col_s.y += x.y; col_s.z += x.z; col_s.w += x.w;
          col_q.x += x.x * x.x; col_q.y += x.y * x.y; col_q.z += x.z * x.z; col_q.w += x.w * x.w;
        }
      }
    }
  }
  if (gst) {
    __syncthreads();
    float* cst = (float*)smem;                         // [BN][2], then the four row quarters folded in order
    float* qrt = cst + 2 * BN;                         // [4][BN][2]
    if (tid < 4 * CQ) {
      const int cq = tid % CQ, rq = tid / CQ;
      float* d = qrt + (rq * BN + cq * 4) * 2;
      d[0] = col_s.x; d[1] = col_q.x; d[2] = col_s.y; d[3] = col_q.y; d[4] = col_s.z; d[5] = col_q.z; d[6] = col_s.w; d[7] = col_q.w;
    }
    __syncthreads();
    if (tid < BN) {
      float ss = 0.f, qq = 0.f;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) { ss += qrt[(rq * BN + tid) * 2]; qq += qrt[(rq * BN + tid) * 2 + 1]; }
      cst[tid * 2] = ss; cst[tid * 2 + 1] = qq;
    }
    __syncthreads();
    const int cpg = a.gstat_cpg;
    if (tid < BN / cpg) {
      float ss = 0.f, qq = 0.f;
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { ss += cst[c * 2]; qq += cst[c * 2 + 1]; }
      const int b = m0 / a.gstat_hw, chunk = (m0 - b * a.gstat_hw) / BM, chunks = a.gstat_hw / BM;
      const int g = (n0 + tid * cpg) / cpg, G = a.N / cpg;
      float* dst = a.gstat + (((long)b * G + g) * chunks + chunk) * 2;
      dst[0] = ss; dst[1] = qq;
    }
  }
}

}  // namespace
