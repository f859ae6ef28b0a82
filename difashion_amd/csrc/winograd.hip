// Winograd F(2x2, 3x3) for the stride-1 3x3 convs of the deep U-Net levels (ResnetBlock2D.conv1 / conv2 at the 16x16 and 8x8 levels of
// the SD-1.5 shape; reference call sites DiFashion/models/difashion.py:249-253,518-523 -> diffusers ResnetBlock2D).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 input patch d (stride 2) -> 2x2 outputs, summed over input channels
//
// so a conv over B x H x W pixels becomes SIXTEEN independent GEMMs [B H W / 4][Cin] x [Cin][Cout] -- 16 * Cin multiply-adds per four
// outputs instead of 36 * Cin (2.25 x fewer) -- which run as ONE batched launch of gemm_bf16_kernel (gemm.h GemmArgs::nbatch).  Why only
// the deep levels: there the direct implicit GEMM is short of rows (M = 4096 / 1024 pixels at batch 16: 256-tile launches, split-K and
// its reduce pass), while the transform-domain tensors (4 x the activation bytes each way) are small enough to stay in the Infinity Cache;
// at the 64x64 / 32x32 levels the transforms would move more bytes than the direct conv's whole launch takes.
//
//   wino_weight_kernel : packed bf16 W [N][ldw >= 9 C] (tap-major columns) -> U [16][N][C] bf16 = G g G^T in fp32, rounded once
//                        (per weight pack, into the fold region of the workspace); each plane optionally in 16 x 64 blocks (w_blocked)
//   wino_input_kernel  : g [B][H][W][C] bf16 (the GroupNorm + SiLU output the conv reads) -> V [16][B H W / 4][C] bf16 = B^T d B
//                        (zero padding = patches that hang over the border read zeros)
//   wino_output_kernel : M [16][B H W / 4][N] bf16 (the batched GEMM's output) -> out [B][H][W][N] bf16 = A^T m A + bias
//                        (+ time-embedding row of the image) (+ residual)
//
// Numerics: U, V and M are rounded to bf16 (fp32 arithmetic inside every kernel and in the MFMA accumulators); measured against
// the fp64 conv of the same bf16 operands the relative L2 error of one conv is 4.9e-3, against 1.7e-3 for the direct kernel (whose
// only error is the bf16 rounding of its output) -- tests/test_gpu_ops.py states the bound.
#include "gemm.h"

namespace {

DFH_DEVICE void load8(const bf16_t* p, float* f) { unpack8(*(const uint4*)p, f); }

__global__ __launch_bounds__(256) void wino_weight_kernel(const bf16_t* __restrict__ W, int ldw, bf16_t* __restrict__ U, int N, int C, int blocked) {
  const int n = blockIdx.x;
  const bf16_t* w = W + (long)n * ldw;
  for (int c = threadIdx.x; c < C; c += 256) {
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = bf2f(w[t * C + c]);
    float t4[4][3];                       // G g
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      t4[0][k] = g[0][k];
      t4[1][k] = 0.5f * (g[0][k] + g[1][k] + g[2][k]);
      t4[2][k] = 0.5f * (g[0][k] - g[1][k] + g[2][k]);
      t4[3][k] = g[2][k];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {         // (G g) G^T
      const float u[4] = {t4[i][0], 0.5f * (t4[i][0] + t4[i][1] + t4[i][2]), 0.5f * (t4[i][0] - t4[i][1] + t4[i][2]), t4[i][2]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // blocked: each plane as [N / 16][C / 64][16][64] (gemm.h GemmArgs::w_blocked)
        const long off = blocked ? ((long)(n >> 4) * (C >> 6) + (c >> 6)) * 1024 + (n & 15) * 64 + (c & 63) : (long)n * C + c;
        U[(long)(i * 4 + j) * N * C + off] = f2bf(u[j]);
      }
    }
  }
}

// one thread = one 4x4 patch x 8 channels
__global__ __launch_bounds__(256) void wino_input_kernel(const bf16_t* __restrict__ g, bf16_t* __restrict__ V, int B, int H, int W, int C) {
  const int CH = C >> 3, TH = H >> 1, TW = W >> 1;
  const long Mt = (long)B * TH * TW;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Mt * CH) return;
  const int ch = (int)(idx % CH);
  const long t = idx / CH;
  const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((long)TW * TH));
  float d[4][4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = 2 * ty - 1 + r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int x = 2 * tx - 1 + s;
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) load8(g + (((long)b * H + y) * W + x) * C + ch * 8, d[r][s]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[r][s][e] = 0.f;
      }
    }
  }
  // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
  float q[4][4][8];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      q[0][s][e] = d[0][s][e] - d[2][s][e];
      q[1][s][e] = d[1][s][e] + d[2][s][e];
      q[2][s][e] = d[2][s][e] - d[1][s][e];
      q[3][s][e] = d[1][s][e] - d[3][s][e];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[0][e] = q[i][0][e] - q[i][2][e];
      v[1][e] = q[i][1][e] + q[i][2][e];
      v[2][e] = q[i][2][e] - q[i][1][e];
      v[3][e] = q[i][1][e] - q[i][3][e];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *(uint4*)(V + ((long)(i * 4 + j) * Mt + t) * C + ch * 8) = pack8(v[j]);
  }
}

// one thread = one 2x2 output tile x 8 channels
__global__ __launch_bounds__(256) void wino_output_kernel(const bf16_t* __restrict__ Mb, bf16_t* __restrict__ out, const float* __restrict__ bias,
                                                          const float* __restrict__ rowvec, int rv_ld, int rv_off,
                                                          const bf16_t* __restrict__ resid, int B, int H, int W, int N) {
  const int CH = N >> 3, TH = H >> 1, TW = W >> 1;
  const long Mt = (long)B * TH * TW;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Mt * CH) return;
  const int ch = (int)(idx % CH);
  const long t = idx / CH;
  const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((long)TW * TH));
  // A^T m: rows (m0 + m1 + m2, m1 - m2 - m3)
  float s[2][4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float m[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) load8(Mb + ((long)(i * 4 + j) * Mt + t) * N + ch * 8, m[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[0][j][e] = m[0][e] + m[1][e] + m[2][e];
      s[1][j][e] = m[1][e] - m[2][e] - m[3][e];
    }
  }
  float add[8];
  {
    const float4 b0 = *(const float4*)(bias + ch * 8), b1 = *(const float4*)(bias + ch * 8 + 4);
    add[0] = b0.x; add[1] = b0.y; add[2] = b0.z; add[3] = b0.w; add[4] = b1.x; add[5] = b1.y; add[6] = b1.z; add[7] = b1.w;
    if (rowvec) {
      const float* rv = rowvec + (long)b * rv_ld + rv_off + ch * 8;
      const float4 r0 = *(const float4*)rv, r1 = *(const float4*)(rv + 4);
      add[0] += r0.x; add[1] += r0.y; add[2] += r0.z; add[3] += r0.w; add[4] += r1.x; add[5] += r1.y; add[6] += r1.z; add[7] += r1.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        y[e] = (j == 0 ? s[i][0][e] + s[i][1][e] + s[i][2][e] : s[i][1][e] - s[i][2][e] - s[i][3][e]) + add[e];
      const long pix = (((long)b * H + 2 * ty + i) * W + 2 * tx + j) * N + ch * 8;
      if (resid) {
        float r[8];
        load8(resid + pix, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] += r[e];
      }
      *(uint4*)(out + pix) = pack8(y);
    }
}

}  // namespace

namespace dfh {

int wino_weight_launch(const bf16_t* W, int ldw, bf16_t* U, int N, int C, int blocked, hipStream_t stream) {
  DFH_REQUIRE(W && U && N > 0 && C > 0 && ldw >= 9 * C, "bad argument");
  DFH_REQUIRE(!blocked || (N % 16 == 0 && C % 64 == 0), "blocked U needs N % 16 == 0 and C % 64 == 0");
  hipLaunchKernelGGL(wino_weight_kernel, dim3(N), dim3(256), 0, stream, W, ldw, U, N, C, blocked);
  return check_launch("wino_weight_kernel");
}

int wino_input_launch(const bf16_t* g, bf16_t* V, int B, int H, int W, int C, hipStream_t stream) {
  DFH_REQUIRE(g && V && B > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0 && C > 0 && C % 8 == 0, "even image sides, channels a multiple of 8");
  const long n = (long)B * (H / 2) * (W / 2) * (C / 8);
  ProfScope ps(PC_CONV3, 0.0, (double)B * H * W * C * 2.0 * 5.0, stream);
  hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, V, B, H, W, C);
  return check_launch("wino_input_kernel");
}

int wino_output_launch(const bf16_t* Mb, bf16_t* out, const float* bias, const float* rowvec, int rv_ld, int rv_off, const bf16_t* resid,
                       int B, int H, int W, int N, hipStream_t stream) {
  DFH_REQUIRE(Mb && out && bias && B > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0 && N > 0 && N % 8 == 0, "even image sides, channels a multiple of 8");
  const long n = (long)B * (H / 2) * (W / 2) * (N / 8);
  ProfScope ps(PC_CONV3, 0.0, (double)B * H * W * N * 2.0 * (resid ? 6.0 : 5.0), stream);
  hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, Mb, out, bias, rowvec, rv_ld, rv_off, resid,
                     B, H, W, N);
  return check_launch("wino_output_kernel");
}

}  // namespace dfh
