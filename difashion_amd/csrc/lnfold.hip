// LayerNorm folded into the projection that consumes it (inference walk of the transformer blocks; reference call site
// DiFashion/models/difashion.py:518-523 -> diffusers BasicTransformerBlock: x + attn1(LN1(x)), x + attn2(LN2(x), ctx), x + ff(LN3(x))).
//
//   LN(x)[k] = (x[k] - mean) * rstd * gamma[k] + beta[k]
//   LN(x) . W[n]  =  rstd * ( x . W'[n]  -  mean * s[n] )  +  b'[n],     W'[n][k] = W[n][k] * gamma[k],
//                                                                        s[n]  = sum_k W'[n][k]      (of the bf16-ROUNDED W': the
//                                                                                 cancellation against x . W' must be exact),
//                                                                        b'[n] = bias[n] + sum_k W[n][k] * beta[k]
// so the GEMM runs on the raw rows x (no LayerNorm kernel, no normalised copy of the tensor in HBM) and its epilogue applies the two
// per-row scalars (gemm.h GemmArgs::ln_stat).  This kernel derives W', s and b' from the packed bf16 matrix whenever the weights are
// packed: one block per output row.  Works on any packed row order (GEGLU-interleaved rows carry their bias in the same order).
#include "gemm.h"

namespace {

__global__ __launch_bounds__(256) void ln_fold_kernel(const bf16_t* __restrict__ W, int ldw, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      bf16_t* __restrict__ WF, float* __restrict__ s_out, float* __restrict__ b_out, int K) {
  const int n = blockIdx.x;
  const bf16_t* w = W + (long)n * ldw;
  bf16_t* wf = WF + (long)n * K;
  float s = 0.f, b = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8) {          // K is a multiple of 8 (16-byte rows)
    float f[8], o[8];
    unpack8(*(const uint4*)(w + k), f);
    const float4 g0 = *(const float4*)(gamma + k), g1 = *(const float4*)(gamma + k + 4);
    const float4 e0 = *(const float4*)(beta + k), e1 = *(const float4*)(beta + k + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
    for (int r = 0; r < 8; ++r) { o[r] = f[r] * g[r]; b = fmaf(f[r], e[r], b); }
    const uint4 packed = pack8(o);
    *(uint4*)(wf + k) = packed;
    unpack8(packed, o);                                           // the rounded values are what the MFMA multiplies
#pragma unroll
    for (int r = 0; r < 8; ++r) s += o[r];
  }
  __shared__ float red[2][4];
  s = wave_sum(s); b = wave_sum(b);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = s; red[1][wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s_out[n] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    b_out[n] = (bias ? bias[n] : 0.f) + ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

// b_out[n] = b_add[n] + sum_j W[n][j] * v[j]   (bias of two chained linears folded into one: W = the second layer's weights)
__global__ __launch_bounds__(256) void matvec_bias_kernel(const bf16_t* __restrict__ W, int ldw, const float* __restrict__ v,
                                                          const float* __restrict__ b_add, float* __restrict__ b_out, int K) {
  const int n = blockIdx.x;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) acc = fmaf(bf2f(W[(long)n * ldw + k]), v[k], acc);
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) b_out[n] = (b_add ? b_add[n] : 0.f) + ((red[0] + red[1]) + (red[2] + red[3]));
}

// Summed weights of the four phase planes of a nearest-2x upsample + 3x3 conv (gemm.h GemmArgs::phase2x):
//   WP[py * 2 + px][n][(a * 2 + b) * C + c] = sum over the 3x3 taps (ky, kx) that land on source offset (a, b) for phase (py, px) of
//   W[n][(ky * 3 + kx) * C + c],     rows: py = 0 -> a = 0: {ky = 0}, a = 1: {1, 2};  py = 1 -> a = 0: {0, 1}, a = 1: {2};  columns alike.
// Sums in fp32 of the packed bf16 taps, rounded once.  One block per output channel.
__global__ __launch_bounds__(256) void ups_phase_fold_kernel(const bf16_t* __restrict__ W, int ldw, bf16_t* __restrict__ WP, int N, int C) {
  const int n = blockIdx.x;
  const bf16_t* w = W + (long)n * ldw;
  for (int i = threadIdx.x; i < 16 * C; i += 256) {
    const int c = i % C, sab = (i / C) & 3, ph = i / (4 * C);
    const int py = ph >> 1, px = ph & 1, a = sab >> 1, b = sab & 1;
    const int ky0 = a == 0 ? 0 : (py == 0 ? 1 : 2), ky1 = a == 0 ? (py == 0 ? 0 : 1) : 2;
    const int kx0 = b == 0 ? 0 : (px == 0 ? 1 : 2), kx1 = b == 0 ? (px == 0 ? 0 : 1) : 2;
    float acc = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx) acc += bf2f(w[(ky * 3 + kx) * C + c]);
    WP[((long)ph * N + n) * (4 * C) + sab * C + c] = f2bf(acc);
  }
}

}  // namespace

namespace dfh {

int ups_phase_fold_launch(const bf16_t* W, int ldw, bf16_t* WP, int N, int C, hipStream_t stream) {
  DFH_REQUIRE(W && WP && N > 0 && C > 0 && ldw >= 9 * C, "bad argument");
  hipLaunchKernelGGL(ups_phase_fold_kernel, dim3(N), dim3(256), 0, stream, W, ldw, WP, N, C);
  return check_launch("ups_phase_fold_kernel");
}

int matvec_bias_launch(const bf16_t* W, int ldw, const float* v, const float* b_add, float* b_out, int N, int K, hipStream_t stream) {
  DFH_REQUIRE(W && v && b_out && N > 0 && K > 0, "bad argument");
  hipLaunchKernelGGL(matvec_bias_kernel, dim3(N), dim3(256), 0, stream, W, ldw, v, b_add, b_out, K);
  return check_launch("matvec_bias_kernel");
}

int ln_fold_launch(const bf16_t* W, int ldw, const float* gamma, const float* beta, const float* bias, bf16_t* WF, float* s, float* b,
                   int N, int K, hipStream_t stream) {
  DFH_REQUIRE(W && gamma && beta && WF && s && b, "null pointer");
  DFH_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0, "K and the row stride must be multiples of 8");
  hipLaunchKernelGGL(ln_fold_kernel, dim3(N), dim3(256), 0, stream, W, ldw, gamma, beta, bias, WF, s, b, K);
  return check_launch("ln_fold_kernel");
}

}  // namespace dfh
