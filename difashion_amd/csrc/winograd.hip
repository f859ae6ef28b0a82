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

// GroupNorm(+SiLU) and the input transform in ONE launch: a workgroup owns one (image, group) slab -- H W pixels x C / G channels, at most
// 16 x 16 x 80 here -- keeps it in LDS as bf16, computes the statistics (two passes: mean, centred squares), normalises in place (rounded to
// bf16 exactly where the unfused GroupNorm kernel rounds its output) and transforms 4x4 patches straight out of LDS.  Saves the normalised
// tensor's round trip through HBM and one launch per conv.  Channels come from two sources (the up blocks' skip concat), 4-channel units.
struct GnWinoArgs {
  const bf16_t* src0; const bf16_t* src1; int C0, C1;   // [B][HW][C0], [B][HW][C1]
  const float* gamma; const float* beta; float eps;
  bf16_t* V; int B, H, W, G;
  // conv1 -> conv2 inside a resnet: the slab is not read from a tensor but rebuilt from the PREVIOUS conv's transform-domain output
  // Mprev [16][B H W / 4][C0] (its output transform A^T m A + bias + time-embedding row, rounded to bf16 as the stored tensor would be):
  // no output-transform launch, no round trip of the tensor between the two convs
  const bf16_t* Mprev; const float* pbias; const float* prowvec; int prv_ld, prv_off;
};

__global__ __launch_bounds__(256) void gn_wino_input_kernel(const GnWinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int C = a.C0 + a.C1, cpg = C / a.G, upp = cpg >> 2, HW = a.H * a.W;
  const int cg = g * cpg, total = HW * upp;
  const int RS = cpg * 2 + 8;                                  // LDS row stride (bytes): one pixel's slab + 8 (bank spread)
  __shared__ float red[8];
  __shared__ __attribute__((aligned(16))) float gam_s[256], bet_s[256];
  if (tid < cpg) { gam_s[tid] = a.gamma[cg + tid]; bet_s[tid] = a.beta[cg + tid]; }
  // ---- pass 1: global -> LDS (raw bf16), sum
  float s = 0.f;
  if (a.Mprev) {
    // one item = one 2x2 output tile x one 4-channel unit of the previous conv: A^T m A + bias (+ time-embedding row), as wino_output_kernel
    const int TWp = a.W >> 1, tilesp = (a.H >> 1) * TWp;
    const long Mtp = (long)a.B * tilesp;
    for (int idx = tid; idx < tilesp * upp; idx += 256) {
      const int t = idx / upp, j = idx - t * upp, c = cg + j * 4;
      const int ty = t / TWp, tx = t - ty * TWp;
      float sm[2][4][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint2 r = *(const uint2*)(a.Mprev + ((long)(i * 4 + jj) * Mtp + (long)b * tilesp + t) * a.C0 + c);
          m[i][0] = __uint_as_float(r.x << 16); m[i][1] = __uint_as_float(r.x & 0xffff0000u);
          m[i][2] = __uint_as_float(r.y << 16); m[i][3] = __uint_as_float(r.y & 0xffff0000u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sm[0][jj][e] = m[0][e] + m[1][e] + m[2][e];
          sm[1][jj][e] = m[1][e] - m[2][e] - m[3][e];
        }
      }
      float4 add = *(const float4*)(a.pbias + c);
      if (a.prowvec) {
        const float4 rv = *(const float4*)(a.prowvec + (long)b * a.prv_ld + a.prv_off + c);
        add.x += rv.x; add.y += rv.y; add.z += rv.z; add.w += rv.w;
      }
      const float ad[4] = {add.x, add.y, add.z, add.w};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            y[e] = (jj == 0 ? sm[i][0][e] + sm[i][1][e] + sm[i][2][e] : sm[i][1][e] - sm[i][2][e] - sm[i][3][e]) + ad[e];
          uint2 r; r.x = pack2bf(y[0], y[1]); r.y = pack2bf(y[2], y[3]);
          const int p = (2 * ty + i) * a.W + 2 * tx + jj;
          *(uint2*)(smem + p * RS + j * 8) = r;
          s += (__uint_as_float(r.x << 16) + __uint_as_float(r.x & 0xffff0000u)) + (__uint_as_float(r.y << 16) + __uint_as_float(r.y & 0xffff0000u));
        }
    }
  } else
  for (int idx = tid; idx < total; idx += 256) {
    const int p = idx / upp, j = idx - p * upp, c = cg + j * 4;
    const uint2 r = c < a.C0 ? *(const uint2*)(a.src0 + ((long)b * HW + p) * a.C0 + c)
                             : *(const uint2*)(a.src1 + ((long)b * HW + p) * a.C1 + (c - a.C0));
    *(uint2*)(smem + p * RS + j * 8) = r;
    s += (__uint_as_float(r.x << 16) + __uint_as_float(r.x & 0xffff0000u)) + (__uint_as_float(r.y << 16) + __uint_as_float(r.y & 0xffff0000u));
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float n = (float)HW * (float)cpg;
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / n;
  // ---- pass 2: centred squares (each thread re-reads the units it wrote)
  float q = 0.f;
  for (int idx = tid; idx < total; idx += 256) {
    const int p = idx / upp, j = idx - p * upp;
    const uint2 r = *(const uint2*)(smem + p * RS + j * 8);
    const float v[4] = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = v[k] - mean; q += d * d; }
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / n + a.eps);
  // ---- pass 3: normalise + SiLU in place, rounded to bf16
  for (int idx = tid; idx < total; idx += 256) {
    const int p = idx / upp, j = idx - p * upp;
    uint2* slot = (uint2*)(smem + p * RS + j * 8);
    const uint2 r = *slot;
    const float4 gm = *(const float4*)(gam_s + j * 4), bt = *(const float4*)(bet_s + j * 4);
    const float y0 = silu_f((__uint_as_float(r.x << 16) - mean) * rstd * gm.x + bt.x);
    const float y1 = silu_f((__uint_as_float(r.x & 0xffff0000u) - mean) * rstd * gm.y + bt.y);
    const float y2 = silu_f((__uint_as_float(r.y << 16) - mean) * rstd * gm.z + bt.z);
    const float y3 = silu_f((__uint_as_float(r.y & 0xffff0000u) - mean) * rstd * gm.w + bt.w);
    uint2 o; o.x = pack2bf(y0, y1); o.y = pack2bf(y2, y3);
    *slot = o;
  }
  __syncthreads();
  // ---- transform: one item = one 4x4 patch x one 4-channel unit
  const int TH = a.H >> 1, TW = a.W >> 1, tiles = TH * TW;
  const long Mt = (long)a.B * tiles;
  for (int idx = tid; idx < tiles * upp; idx += 256) {
    const int t = idx / upp, j = idx - t * upp;
    const int ty = t / TW, tx = t - ty * TW;
    float d[4][4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 2 * ty - 1 + r;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const int x = 2 * tx - 1 + c4;
        uint2 v = uint2{0u, 0u};
        if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) v = *(const uint2*)(smem + (y * a.W + x) * RS + j * 8);
        d[r][c4][0] = __uint_as_float(v.x << 16); d[r][c4][1] = __uint_as_float(v.x & 0xffff0000u);
        d[r][c4][2] = __uint_as_float(v.y << 16); d[r][c4][3] = __uint_as_float(v.y & 0xffff0000u);
      }
    }
    float qv[4][4][4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qv[0][c4][e] = d[0][c4][e] - d[2][c4][e];
        qv[1][c4][e] = d[1][c4][e] + d[2][c4][e];
        qv[2][c4][e] = d[2][c4][e] - d[1][c4][e];
        qv[3][c4][e] = d[1][c4][e] - d[3][c4][e];
      }
    bf16_t* dst = a.V + ((long)b * tiles + t) * C + cg + j * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[4][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[0][e] = qv[i][0][e] - qv[i][2][e];
        v[1][e] = qv[i][1][e] + qv[i][2][e];
        v[2][e] = qv[i][2][e] - qv[i][1][e];
        v[3][e] = qv[i][1][e] - qv[i][3][e];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        uint2 o; o.x = pack2bf(v[jj][0], v[jj][1]); o.y = pack2bf(v[jj][2], v[jj][3]);
        *(uint2*)(dst + (long)(i * 4 + jj) * Mt * C) = o;
      }
    }
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

bool gn_wino_ok(int C0, int C1, int G, int H, int W) {
  const int C = C0 + C1;
  if (G <= 0 || C % G) return false;
  const int cpg = C / G;
  return cpg % 4 == 0 && cpg <= 256 && C0 % 4 == 0 && (H & 1) == 0 && (W & 1) == 0 && (size_t)H * W * (cpg * 2 + 8) <= 96 * 1024;
}

int gn_wino_input_launch(const bf16_t* src0, int C0, const bf16_t* src1, int C1, const float* gamma, const float* beta, float eps, int G,
                         bf16_t* V, int B, int H, int W, hipStream_t stream, const bf16_t* Mprev, const float* pbias, const float* prowvec,
                         int prv_ld, int prv_off) {
  DFH_REQUIRE((src0 || Mprev) && gamma && beta && V && (C1 == 0 || src1), "null pointer");
  DFH_REQUIRE(!Mprev || (pbias && C1 == 0), "chained form: the previous conv's bias, one source");
  DFH_REQUIRE(gn_wino_ok(C0, C1, G, H, W), "shape not supported by the fused GroupNorm + input transform (gn_wino_ok)");
  GnWinoArgs a; a.src0 = src0; a.src1 = src1; a.C0 = C0; a.C1 = C1; a.gamma = gamma; a.beta = beta; a.eps = eps; a.V = V;
  a.B = B; a.H = H; a.W = W; a.G = G;
  a.Mprev = Mprev; a.pbias = pbias; a.prowvec = prowvec; a.prv_ld = prv_ld; a.prv_off = prv_off;
  const int C = C0 + C1;
  const size_t lds = (size_t)H * W * ((C / G) * 2 + 8);
  static size_t lds_set = 0;
  if (lds > 48 * 1024 && lds > lds_set) {
    (void)hipFuncSetAttribute((const void*)gn_wino_input_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    lds_set = 96 * 1024;
  }
  ProfScope ps(PC_GNORM, 0.0, (double)B * H * W * C * 2.0 * 5.0, stream);
  hipLaunchKernelGGL(gn_wino_input_kernel, dim3(G, B), dim3(256), lds, stream, a);
  return check_launch("gn_wino_input_kernel");
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
