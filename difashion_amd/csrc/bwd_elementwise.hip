// Elementwise / layout kernels of the training step (reference: DiFashion/train.py:691-716 --
// accelerator.backward, clip_grad_norm_, AdamW, EMA -- and the autograd of the glue in
// DiFashion/models/difashion.py:160-267).  HBM-bound; bf16 data as 16-byte vectors, fp32 state as float4.
#include <map>
#include <mutex>
#include <utility>
#include <unordered_map>
#include "dfh_common.h"
#include "bwd_elementwise.h"

namespace {

constexpr int EW_BLOCK = 256;
DFH_DEVICE long gtid() { return (long)blockIdx.x * blockDim.x + threadIdx.x; }
inline dim3 ew_grid(long n) { return dim3((unsigned)((n + EW_BLOCK - 1) / EW_BLOCK)); }

DFH_DEVICE int geglu_row(int n, int N) {
  const int half = N >> 1;
  const int j = n < half ? n : n - half;
  return (j >> 4) * 32 + (n < half ? 0 : 16) + (j & 15);
}

// ---- transposed weight packing for data-gradient GEMMs ------------------------------------------------------
// linear / 1x1: w [N][K] fp32 -> Wt[t_row_off + k][t_col_off + perm(n)] bf16  (dX = dY . W  ==  NT GEMM against Wt)
__global__ void pack_matrix_t_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int N, int K, int ldt,
                                     int t_row_off, int t_col_off, int geglu) {
  const long i = gtid();
  if (i >= (long)N * K) return;
  const int n = (int)(i / K), k = (int)(i - (long)n * K);
  const int r = geglu ? geglu_row(n, N) : n;
  out[(long)(t_row_off + k) * ldt + t_col_off + r] = f2bf(w[i]);
}
// conv3x3: w [O][I][3][3] -> W'[c][t_col_off + (8 - t) * o_pad + o]: spatially flipped, in/out swapped, so that
// dX = conv3x3(dY, W') with the same implicit-GEMM kernel
__global__ void pack_conv3x3_t_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int ldt,
                                      int t_col_off, int o_pad) {
  const long i = gtid();
  if (i >= (long)Cout * Cin * 9) return;
  const int t = (int)(i % 9);
  const long oc = i / 9;
  const int c = (int)(oc % Cin), o = (int)(oc / Cin);
  out[(long)c * ldt + t_col_off + (8 - t) * o_pad + o] = f2bf(w[i]);
}

// ---- gradient un-packing: packed fp32 gradient arena -> master-layout .grad (+=) ------------------------------
__global__ void unpack_matrix_kernel(const float* __restrict__ g, float* __restrict__ grad, int N, int K, int ldw,
                                     int row_off, int col_off, int geglu) {
  const long i = gtid();
  if (i >= (long)N * K) return;
  const int n = (int)(i / K), k = (int)(i - (long)n * K);
  const int r = geglu ? geglu_row(n, N) : n;
  grad[i] += g[(long)(row_off + r) * ldw + col_off + k];
}
__global__ void unpack_conv3x3_kernel(const float* __restrict__ g, float* __restrict__ grad, int Cout, int Cin, int ldw,
                                      int col_off, int cin_pad) {
  const long i = gtid();
  if (i >= (long)Cout * Cin * 9) return;
  const int t = (int)(i % 9);
  const long oc = i / 9;
  const int c = (int)(oc % Cin), o = (int)(oc / Cin);
  grad[i] += g[(long)o * ldw + col_off + t * cin_pad + c];
}
__global__ void unpack_vector_kernel(const float* __restrict__ g, float* __restrict__ grad, int N, int off, int geglu) {
  const long i = gtid();
  if (i >= N) return;
  const int r = geglu ? geglu_row((int)i, N) : (int)i;
  grad[i] += g[off + r];
}

// ---- phase planes of an upsample conv in the backward pass (gemm.h GemmArgs::phase2x, wgrad.h WgradArgs::tap2) -----------------------------
// output gradient dY [B][2H][2W][C] -> phase-major [4][B][H][W][C]: plane (py, px) holds the pixels (2y + py, 2x + px)
__global__ void phase_gather_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int H, int W, int C8) {
  const long i = gtid();
  const long per = (long)B * H * W * C8;
  if (i >= 4 * per) return;
  const int plane = (int)(i / per);
  long r = i - plane * per;
  const int o = (int)(r % C8); r /= C8;
  const int x = (int)(r % W); r /= W;
  const int y = (int)(r % H);
  const int b = (int)(r / H);
  const int py = plane >> 1, px = plane & 1;
  ((uint4*)out)[i] = *(const uint4*)(in + ((((long)b * 2 * H + 2 * y + py) * (2 * W) + 2 * x + px) * C8 + o) * 8);
}
// dx (=|+=) sum of the four planes' data gradients [4][n] (fp32 sum, one rounding)
__global__ void phase_sum4_kernel(const bf16_t* __restrict__ planes, bf16_t* __restrict__ dx, long n8, int accumulate) {
  const long i = gtid();
  if (i >= n8) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (accumulate) unpack8(((const uint4*)dx)[i], acc);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float f[8];
    unpack8(((const uint4*)planes)[p * n8 + i], f);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += f[k];
  }
  ((uint4*)dx)[i] = pack8(acc);
}
// gradient of the 3x3 weights (packed [N][9 * C], fp32) from the gradients of the four planes' SUMMED weights dWp [4][N][4 * C]:
// plane (py, px) folded tap (ky, kx) into its slot (sr, sc) with sr = (py == 0 ? ky >= 1 : ky == 2), likewise sc (lnfold.hip
// ups_phase_fold), so dW3[ky][kx] = sum over the four planes of dWp[plane][slot(plane, ky, kx)]
__global__ void ups_phase_unfold_kernel(const float* __restrict__ dwp, float* __restrict__ dw3, int N, int C, int ldw, int overwrite) {
  const long i = gtid();
  if (i >= (long)N * 9 * C) return;
  const int c = (int)(i % C);
  long r = i / C;
  const int t = (int)(r % 9), n = (int)(r / 9);
  const int ky = t / 3, kx = t - ky * 3;
  float acc = 0.f;
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      const int sr = py == 0 ? (ky >= 1) : (ky == 2), sc = px == 0 ? (kx >= 1) : (kx == 2);
      acc += dwp[(((long)(py * 2 + px) * N + n) * 4 + sr * 2 + sc) * C + c];
    }
  float* dst = dw3 + (long)n * ldw + t * C + c;
  *dst = overwrite ? acc : *dst + acc;
}

// ---- activations / layout ---------------------------------------------------------------------------------
// nearest-2x upsample backward: out[b][y][x][c] = sum of the 2x2 block of in[b][2y..][2x..][c]   (NHWC bf16)
__global__ void pool2x2_sum_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int H, int W, int C8) {
  const long i = gtid();
  if (i >= (long)B * H * W * C8) return;
  const int o = (int)(i % C8);
  long p = i / C8;
  const int x = (int)(p % W); p /= W;
  const int y = (int)(p % H);
  const int b = (int)(p / H);
  const int C = C8 * 8, W2 = 2 * W;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      float f[8];
      unpack8(*(const uint4*)(in + (((long)b * 2 * H + 2 * y + dy) * W2 + 2 * x + dx) * C + o * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += f[k];
    }
  *(uint4*)(out + i * 8) = pack8(acc);
}

// dst (=|+=) src, bf16 vectors
__global__ void add_bf16_kernel(bf16_t* __restrict__ dst, const bf16_t* __restrict__ src, long n8, int accumulate) {
  const long i = gtid();
  if (i >= n8) return;
  uint4 s = *(const uint4*)(src + i * 8);
  if (accumulate) {
    float a[8], b[8];
    unpack8(*(const uint4*)(dst + i * 8), a);
    unpack8(s, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
    s = pack8(a);
  }
  *(uint4*)(dst + i * 8) = s;
}

// GEGLU backward on the packed (16-value / 16-gate interleaved) pre-activation layout:
//   y = v * gelu(g)  ->  dv = dy * gelu(g),  dg = dy * v * gelu'(g);  gelu'(g) = Phi(g) + g * phi(g)
__global__ void geglu_bwd_kernel(const bf16_t* __restrict__ pre, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dpre,
                                 long M, int N2) {   // N2 = 8C packed columns; dy has N2/2 columns
  const long i = gtid();                             // one thread per (row, 16-column value block half = 8 columns)
  const int blocks8 = N2 / 32 * 2;                   // 8-wide pieces of the value halves per row
  if (i >= M * blocks8) return;
  const long m = i / blocks8;
  const int pb = (int)(i - m * blocks8);
  const int blk = pb >> 1, half = pb & 1;
  const bf16_t* prow = pre + m * N2 + blk * 32 + half * 8;
  float v[8], g[8], d[8], dv[8], dg[8];
  unpack8(*(const uint4*)prow, v);
  unpack8(*(const uint4*)(prow + 16), g);
  unpack8(*(const uint4*)(dy + m * (N2 / 2) + blk * 16 + half * 8), d);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float x = g[k];
    const float cdf = 0.5f * (1.0f + erf_as_f(x * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    dv[k] = d[k] * x * cdf;
    dg[k] = d[k] * v[k] * (cdf + x * pdf);
  }
  bf16_t* orow = dpre + m * N2 + blk * 32 + half * 8;
  *(uint4*)orow = pack8(dv);
  *(uint4*)(orow + 16) = pack8(dg);
}
// forward counterpart used in training mode (the inference path fuses this into the GEMM epilogue)
__global__ void geglu_fwd_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ y, long M, int N2) {
  const long i = gtid();
  const int blocks8 = N2 / 32 * 2;
  if (i >= M * blocks8) return;
  const long m = i / blocks8;
  const int pb = (int)(i - m * blocks8);
  const int blk = pb >> 1, half = pb & 1;
  const bf16_t* prow = pre + m * N2 + blk * 32 + half * 8;
  float v[8], g[8];
  unpack8(*(const uint4*)prow, v);
  unpack8(*(const uint4*)(prow + 16), g);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = v[k] * gelu_erf_f(g[k]);
  *(uint4*)(y + m * (N2 / 2) + blk * 16 + half * 8) = pack8(v);
}

// pointwise activation forward / backward on bf16 rows (time-embedding MLP, MutualEncoder)
//   kind 1 silu (from pre-activation), 2 leaky_relu 0.01 (sign from the OUTPUT), 3 tanh (from the OUTPUT)
__global__ void act_fwd_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ y, long n, int kind) {
  const long i = gtid();
  if (i >= n) return;
  const float x = bf2f(pre[i]);
  y[i] = f2bf(kind == 1 ? silu_f(x) : (kind == 2 ? (x > 0.f ? x : 0.01f * x) : tanhf(x)));
}
__global__ void act_bwd_kernel(const bf16_t* __restrict__ ref, const float* __restrict__ ref_f32, const bf16_t* __restrict__ dy,
                               const float* __restrict__ dy_f32, bf16_t* __restrict__ dpre, long n, int kind, float scale) {
  const long i = gtid();
  if (i >= n) return;
  const float r = ref_f32 ? ref_f32[i] : bf2f(ref[i]);
  const float d = (dy_f32 ? dy_f32[i] : bf2f(dy[i])) * scale;
  float g;
  if (kind == 1) { const float s = 1.0f / (1.0f + __expf(-r)); g = s * (1.0f + r * (1.0f - s)); }
  else if (kind == 2) g = r > 0.f ? 1.0f : 0.01f;
  else g = 1.0f - r * r;
  dpre[i] = f2bf(d * g);
}

// NHWC bf16 [B][HW][Cp] -> NCHW fp32 [B][C][HW] (first C channels), dst (=|+=) scale * src
__global__ void nhwc_to_nchw_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int B, int HW, int Cp, int C,
                                        float scale, int accumulate) {
  const long i = gtid();
  if (i >= (long)B * C * HW) return;
  const int p = (int)(i % HW);
  const long bc = i / HW;
  const int c = (int)(bc % C), b = (int)(bc / C);
  const float v = scale * bf2f(src[((long)b * HW + p) * Cp + c]);
  dst[i] = accumulate ? dst[i] + v : v;
}

// bf16 [B][R][ld_in] (C columns used) -> [B][C][ld_out] (R columns written): V -> V^T for the forward attention
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C, int ld_in,
                                      int ld_out, long in_bstride, long out_bstride) {
  __shared__ bf16_t tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? in[b * in_bstride + (long)r * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < C && r < R) out[b * out_bstride + (long)c * ld_out + r] = tile[tx][j];
  }
}

// ---- loss / glue backward -----------------------------------------------------------------------------------
// d pred = w[row] * 2 * (pred - target) / (rows * L)     (mean over rows of per-row MSE * weight; df.py:255-265)
__global__ void mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ w,
                               float* __restrict__ dpred, int rows, int L, float loss_scale,
                               const float* __restrict__ scale_dev) {
  const long i = gtid();
  if (i >= (long)rows * L) return;
  const int r = (int)(i / L);
  if (scale_dev) loss_scale *= *scale_dev;      // upstream d loss (autograd) stays on the device: no host sync
  const float ww = (w ? w[r] : 1.0f) * loss_scale * 2.0f / ((float)rows * (float)L);
  dpred[i] = ww * (pred[i] - target[i]);
}
// input assembly backward (df.py:215): d mutual[row][e] = eta * dx[row][0:CL][e] where the mutual condition was real
__global__ void assemble_bwd_kernel(const float* __restrict__ dx, const unsigned char* __restrict__ mutual_real,
                                    float* __restrict__ dmutual, int rows, int CL, float eta) {
  const long i = gtid();
  if (i >= (long)rows * CL) return;
  const int r = (int)(i / CL), e = (int)(i - (long)r * CL);
  dmutual[i] = mutual_real[r] ? eta * dx[(long)r * 2 * CL + e] : 0.0f;
}

// ---- optimizer (train.py:586-593,700-711) ---------------------------------------------------------------------
// *out += sum(g^2), bit-reproducible: per-block partials in a fixed order, summed in index order by whichever block arrives
// last (ticket counter) -- no float atomics.  Data-parallel replicas derive their clip coefficient from this number: with an
// atomic accumulation they drift apart in the last bits after every step (tests/test_gpu_ddp.py).
__global__ void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out, float* __restrict__ partials,
                             unsigned* __restrict__ counter) {
  __shared__ float red[4];
  __shared__ bool last;
  float s = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  if (((uintptr_t)g & 15) == 0) {                                // 16-byte loads, four running sums (fixed order for a fixed grid)
    const long n4 = n >> 2;
    float4 acc = float4{0.f, 0.f, 0.f, 0.f};
    long i = gtid();
    for (; i + 3 * stride < n4; i += 4 * stride) {               // four loads in flight per thread
      const float4 a = ((const float4*)g)[i], b = ((const float4*)g)[i + stride], c = ((const float4*)g)[i + 2 * stride],
                   d = ((const float4*)g)[i + 3 * stride];
      acc.x += a.x * a.x + b.x * b.x + c.x * c.x + d.x * d.x; acc.y += a.y * a.y + b.y * b.y + c.y * c.y + d.y * d.y;
      acc.z += a.z * a.z + b.z * b.z + c.z * c.z + d.z * d.z; acc.w += a.w * a.w + b.w * b.w + c.w * c.w + d.w * d.w;
    }
    for (; i < n4; i += stride) {
      const float4 a = ((const float4*)g)[i];
      acc.x += a.x * a.x; acc.y += a.y * a.y; acc.z += a.z * a.z; acc.w += a.w * a.w;
    }
    s = (acc.x + acc.y) + (acc.z + acc.w);
    for (long j = (n4 << 2) + gtid(); j < n; j += stride) s += g[j] * g[j];
  } else {
    for (long i = gtid(); i < n; i += stride) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __threadfence();                                             // release: the partial is visible before the ticket
    last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();                                               // acquire: every other block's partial
  float t = 0.f;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) t += ((const volatile float*)partials)[i];
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    *out += (red[0] + red[1]) + (red[2] + red[3]);
    *counter = 0u;                                               // ready for the next launch on this stream
  }
}
// torch.optim.AdamW step (decoupled weight decay), gradient pre-scaled by clip_coef = min(1, max_norm / (norm + 1e-6))
// read from device memory so the clip needs no host synchronisation
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                             const float* __restrict__ sumsq, float max_norm, float* __restrict__ shadow, float ema_omd) {
  const long i = gtid();
  if (i >= n) return;
  float clip = 1.0f;
  if (sumsq) { const float nrm = sqrtf(*sumsq); clip = fminf(1.0f, max_norm / (nrm + 1e-6f)); }
  const float gi = g[i] * clip;
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi;
  if (shadow) {                  // diffusers EMAModel.step folded into the same pass: shadow -= (1 - decay) * (shadow - p)
    const float sh = shadow[i];
    shadow[i] = sh - ema_omd * (sh - pi);
  }
}
// diffusers EMAModel.step: shadow -= (1 - decay) * (shadow - param)
// ---- bf16 wire format of the data-parallel gradient exchange (difashion_amd/dist.py exchange_bf16; reference: DDP's all-reduce behind
//      accelerator.backward, train.py:611,699).  Eight elements per thread, 16-byte accesses.
// fp32 range -> bf16 wire (round to nearest even), zero-filled up to the padded length (a multiple of 8 per shard)
__global__ void wire_pack_kernel(const float* __restrict__ g, bf16_t* __restrict__ wire, long n, long n_pad8) {
  const long i = gtid();
  if (i >= n_pad8) return;
  float f[8];
  if (i * 8 + 8 <= n) {
    const float4 a = *(const float4*)(g + i * 8), b = *(const float4*)(g + i * 8 + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) f[r] = (i * 8 + r < n) ? g[i * 8 + r] : 0.f;
  }
  *(uint4*)(wire + i * 8) = pack8(f);
}
// this rank's shard: the W contributions summed in fp32 IN RANK ORDER (deterministic; every rank computes its own shard only, so all
// ranks end with identical values), divided by W, rounded to bf16 once
__global__ void wire_shard_mean_kernel(const bf16_t* __restrict__ recv, bf16_t* __restrict__ shard, int world, long per8, float wf) {
  const long i = gtid();
  if (i >= per8) return;
  float acc[8];
  unpack8(*(const uint4*)(recv + i * 8), acc);
  for (int r = 1; r < world; ++r) {
    float f[8];
    unpack8(*(const uint4*)(recv + ((long)r * per8 + i) * 8), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += f[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = acc[k] / wf;
  *(uint4*)(shard + i * 8) = pack8(acc);
}
// bf16 wire -> fp32 range
__global__ void wire_unpack_kernel(const bf16_t* __restrict__ wire, float* __restrict__ g, long n) {
  const long i = gtid();
  if (i * 8 >= n) return;
  float f[8];
  unpack8(*(const uint4*)(wire + i * 8), f);
  if (i * 8 + 8 <= n) {
    *(float4*)(g + i * 8) = float4{f[0], f[1], f[2], f[3]};
    *(float4*)(g + i * 8 + 4) = float4{f[4], f[5], f[6], f[7]};
  } else {
    for (int r = 0; i * 8 + r < n; ++r) g[i * 8 + r] = f[r];
  }
}

__global__ void ema_kernel(float* __restrict__ shadow, const float* __restrict__ p, long n, float one_minus_decay) {
  const long i = gtid();
  if (i >= n) return;
  const float s = shadow[i];
  shadow[i] = s - one_minus_decay * (s - p[i]);
}

}  // namespace

namespace dfh {
#define EW_LAUNCH(kernel, n, ...)                                                      \
  hipLaunchKernelGGL(kernel, ew_grid(n), dim3(EW_BLOCK), 0, s, __VA_ARGS__);           \
  return check_launch(#kernel)

int pack_matrix_t_launch(const float* w, bf16_t* out, int N, int K, int ldt, int t_row_off, int t_col_off, int geglu, hipStream_t s) {
  EW_LAUNCH(pack_matrix_t_kernel, (long)N * K, w, out, N, K, ldt, t_row_off, t_col_off, geglu);
}
int pack_conv3x3_t_launch(const float* w, bf16_t* out, int Cout, int Cin, int ldt, int t_col_off, int o_pad, hipStream_t s) {
  if (o_pad <= 0) o_pad = Cout;
  EW_LAUNCH(pack_conv3x3_t_kernel, (long)Cout * Cin * 9, w, out, Cout, Cin, ldt, t_col_off, o_pad);
}
int unpack_matrix_launch(const float* g, float* grad, int N, int K, int ldw, int row_off, int col_off, int geglu, hipStream_t s) {
  EW_LAUNCH(unpack_matrix_kernel, (long)N * K, g, grad, N, K, ldw, row_off, col_off, geglu);
}
int unpack_conv3x3_launch(const float* g, float* grad, int Cout, int Cin, int ldw, int col_off, int cin_pad, hipStream_t s) {
  if (cin_pad <= 0) cin_pad = Cin;
  EW_LAUNCH(unpack_conv3x3_kernel, (long)Cout * Cin * 9, g, grad, Cout, Cin, ldw, col_off, cin_pad);
}
int unpack_vector_launch(const float* g, float* grad, int N, int off, int geglu, hipStream_t s) {
  EW_LAUNCH(unpack_vector_kernel, (long)N, g, grad, N, off, geglu);
}
int phase_gather_launch(const bf16_t* in, bf16_t* out, int B, int H, int W, int C, hipStream_t s) {
  DFH_REQUIRE(C % 8 == 0, "phase gather: channels must be a multiple of 8");
  EW_LAUNCH(phase_gather_kernel, 4L * B * H * W * (C / 8), in, out, B, H, W, C / 8);
}
int phase_sum4_launch(const bf16_t* planes, bf16_t* dx, long n, int accumulate, hipStream_t s) {
  DFH_REQUIRE(n % 8 == 0, "phase sum: element count must be a multiple of 8");
  EW_LAUNCH(phase_sum4_kernel, n / 8, planes, dx, n / 8, accumulate);
}
int ups_phase_unfold_launch(const float* dwp, float* dw3, int N, int C, int ldw, int overwrite, hipStream_t s) {
  EW_LAUNCH(ups_phase_unfold_kernel, (long)N * 9 * C, dwp, dw3, N, C, ldw, overwrite);
}
int pool2x2_sum_launch(const bf16_t* in, bf16_t* out, int B, int H, int W, int C, hipStream_t s) {
  DFH_REQUIRE(C % 8 == 0, "channels must be a multiple of 8");
  EW_LAUNCH(pool2x2_sum_kernel, (long)B * H * W * (C / 8), in, out, B, H, W, C / 8);
}
int add_bf16_launch(bf16_t* dst, const bf16_t* src, long n, int accumulate, hipStream_t s) {
  DFH_REQUIRE(n % 8 == 0, "length must be a multiple of 8");
  EW_LAUNCH(add_bf16_kernel, n / 8, dst, src, n / 8, accumulate);
}
int geglu_bwd_launch(const bf16_t* pre, const bf16_t* dy, bf16_t* dpre, long M, int N2, hipStream_t s) {
  DFH_REQUIRE(N2 % 32 == 0, "GEGLU width must be a multiple of 32");
  EW_LAUNCH(geglu_bwd_kernel, M * (N2 / 16), pre, dy, dpre, M, N2);
}
int geglu_fwd_launch(const bf16_t* pre, bf16_t* y, long M, int N2, hipStream_t s) {
  DFH_REQUIRE(N2 % 32 == 0, "GEGLU width must be a multiple of 32");
  EW_LAUNCH(geglu_fwd_kernel, M * (N2 / 16), pre, y, M, N2);
}
int act_fwd_launch(const bf16_t* pre, bf16_t* y, long n, int kind, hipStream_t s) {
  EW_LAUNCH(act_fwd_kernel, n, pre, y, n, kind);
}
int act_bwd_launch(const bf16_t* ref, const float* ref_f32, const bf16_t* dy, const float* dy_f32, bf16_t* dpre, long n, int kind,
                   float scale, hipStream_t s) {
  DFH_REQUIRE((ref || ref_f32) && (dy || dy_f32), "act_bwd: missing operand");
  EW_LAUNCH(act_bwd_kernel, n, ref, ref_f32, dy, dy_f32, dpre, n, kind, scale);
}
int nhwc_to_nchw_f32_launch(const bf16_t* src, float* dst, int B, int HW, int Cp, int C, float scale, int accumulate, hipStream_t s) {
  EW_LAUNCH(nhwc_to_nchw_f32_kernel, (long)B * C * HW, src, dst, B, HW, Cp, C, scale, accumulate);
}
int transpose_bf16_launch(const bf16_t* in, bf16_t* out, int B, int R, int C, int ld_in, int ld_out, long in_bstride, long out_bstride,
                          hipStream_t s) {
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 31) / 32, (R + 31) / 32, B), dim3(256), 0, s, in, out, R, C, ld_in, ld_out,
                     in_bstride, out_bstride);
  return check_launch("transpose_bf16_kernel");
}
int mse_bwd_launch(const float* pred, const float* target, const float* w, float* dpred, int rows, int L, float loss_scale,
                   const float* scale_dev, hipStream_t s) {
  EW_LAUNCH(mse_bwd_kernel, (long)rows * L, pred, target, w, dpred, rows, L, loss_scale, scale_dev);
}
int assemble_bwd_launch(const float* dx, const unsigned char* mutual_real, float* dmutual, int rows, int CL, float eta, hipStream_t s) {
  EW_LAUNCH(assemble_bwd_kernel, (long)rows * CL, dx, mutual_real, dmutual, rows, CL, eta);
}
constexpr unsigned SUMSQ_MAX_BLOCKS = 2048;
// per-stream scratch of sumsq_kernel: SUMSQ_MAX_BLOCKS partials + the ticket counter (zeroed once; the kernel resets it)
static float* sumsq_scratch_for(hipStream_t stream) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, float*> table;     // per (device, stream): the null stream exists on every device
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const auto key = std::make_pair(dev, stream);
  auto it = table.find(key);
  if (it != table.end()) return it->second;
  float* p = nullptr;
  if (hipMalloc((void**)&p, (SUMSQ_MAX_BLOCKS + 1) * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMemset(p, 0, (SUMSQ_MAX_BLOCKS + 1) * sizeof(float)) != hipSuccess) { (void)hipFree(p); return nullptr; }
  table[key] = p;
  return p;
}
int sumsq_launch(const float* g, long n, float* out, hipStream_t s) {
  float* scratch = sumsq_scratch_for(s);
  DFH_REQUIRE(scratch != nullptr, "sumsq scratch allocation failed");
  ProfScope ps(PC_OPTIM, 0.0, 4.0 * n, s);
  const unsigned blocks = (unsigned)std::max<long>(1, std::min<long>((n + EW_BLOCK - 1) / EW_BLOCK, SUMSQ_MAX_BLOCKS));
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(EW_BLOCK), 0, s, g, n, out, scratch, (unsigned*)(scratch + SUMSQ_MAX_BLOCKS));
  return check_launch("sumsq_kernel");
}
int adamw_launch(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
                 int step, const float* sumsq, float max_norm, hipStream_t s, float* shadow, float ema_decay) {
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  ProfScope ps(PC_OPTIM, 0.0, (shadow ? 36.0 : 28.0) * n, s);      // p r+w, g r, m r+w, v r+w (+ shadow r+w)
  EW_LAUNCH(adamw_kernel, n, p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2, sumsq, max_norm, shadow, 1.0f - ema_decay);
}
int wire_pack_launch(const float* g, bf16_t* wire, long n, long n_pad, hipStream_t s) {
  DFH_REQUIRE(n >= 0 && n_pad >= n && n_pad % 8 == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)wire % 16) == 0, "wire_pack: padded length % 8, 16-byte aligned buffers");
  EW_LAUNCH(wire_pack_kernel, n_pad / 8, g, wire, n, n_pad / 8);
}
int wire_shard_mean_launch(const bf16_t* recv, bf16_t* shard, int world, long per, hipStream_t s) {
  DFH_REQUIRE(world >= 1 && per % 8 == 0 && ((uintptr_t)recv % 16) == 0 && ((uintptr_t)shard % 16) == 0, "wire_shard_mean: shard length % 8, 16-byte aligned buffers");
  EW_LAUNCH(wire_shard_mean_kernel, per / 8, recv, shard, world, per / 8, (float)world);
}
int wire_unpack_launch(const bf16_t* wire, float* g, long n, hipStream_t s) {
  DFH_REQUIRE(((uintptr_t)g % 16) == 0 && ((uintptr_t)wire % 16) == 0, "wire_unpack: 16-byte aligned buffers");
  EW_LAUNCH(wire_unpack_kernel, (n + 7) / 8, wire, g, n);
}
int ema_launch(float* shadow, const float* p, long n, float decay, hipStream_t s) {
  ProfScope ps(PC_OPTIM, 0.0, 12.0 * n, s);
  EW_LAUNCH(ema_kernel, n, shadow, p, n, 1.0f - decay);
}
}  // namespace dfh
