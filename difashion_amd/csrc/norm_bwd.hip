// GroupNorm(+SiLU) and LayerNorm backward for NHWC bf16 activations (training step: torch autograd of
// diffusers ResnetBlock2D.norm1/norm2, Transformer2DModel.norm, BasicTransformerBlock.norm1/2/3 reached
// through DiFashion/train.py:699).  Same thread mapping and deterministic group reduction as the
// forward kernels in norm.hip; per-channel dgamma / dbeta are accumulated with fp32 atomics.
//
//   x^ = (x - mu) * rstd,  z = gamma * x^ + beta,  y = silu(z) | z
//   dz = dy * silu'(z) | dy;  dbeta_c = sum dz;  dgamma_c = sum dz * x^
//   per (b, group), n = HW * cpg:  s1 = sum gamma*dz,  s2 = sum gamma*dz*x^
//   dx = rstd * (gamma*dz - s1/n - x^ * s2/n)
#include "dfh_common.h"
#include "norm.h"

namespace {

DFH_DEVICE const uint4* gnb_src(const GnBwdArgs& a, int b, int p, int o) {
  const int o0 = a.C0 >> 3;
  if (o < o0) return (const uint4*)(a.src0 + ((long)(b * a.HW + p) * a.C0 + o * 8));
  return (const uint4*)(a.src1 + ((long)(b * a.HW + p) * a.C1 + (o - o0) * 8));
}

// d/dz silu(z) = s (1 + z (1 - s)), s = sigmoid(z): hardware reciprocal instead of an IEEE division (~10 VALU instructions; these
// kernels run two passes over 2.9 G elements per training step and are VALU-heavy), as silu_f in dfh_common.h
DFH_DEVICE float dsilu(float z) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896340736f));
  return s * (1.0f + z * (1.0f - s));
}

// grid (chunks, B): per-channel sums A_c = sum dz, B_c = sum dz*x^ over this chunk's pixels
// (blocks of <= 512 threads, see groupnorm_bwd_launch: without the bound the compiler budgets 128 VGPRs for 1024 threads and spills in the loops)
__global__ __launch_bounds__(512) void gn_bwd_stats_kernel(const GnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [PL][C][2]
  const int C8 = a.C >> 3, cpg = a.C / a.G;
  const int tid = threadIdx.x;
  const int o = tid % C8, pl = tid / C8;
  const int b = blockIdx.y;
  const int p_begin = blockIdx.x * a.pix_per_chunk, p_end = min(a.HW, p_begin + a.pix_per_chunk);
  if (pl < a.PL) {
    float ga[8], be[8], mu[8], rs[8], A[8], Bq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = o * 8 + k, g = c / cpg;
      ga[k] = a.gamma[c]; be[k] = a.beta[c];
      mu[k] = a.stats[((long)b * a.G + g) * 2]; rs[k] = a.stats[((long)b * a.G + g) * 2 + 1];
      A[k] = 0.f; Bq[k] = 0.f;
    }
    auto accum = [&](const uint4& xr, const uint4& dr) {
      float x[8], d[8];
      unpack8(xr, x); unpack8(dr, d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (x[k] - mu[k]) * rs[k];
        const float dz = a.silu ? d[k] * dsilu(ga[k] * xh + be[k]) : d[k];
        A[k] += dz; Bq[k] += dz * xh;
      }
    };
    int p = p_begin + pl;
    for (; p + 3 * a.PL < p_end; p += 4 * a.PL) {    // eight 16-byte loads in flight per thread; accumulation order stays p-ascending
      uint4 xr[4], dr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xr[u] = *gnb_src(a, b, p + u * a.PL, o);
        dr[u] = *(const uint4*)(a.dy + ((long)(b * a.HW + p + u * a.PL) * a.C + o * 8));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accum(xr[u], dr[u]);
    }
    for (; p < p_end; p += a.PL) accum(*gnb_src(a, b, p, o), *(const uint4*)(a.dy + ((long)(b * a.HW + p) * a.C + o * 8)));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[((pl * a.C) + o * 8 + k) * 2 + 0] = A[k];
      red[((pl * a.C) + o * 8 + k) * 2 + 1] = Bq[k];
    }
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += blockDim.x) {   // reduce over pixel lanes into slot 0, publish dgamma / dbeta
    float sa = 0.f, sb = 0.f;
    for (int l = 0; l < a.PL; ++l) { sa += red[(l * a.C + c) * 2]; sb += red[(l * a.C + c) * 2 + 1]; }
    red[c * 2] = sa; red[c * 2 + 1] = sb;
    atomicAdd(a.dbeta + c, sa);
    atomicAdd(a.dgamma + c, sb);
  }
  __syncthreads();
  if (tid < a.G) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { s1 += a.gamma[c] * red[c * 2]; s2 += a.gamma[c] * red[c * 2 + 1]; }
    float* dst = a.partial + (((long)b * a.G + tid) * gridDim.x + blockIdx.x) * 2;   // [b][group][chunk][2], as the forward
    dst[0] = s1; dst[1] = s2;
  }
}

// grid (achunks, B): dx
__global__ __launch_bounds__(512) void gn_bwd_apply_kernel(const GnBwdArgs a) {
  __shared__ float s1_s[64], s2_s[64];
  const int C8 = a.C >> 3, cpg = a.C / a.G;
  const int tid = threadIdx.x;
  const int o = tid % C8, pl = tid / C8;
  const int b = blockIdx.y;
  // the image's partials per group, summed by eight threads per group in a fixed order (see gn_apply_kernel in norm.hip)
  __shared__ float2 run_s[64][8];
  const bool wide_red = blockDim.x >= (unsigned)a.G * 8u;
  if (wide_red) {
    if (tid < a.G * 8) {
      const int g = tid >> 3, sub = tid & 7;
      const int per = (a.chunks + 7) >> 3, c0 = sub * per, c1 = min(a.chunks, c0 + per);
      const float2* src = (const float2*)(a.partial + ((long)b * a.G + g) * a.chunks * 2);
      float s1 = 0.f, s2 = 0.f;
      for (int c = c0; c < c1; ++c) { const float2 v = src[c]; s1 += v.x; s2 += v.y; }
      run_s[g][sub] = float2{s1, s2};
    }
    __syncthreads();
  }
  if (tid < a.G) {
    float s1 = 0.f, s2 = 0.f;
    if (wide_red) {
#pragma unroll
      for (int sub = 0; sub < 8; ++sub) { s1 += run_s[tid][sub].x; s2 += run_s[tid][sub].y; }
    } else {
      const float2* src = (const float2*)(a.partial + ((long)b * a.G + tid) * a.chunks * 2);
      for (int c = 0; c < a.chunks; ++c) { s1 += src[c].x; s2 += src[c].y; }
    }
    const float n = (float)a.HW * (float)cpg;
    s1_s[tid] = s1 / n; s2_s[tid] = s2 / n;
  }
  __syncthreads();
  if (pl >= a.PL) return;
  float ga[8], be[8], mu[8], rs[8], m1[8], m2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = o * 8 + k, g = c / cpg;
    ga[k] = a.gamma[c]; be[k] = a.beta[c];
    mu[k] = a.stats[((long)b * a.G + g) * 2]; rs[k] = a.stats[((long)b * a.G + g) * 2 + 1];
    m1[k] = s1_s[g]; m2[k] = s2_s[g];
  }
  const int o0 = a.C0 >> 3;
  const bool first = o < o0;
  bf16_t* dbase = first ? a.dx0 : a.dx1;
  const int Cd = first ? a.C0 : a.C1, od = first ? o : o - o0;
  const int acc = first ? a.acc0 : a.acc1;
  const int p_begin = blockIdx.x * a.apix_per_chunk, p_end = min(a.HW, p_begin + a.apix_per_chunk);
  auto emit = [&](const uint4& xr, const uint4& dr, const uint4& rr, uint4* dst) {
    float x[8], d[8], r[8];
    unpack8(xr, x); unpack8(dr, d);
    if (acc) unpack8(rr, r);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (x[k] - mu[k]) * rs[k];
      const float dz = a.silu ? d[k] * dsilu(ga[k] * xh + be[k]) : d[k];
      const float dx = rs[k] * (ga[k] * dz - m1[k] - xh * m2[k]);
      r[k] = acc ? r[k] + dx : dx;
    }
    *dst = pack8(r);
  };
  auto dst_of = [&](int p) { return (uint4*)(dbase + ((long)(b * a.HW + p) * Cd + od * 8)); };
  int p = p_begin + pl;
  for (; p + 3 * a.PL < p_end; p += 4 * a.PL) {      // 8-12 loads in flight per thread
    uint4 xr[4], dr[4], rr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xr[u] = *gnb_src(a, b, p + u * a.PL, o);
      dr[u] = *(const uint4*)(a.dy + ((long)(b * a.HW + p + u * a.PL) * a.C + o * 8));
      rr[u] = make_uint4(0u, 0u, 0u, 0u);
      if (acc) rr[u] = *dst_of(p + u * a.PL);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) emit(xr[u], dr[u], rr[u], dst_of(p + u * a.PL));
  }
  for (; p < p_end; p += a.PL) {
    uint4 rr = make_uint4(0u, 0u, 0u, 0u);
    if (acc) rr = *dst_of(p);
    emit(*gnb_src(a, b, p, o), *(const uint4*)(a.dy + ((long)(b * a.HW + p) * a.C + o * 8)), rr, dst_of(p));
  }
}

// LayerNorm backward: a workgroup owns 64 consecutive rows, one wave per row in turn; lanes own fixed columns so
// dgamma / dbeta accumulate in registers and leave the workgroup as one atomic per column.
template <int MAXO>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                            const float* __restrict__ gamma, bf16_t* __restrict__ dx,
                                                            int accumulate, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int M, int C, float eps) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][C][2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C8 = C >> 3;
  float gg[MAXO][8], dgam[MAXO][8], dbet[MAXO][8];
#pragma unroll
  for (int i = 0; i < MAXO; ++i) {
    const int o = lane + i * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) { gg[i][k] = o < C8 ? gamma[o * 8 + k] : 0.f; dgam[i][k] = 0.f; dbet[i][k] = 0.f; }
  }
  const int r_end = min(M, (int)(blockIdx.x + 1) * 64);
  // the NEXT row's x / dy (and dx when accumulating) are loaded before the current row is reduced: one row per iteration with its
  // loads at the top was a chain of dependent round trips (16 per wave)
  uint4 xn[MAXO], dn[MAXO], rn[MAXO];
  auto fetch = [&](int row) {
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      const int o = lane + i * 64;
      xn[i] = make_uint4(0u, 0u, 0u, 0u); dn[i] = xn[i]; rn[i] = xn[i];
      if (o < C8 && row < r_end) {
        xn[i] = *(const uint4*)(x + (long)row * C + o * 8);
        dn[i] = *(const uint4*)(dy + (long)row * C + o * 8);
        if (accumulate) rn[i] = *(const uint4*)(dx + (long)row * C + o * 8);
      }
    }
  };
  fetch(blockIdx.x * 64 + wave);
  for (int row = blockIdx.x * 64 + wave; row < r_end; row += 4) {
    float v[MAXO][8], d[MAXO][8];
    uint4 rcur[MAXO];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      unpack8(xn[i], v[i]); unpack8(dn[i], d[i]); rcur[i] = rn[i];
      if (lane + i * 64 < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[i][k];
      }
    }
    fetch(row + 4);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i)
      if (lane + i * 64 < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float t = v[i][k] - mean; q += t * t; }
      }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i)
      if (lane + i * 64 < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xh = (v[i][k] - mean) * rstd;
          v[i][k] = xh;
          dbet[i][k] += d[i][k]; dgam[i][k] += d[i][k] * xh;
          d[i][k] *= gg[i][k];
          s1 += d[i][k]; s2 += d[i][k] * xh;
        }
      }
    s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      const int o = lane + i * 64;
      if (o < C8) {
        float r[8];
        uint4* dst = (uint4*)(dx + (long)row * C + o * 8);
        if (accumulate) unpack8(rcur[i], r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float g = rstd * (d[i][k] - s1 - v[i][k] * s2);
          r[k] = accumulate ? r[k] + g : g;
        }
        *dst = pack8(r);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXO; ++i) {
    const int o = lane + i * 64;
    if (o < C8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(wave * C + o * 8 + k) * 2] = dgam[i][k];
        red[(wave * C + o * 8 + k) * 2 + 1] = dbet[i][k];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < 4; ++w) { a += red[(w * C + c) * 2]; b += red[(w * C + c) * 2 + 1]; }
    atomicAdd(dgamma + c, a);
    atomicAdd(dbeta + c, b);
  }
}

}  // namespace

namespace dfh {

int groupnorm_bwd_launch(GnBwdArgs a, hipStream_t stream) {
  a.C = a.C0 + a.C1;
  DFH_REQUIRE(a.C % 8 == 0 && a.C0 % 8 == 0 && a.C1 % 8 == 0, "channels must be multiples of 8");
  DFH_REQUIRE(a.G > 0 && a.G <= 64 && a.C % a.G == 0, "bad group count");
  DFH_REQUIRE(a.src0 && a.dy && a.stats && a.gamma && a.beta && a.dx0 && a.dgamma && a.dbeta && a.partial, "null pointer");
  DFH_REQUIRE(a.C1 == 0 || (a.src1 && a.dx1), "second source / gradient missing");
  const int C8 = a.C / 8;
  int PL = 256 / C8;
  if (PL < 1) PL = 1;
  if (PL > a.HW) PL = a.HW;
  a.PL = PL;
  const int block = ((C8 * PL + 63) / 64) * 64;
  DFH_REQUIRE(block <= 512, "block too large (more than 4096 channels)");
  const int max_by_pix = (a.HW + PL - 1) / PL;
  int chunks = std::max(1, std::min({(512 + a.B - 1) / a.B, max_by_pix, (int)GN_MAX_CHUNKS}));
  a.pix_per_chunk = (a.HW + chunks - 1) / chunks;
  a.chunks = (a.HW + a.pix_per_chunk - 1) / a.pix_per_chunk;
  int ac = std::max(1, std::min((2048 + a.B - 1) / a.B, max_by_pix));
  a.apix_per_chunk = (a.HW + ac - 1) / ac;
  const int achunks = (a.HW + a.apix_per_chunk - 1) / a.apix_per_chunk;
  const size_t lds = (size_t)a.PL * a.C * 2 * sizeof(float);
  DFH_REQUIRE(lds <= 64 * 1024, "GroupNorm LDS reduction too large");
  ProfScope ps(PC_NORM_BWD, 0.0, 10.0 * a.B * (double)a.HW * a.C, stream);   // x, dy read twice, dx written
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(a.chunks, a.B), dim3(block), lds, stream, a);
  if (int rc = check_launch("gn_bwd_stats_kernel")) return rc;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(achunks, a.B), dim3(block), 0, stream, a);
  return check_launch("gn_bwd_apply_kernel");
}

int layernorm_bwd_launch(const bf16_t* x, const bf16_t* dy, const float* gamma, bf16_t* dx, int accumulate, float* dgamma,
                         float* dbeta, int M, int C, float eps, hipStream_t stream) {
  DFH_REQUIRE(C % 8 == 0 && C <= 8 * 64 * 4, "LayerNorm width must be a multiple of 8 and <= 2048");
  const dim3 grid((M + 63) / 64), block(256);
  const size_t lds = (size_t)4 * C * 2 * sizeof(float);
  ProfScope ps(PC_NORM_BWD, 0.0, 6.0 * (double)M * C, stream);
  if (C <= 512) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, block, lds, stream, x, dy, gamma, dx, accumulate, dgamma, dbeta, M, C, eps);
  else if (C <= 1024) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, block, lds, stream, x, dy, gamma, dx, accumulate, dgamma, dbeta, M, C, eps);
  else hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, block, lds, stream, x, dy, gamma, dx, accumulate, dgamma, dbeta, M, C, eps);
  return check_launch("layernorm_bwd_kernel");
}

}  // namespace dfh
