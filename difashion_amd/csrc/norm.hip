// GroupNorm(32)(+SiLU) and LayerNorm for NHWC bf16 activations on gfx950.
// Reference call sites: diffusers ResnetBlock2D norm1/norm2 + SiLU, conv_norm_out, Transformer2DModel.norm
// (eps 1e-6, no activation) and BasicTransformerBlock norm1/2/3 reached through
// DiFashion/models/difashion.py:249-253,518-523.
//
// HBM-bound kernels (DESIGN.md "Kernels/norm"): every access is a 16-byte bf16x8 vector, each
// thread owns a fixed channel octet so per-channel scale/shift live in registers, statistics are
// reduced deterministically (fixed order, no float atomics): per-thread registers -> LDS ->
// 32 group lanes -> [B][chunk][32][2] partials -> summed in chunk order by the apply kernel.
// The channel concat of the up-path (torch.cat([h, skip], dim=1)) is fused into the read: an
// octet below C0/8 comes from src0, the rest from src1.
#include "dfh_common.h"
#include "norm.h"
#include <cstring>
#include <cstdlib>

namespace {

struct GnGeom {
  int C, C0, C1, HW, cpg, C8, PL, chunks, pix_per_chunk;
};

DFH_DEVICE const uint4* gn_src(const GnArgs& a, int b, int p, int o) {
  const int o0 = a.C0 >> 3;
  if (o < o0) return (const uint4*)(a.src0 + ((long)(b * a.HW + p) * a.C0 + o * 8));
  return (const uint4*)(a.src1 + ((long)(b * a.HW + p) * a.C1 + (o - o0) * 8));
}

// grid (chunks, B); block = roundup64(C8 * PL) threads; thread -> (pixel lane pl, octet o)
// (<= 512 threads: C <= 4096; without the bound the compiler budgets for 1024 threads, 128 VGPRs, and spills 28 of them inside the load loop)
__global__ __launch_bounds__(512) void gn_stats_kernel(const GnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [PL][C][2]
  const int C8 = a.C >> 3;
  const int tid = threadIdx.x;
  const int o = tid % C8, pl = tid / C8;
  const int b = blockIdx.y;
  const int p_begin = blockIdx.x * a.pix_per_chunk;
  const int p_end = min(a.HW, p_begin + a.pix_per_chunk);
  float s[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
  if (pl < a.PL) {
    // 4 independent 16-byte loads in flight per thread (a single dependent load per iteration is
    // latency-bound at ~1 TB/s); accumulation order stays p-ascending, so results are unchanged.
    int p = p_begin + pl;
    for (; p + 7 * a.PL < p_end; p += 8 * a.PL) {    // eight loads in flight
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *gn_src(a, b, p + u * a.PL, o);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float f[8];
        unpack8(v[u], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] += f[k] * f[k]; }
      }
    }
    for (; p + 3 * a.PL < p_end; p += 4 * a.PL) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *gn_src(a, b, p + u * a.PL, o);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8(v[u], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] += f[k] * f[k]; }
      }
    }
    for (; p < p_end; p += a.PL) {
      float f[8];
      unpack8(*gn_src(a, b, p, o), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] += f[k] * f[k]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[((pl * a.C) + o * 8 + k) * 2 + 0] = s[k];
      red[((pl * a.C) + o * 8 + k) * 2 + 1] = q[k];
    }
  }
  __syncthreads();
  if (tid < a.G) {
    const int cpg = a.C / a.G;
    float ss = 0.f, qq = 0.f;
    for (int l = 0; l < a.PL; ++l)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
        ss += red[(l * a.C + c) * 2 + 0];
        qq += red[(l * a.C + c) * 2 + 1];
      }
    // [b][group][chunk][2]: a group's partials are contiguous, the apply kernel reads them as float4s
    float* dst = a.partial + (((long)b * a.G + tid) * gridDim.x + blockIdx.x) * 2;
    dst[0] = ss; dst[1] = qq;
  }
}

// grid (achunks, B); same thread mapping; normalise (+SiLU) and write bf16 [B][HW][C].  The first GN_PRE pixels of a thread are
// loaded BEFORE the statistics are reduced: partials -> barrier -> gamma / beta -> data were three dependent memory round
// trips per block, and a block only lives for a handful of pixels per thread.
constexpr int GN_PRE = 6;
__global__ void gn_apply_kernel(const GnArgs a) {
  __shared__ float mean_s[64], rstd_s[64];
  const int C8 = a.C >> 3;
  const int tid = threadIdx.x;
  const int o = tid % C8, pl = tid / C8;
  const int b = blockIdx.y;
  const int cpg = a.C / a.G;
  const int p_begin = blockIdx.x * a.apix_per_chunk;
  const int p_end = min(a.HW, p_begin + a.apix_per_chunk);
  const bool live = pl < a.PL;
  uint4 pre[GN_PRE];
#pragma unroll
  for (int u = 0; u < GN_PRE; ++u) {
    const int p = p_begin + pl + u * a.PL;
    pre[u] = make_uint4(0u, 0u, 0u, 0u);
    if (live && p < p_end) pre[u] = *gn_src(a, b, p, o);
  }
  float gmr[8], btr[8];
  if (live && !a.out8) {
    const float4 g0 = *(const float4*)(a.gamma + o * 8), g1 = *(const float4*)(a.gamma + o * 8 + 4);
    const float4 b0 = *(const float4*)(a.beta + o * 8), b1 = *(const float4*)(a.beta + o * 8 + 4);
    gmr[0] = g0.x; gmr[1] = g0.y; gmr[2] = g0.z; gmr[3] = g0.w; gmr[4] = g1.x; gmr[5] = g1.y; gmr[6] = g1.z; gmr[7] = g1.w;
    btr[0] = b0.x; btr[1] = b0.y; btr[2] = b0.z; btr[3] = b0.w; btr[4] = b1.x; btr[5] = b1.y; btr[6] = b1.z; btr[7] = b1.w;
  }
  // statistics: the image's `chunks` partials per group, summed by eight threads per group (each a contiguous run of chunks in
  // ascending order, then the eight runs in ascending order: a fixed order).  Summed by one thread per group this prologue was a
  // chain of 8-16 dependent load batches per block and paced the whole kernel (37 -> 28 us on 64x64 x 320 with it gone).
  __shared__ float2 run_s[64][8];
  const bool wide_red = blockDim.x >= (unsigned)a.G * 8u;
  if (wide_red) {
    if (tid < a.G * 8) {
      const int g = tid >> 3, sub = tid & 7;
      const int per = (a.chunks + 7) >> 3, c0 = sub * per, c1 = min(a.chunks, c0 + per);
      const float2* src = (const float2*)(a.partial + ((long)b * a.G + g) * a.chunks * 2);
      float ss = 0.f, qq = 0.f;
      for (int c = c0; c < c1; ++c) { const float2 v = src[c]; ss += v.x; qq += v.y; }
      run_s[g][sub] = float2{ss, qq};
    }
    __syncthreads();
  }
  if (tid < a.G) {
    float ss = 0.f, qq = 0.f;
    if (wide_red) {
#pragma unroll
      for (int sub = 0; sub < 8; ++sub) { ss += run_s[tid][sub].x; qq += run_s[tid][sub].y; }
    } else {
      const float2* src = (const float2*)(a.partial + ((long)b * a.G + tid) * a.chunks * 2);      // contiguous (gn_stats_kernel)
      for (int c = 0; c < a.chunks; ++c) { ss += src[c].x; qq += src[c].y; }
    }
    const float n = (float)a.HW * (float)cpg;
    const float mean = ss / n;
    const float var = fmaxf(qq / n - mean * mean, 0.f);
    mean_s[tid] = mean;
    rstd_s[tid] = rsqrtf(var + a.eps);
    if (a.stats_out && blockIdx.x == 0) {
      a.stats_out[((long)b * a.G + tid) * 2 + 0] = mean;
      a.stats_out[((long)b * a.G + tid) * 2 + 1] = rstd_s[tid];
    }
  }
  __syncthreads();
  if (!live) return;
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = o * 8 + k;
    const int g = c / cpg;
    const float w = (a.out8 ? a.q_mul : gmr[k]) * rstd_s[g];
    sc[k] = w;
    sh[k] = (a.out8 ? 0.f : btr[k]) - mean_s[g] * w;
  }
  auto emit = [&](const uint4& raw, int p) {
    float f[8];
    unpack8(raw, f);
    if (a.out8) {                                    // e4m3 of the normalised value under the static scale (GnArgs::out8)
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = __builtin_amdgcn_fmed3f(f[k] * sc[k] + sh[k], -448.f, 448.f);
      uint2 w;
      w.x = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false), true);
      w.y = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false), true);
      *(uint2*)(a.out8 + ((long)(b * a.HW + p) * a.C + o * 8)) = w;
      return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float y = f[k] * sc[k] + sh[k];
      f[k] = a.silu ? silu_f(y) : y;
    }
    *(uint4*)(a.out + ((long)(b * a.HW + p) * a.C + o * 8)) = pack8(f);
  };
#pragma unroll
  for (int u = 0; u < GN_PRE; ++u) {
    const int p = p_begin + pl + u * a.PL;
    if (p < p_end) emit(pre[u], p);
  }
  int p = p_begin + pl + GN_PRE * a.PL;
  for (; p + 3 * a.PL < p_end; p += 4 * a.PL) {      // 4 loads in flight per thread
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *gn_src(a, b, p + u * a.PL, o);
#pragma unroll
    for (int u = 0; u < 4; ++u) emit(v[u], p + u * a.PL);
  }
  for (; p < p_end; p += a.PL) emit(*gn_src(a, b, p, o), p);
}

// one wave per R token rows; exact two-pass variance in registers (C <= 8*64*MAXO).  The loads of all R rows (and gamma / beta)
// are issued before anything is reduced: with one 640-byte row per wave the chip had ~5 MB in flight, half of what 5 TB/s
// needs at ~2 us of loaded latency.
template <int MAXO, int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                        int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  const int C8 = C >> 3;
  uint4 raw[R][MAXO];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      const int o = lane + i * 64;
      raw[r][i] = make_uint4(0u, 0u, 0u, 0u);
      if (o < C8 && row0 + r < M) raw[r][i] = *(const uint4*)(x + (long)(row0 + r) * C + o * 8);
    }
  float gg[MAXO][8], bb[MAXO][8];
#pragma unroll
  for (int i = 0; i < MAXO; ++i) {
    const int o = lane + i * 64;
    if (o < C8) {
      const float4 g0 = *(const float4*)(gamma + o * 8), g1 = *(const float4*)(gamma + o * 8 + 4);
      const float4 b0 = *(const float4*)(beta + o * 8), b1 = *(const float4*)(beta + o * 8 + 4);
      gg[i][0] = g0.x; gg[i][1] = g0.y; gg[i][2] = g0.z; gg[i][3] = g0.w; gg[i][4] = g1.x; gg[i][5] = g1.y; gg[i][6] = g1.z; gg[i][7] = g1.w;
      bb[i][0] = b0.x; bb[i][1] = b0.y; bb[i][2] = b0.z; bb[i][3] = b0.w; bb[i][4] = b1.x; bb[i][5] = b1.y; bb[i][6] = b1.z; bb[i][7] = b1.w;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= M) break;
    float v[MAXO][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      unpack8(raw[r][i], v[i]);                    // lanes past C8 hold zeros
      if (lane + i * 64 < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[i][k];
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      if (lane + i * 64 < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      const int o = lane + i * 64;
      if (o < C8) {
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (v[i][k] - mean) * rstd * gg[i][k] + bb[i][k];
        *(uint4*)(y + (long)(row0 + r) * C + o * 8) = pack8(f);
      }
    }
  }
}

}  // namespace

// ---- small tensors (16x16 / 8x8 levels): ONE launch.  A block owns one (batch, group) slab of
// HW x cpg <= 32768 elements, reads it once into registers (8-byte units = 4 channels of a pixel), reduces it in a fixed
// order (two-pass variance from the registers), normalises and writes.  The two-kernel path above spends ~2 x 8 us of launch
// and dependency latency on tensors that hold a few hundred KB.
template <int UNITS>
__global__ __launch_bounds__(256) void gn_small_kernel(const GnArgs a) {
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = a.C / a.G, upp = cpg >> 2, total = a.HW * upp;
  const int cg = g * cpg;                                     // first channel of the group in the concatenated tensor
  const bool first = cg < a.C0;
  const bf16_t* src = first ? a.src0 + (long)b * a.HW * a.C0 + cg : a.src1 + (long)b * a.HW * a.C1 + (cg - a.C0);
  const int ldc = first ? a.C0 : a.C1;
  __shared__ float red[8];
  // the group's gamma / beta go to LDS once: read per unit from global they were two more loads for every 8-byte data load
  __shared__ __attribute__((aligned(16))) float gam_s[256], bet_s[256];
  const bool gb_lds = cpg <= 256;
  if (gb_lds && tid < cpg && !a.out8) { gam_s[tid] = a.gamma[cg + tid]; bet_s[tid] = a.beta[cg + tid]; }
  float v[UNITS][4];
  int pix[UNITS], ch[UNITS];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int idx = tid + u * 256;
    pix[u] = -1; ch[u] = 0;
    v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
    if (idx < total) {
      const int p = idx / upp, j = idx - p * upp;
      pix[u] = p; ch[u] = j * 4;
      const uint2 r = *(const uint2*)(src + (long)p * ldc + j * 4);
      v[u][0] = __uint_as_float(r.x << 16); v[u][1] = __uint_as_float(r.x & 0xffff0000u);
      v[u][2] = __uint_as_float(r.y << 16); v[u][3] = __uint_as_float(r.y & 0xffff0000u);
      s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float n = (float)a.HW * (float)cpg;
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / n;
  float q = 0.f;
#pragma unroll
  for (int u = 0; u < UNITS; ++u)
    if (pix[u] >= 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[u][k] - mean; q += d * d; }
    }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / n + a.eps);
  if (a.stats_out && tid == 0) { a.stats_out[((long)b * a.G + g) * 2] = mean; a.stats_out[((long)b * a.G + g) * 2 + 1] = rstd; }
  if (a.out8) {                                      // e4m3 of the normalised value under the static scale (GnArgs::out8)
    uint8_t* dst8 = a.out8 + (long)b * a.HW * a.C + cg;
    const float w = rstd * a.q_mul;
#pragma unroll
    for (int u = 0; u < UNITS; ++u)
      if (pix[u] >= 0) {
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = __builtin_amdgcn_fmed3f((v[u][k] - mean) * w, -448.f, 448.f);
        *(unsigned*)(dst8 + (long)pix[u] * a.C + ch[u]) =
            (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], 0, false), true);
      }
    return;
  }
  bf16_t* dst = a.out + (long)b * a.HW * a.C + cg;
#pragma unroll
  for (int u = 0; u < UNITS; ++u)
    if (pix[u] >= 0) {
      const float4 gm = gb_lds ? *(const float4*)(gam_s + ch[u]) : *(const float4*)(a.gamma + cg + ch[u]);
      const float4 bt = gb_lds ? *(const float4*)(bet_s + ch[u]) : *(const float4*)(a.beta + cg + ch[u]);
      float y[4] = {(v[u][0] - mean) * rstd * gm.x + bt.x, (v[u][1] - mean) * rstd * gm.y + bt.y,
                    (v[u][2] - mean) * rstd * gm.z + bt.z, (v[u][3] - mean) * rstd * gm.w + bt.w};
      if (a.silu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = silu_f(y[k]);
      }
      uint2 o; o.x = pack2bf(y[0], y[1]); o.y = pack2bf(y[2], y[3]);
      *(uint2*)(dst + (long)pix[u] * a.C + ch[u]) = o;
    }
}

// ---- mid-size tensors (everything below the 64x64 level): ONE launch, coalesced.  gn_small_kernel above gives a block one
// (image, group) slab: cpg * 2 = 40..160 contiguous bytes per pixel at a stride of the whole channel count, so every 128-byte line is
// fetched by the 2-4 blocks whose groups share it (usually on different XCDs), and gamma / beta are re-read per 8-byte unit.
// Here a block owns GQ ADJACENT groups of one image (160-640 contiguous bytes per pixel), a thread owns a FIXED 4-channel unit
// of that span (gamma / beta / group index live in registers) and walks pixels; the slab stays in registers as packed bf16
// (2 VGPRs per unit), the statistics are reduced in a fixed order through LDS (two-pass variance), one read, one write.
//   threads = PPB pixel lanes x UPPB units per pixel (UPPB = GQ * cpg / 4), thread -> (pp = tid / UPPB, j = tid % UPPB)
//   MAXT: largest block the instantiation is launched with.  More than 16 units per thread (64+ VGPRs of slab plus as many load addresses
//   in flight) do not fit the 128 VGPRs of a 1024-thread block (the 24- / 32-unit variants spilled 22 / 72 registers): those run with at
//   most 512 threads (256 VGPRs), the launcher picks the geometry accordingly.
template <int UNITS, int MAXT = 1024>
__global__ __launch_bounds__(MAXT) void gn_mid_kernel(const GnArgs a, const int gq, const int ppb) {
  __shared__ float red[MAXT];
  __shared__ float colsum[2][256];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int cpg = a.C / a.G, upp = cpg >> 2, uppb = gq * upp;
  const int g0 = blockIdx.x * gq, cg0 = g0 * cpg;
  const bool active = tid < ppb * uppb;
  const int pp = tid / uppb, j = tid - pp * uppb;
  const int c = cg0 + j * 4;                                   // this thread's four channels in the concatenated tensor
  const bool first = c < a.C0;
  const bf16_t* src = first ? a.src0 + (long)b * a.HW * a.C0 + c : a.src1 + (long)b * a.HW * a.C1 + (c - a.C0);
  const int ldc = first ? a.C0 : a.C1;
  uint2 v[UNITS];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int p = pp + u * ppb;
    v[u] = uint2{0u, 0u};
    if (active && p < a.HW) v[u] = *(const uint2*)(src + (long)p * ldc);
  }
#pragma unroll
  for (int u = 0; u < UNITS; ++u)      // out-of-range units hold zeros: they add nothing here and are masked in the second pass
    s += (__uint_as_float(v[u].x << 16) + __uint_as_float(v[u].x & 0xffff0000u)) +
         (__uint_as_float(v[u].y << 16) + __uint_as_float(v[u].y & 0xffff0000u));
  const float n = (float)a.HW * (float)cpg;
  const int gi = j / upp;                                       // group of this thread inside the block
  auto group_total = [&](float x, int slot) -> float {
    red[tid] = active ? x : 0.f;
    __syncthreads();
    if (tid < uppb) {
      float t = 0.f;
      for (int q = 0; q < ppb; ++q) t += red[q * uppb + tid];
      colsum[slot][tid] = t;
    }
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < upp; ++k) t += colsum[slot][gi * upp + k];
    return t;
  };
  const float mean = group_total(s, 0) / n;
  float q = 0.f;
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    if (pp + u * ppb < a.HW) {
      const float d0 = __uint_as_float(v[u].x << 16) - mean, d1 = __uint_as_float(v[u].x & 0xffff0000u) - mean;
      const float d2 = __uint_as_float(v[u].y << 16) - mean, d3 = __uint_as_float(v[u].y & 0xffff0000u) - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = rsqrtf(group_total(q, 1) / n + a.eps);
  if (!active) return;
  if (a.stats_out && pp == 0 && j == gi * upp) {
    a.stats_out[((long)b * a.G + g0 + gi) * 2] = mean; a.stats_out[((long)b * a.G + g0 + gi) * 2 + 1] = rstd;
  }
  const float4 gm = *(const float4*)(a.gamma + c), bt = *(const float4*)(a.beta + c);
  const float w0 = rstd * gm.x, w1 = rstd * gm.y, w2 = rstd * gm.z, w3 = rstd * gm.w;
  bf16_t* dst = a.out + (long)b * a.HW * a.C + c;
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int p = pp + u * ppb;
    if (p < a.HW) {
      float y[4] = {(__uint_as_float(v[u].x << 16) - mean) * w0 + bt.x, (__uint_as_float(v[u].x & 0xffff0000u) - mean) * w1 + bt.y,
                    (__uint_as_float(v[u].y << 16) - mean) * w2 + bt.z, (__uint_as_float(v[u].y & 0xffff0000u) - mean) * w3 + bt.w};
      if (a.silu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = silu_f(y[k]);
      }
      uint2 o; o.x = pack2bf(y[0], y[1]); o.y = pack2bf(y[2], y[3]);
      *(uint2*)(dst + (long)p * a.C) = o;
    }
  }
}

namespace {
// grid (N / 8, B), 256 threads: the image's group statistics from the partials (as gn_apply_kernel), then eight rows of the projection,
// two per wave (a lane takes 8-column chunks lane, lane + 64, ...).
__global__ __launch_bounds__(256) void gn_fold_kernel(const GnFoldArgs a, const float* __restrict__ partial, int chunks) {
  __shared__ float mean_s[64], rstd_s[64];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int cpg = a.C / a.G;
  if (tid < a.G) {
    const float2* src = (const float2*)(partial + ((long)b * a.G + tid) * chunks * 2);
    float ss = 0.f, qq = 0.f;
    for (int c = 0; c < chunks; ++c) { ss += src[c].x; qq += src[c].y; }
    const float n = (float)a.HW * (float)cpg;
    const float mean = ss / n;
    const float var = fmaxf(qq / n - mean * mean, 0.f);
    mean_s[tid] = mean;
    rstd_s[tid] = rsqrtf(var + a.eps);
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int r = wave; r < 8; r += 4) {
    const int n = blockIdx.x * 8 + r;
    if (n >= a.N) break;
    float acc = 0.f;
    for (int ch = lane; ch < (a.C >> 3); ch += 64) {
      float w[8], o[8];
      unpack8(*(const uint4*)(a.W + (long)n * a.ldw + ch * 8), w);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = ch * 8 + k, g = c / cpg;
        o[k] = w[k] * a.gamma[c] * rstd_s[g];
      }
      const uint4 packed = pack8(o);
      float wr[8];
      unpack8(packed, wr);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = ch * 8 + k;
        acc += w[k] * a.beta[c] - wr[k] * mean_s[c / cpg];
      }
      *(uint4*)(a.Wimg + ((long)b * a.N + n) * a.C + ch * 8) = packed;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) a.rv[(long)b * a.N + n] = (a.bias ? a.bias[n] : 0.f) + acc;
  }
}
}  // namespace

namespace dfh {

static void gn_geometry(GnArgs& a, int* block, int* achunks) {
  const int C8 = a.C / 8;
  int PL = 256 / C8;
  if (PL < 1) PL = 1;
  if (PL > a.HW) PL = a.HW;
  a.PL = PL;
  *block = ((C8 * PL + 63) / 64) * 64;
  // statistics: enough chunks to fill the chip, at most GN_MAX_CHUNKS (partial buffer size)
  static const int env_sc = [] { const char* e = getenv("DFH_GN_SC"); return e ? atoi(e) : 512; }();    // probe knobs
  static const int env_ac = [] { const char* e = getenv("DFH_GN_AC"); return e ? atoi(e) : 1024; }();
  int chunks = (env_sc + a.B - 1) / a.B;
  const int max_by_pix = (a.HW + PL - 1) / PL;
  chunks = std::max(1, std::min({chunks, max_by_pix, (int)GN_MAX_CHUNKS}));
  a.pix_per_chunk = (a.HW + chunks - 1) / chunks;
  a.chunks = (a.HW + a.pix_per_chunk - 1) / a.pix_per_chunk;
  int ac = (env_ac + a.B - 1) / a.B;
  ac = std::max(1, std::min(ac, max_by_pix));
  a.apix_per_chunk = (a.HW + ac - 1) / ac;
  *achunks = (a.HW + a.apix_per_chunk - 1) / a.apix_per_chunk;
}

int groupnorm_launch(GnArgs a, hipStream_t stream) {
  a.C = a.C0 + a.C1;
  DFH_REQUIRE(a.C % 8 == 0 && a.C0 % 8 == 0 && a.C1 % 8 == 0, "channels must be multiples of 8");
  DFH_REQUIRE(a.G > 0 && a.G <= 64 && a.C % a.G == 0, "bad group count");
  // the statistics / apply kernels run one thread per 8 channels in blocks of at most 512 threads (their launch bounds): 4096 channels,
  // 3.2 x the widest tensor of the SD U-Nets (1280 + 1280 concatenated = 2560)
  DFH_REQUIRE(a.C <= 4096, "GroupNorm over more than 4096 channels is not supported");
  DFH_REQUIRE((a.partial != nullptr || a.pre != nullptr) && (a.out != nullptr || a.out8 != nullptr) && a.src0 != nullptr, "null pointer");
  if (a.out8) DFH_REQUIRE(a.C1 == 0 && !a.silu && a.q_mul > 0.f && !a.stats_out, "e4m3 GroupNorm output: one source, no SiLU, a positive scale");
  DFH_REQUIRE(a.C1 == 0 || a.src1 != nullptr, "second source missing");
  if (a.pre) {
    // the producer's epilogue already summed the tensor (one source, its own group structure): normalise only
    DFH_REQUIRE(a.C1 == 0 && a.pre_chunks > 0 && a.pre_chunks <= (int)GN_MAX_CHUNKS, "producer statistics: one source, 1..64 chunks");
    int block, achunks;
    gn_geometry(a, &block, &achunks);
    DFH_REQUIRE(block <= 512, "block too large (more than 4096 channels)");
    a.partial = const_cast<float*>(a.pre); a.chunks = a.pre_chunks;
    ProfScope ps(PC_GNORM, 0.0, (a.out8 ? 3.0 : 4.0) * a.B * (double)a.HW * a.C, stream);
    census(CK_GN_PRE);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(achunks, a.B), dim3(block), 0, stream, a);
    return check_launch("gn_apply_kernel");
  }
  {
    const int cpg = a.C / a.G;
    const long units = (long)a.HW * (cpg >> 2);
    const bool one_source_per_group = a.C1 == 0 || a.C0 % cpg == 0;
    static const int small_max = [] { const char* e = getenv("DFH_GN_SMALL_MAX"); return e ? atoi(e) : 16; }();   // probe knob; > 16 units per thread (32x32 x 640) the two-kernel path is faster since its prologue fix: 25.6 -> 21.3 us
    if ((cpg & 3) == 0 && units <= 256 * small_max && one_source_per_group && (long)a.B * a.G >= 64) {
      ProfScope ps(PC_GNORM, 0.0, (a.out8 ? 3.0 : 4.0) * a.B * (double)a.HW * a.C, stream);
      const dim3 grid(a.G, a.B);
      census(CK_GN_SMALL);
      if (units <= 256 * 8) hipLaunchKernelGGL(gn_small_kernel<8>, grid, dim3(256), 0, stream, a);
      else if (units <= 256 * 16) hipLaunchKernelGGL(gn_small_kernel<16>, grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL(gn_small_kernel<32>, grid, dim3(256), 0, stream, a);
      return check_launch("gn_small_kernel");
    }
  }
  {
    // mid path (16x16 / 8x8 tensors the one-slab kernel cannot take: groups that straddle the two concat sources, e.g. 1280 + 640
    // channels): GQ adjacent groups per block.  Measured against the alternatives (scripts/norm_microbench.py): 24.4 -> 13.4 us
    // on 16x16 x (1280 + 640); where gn_small_kernel is eligible it is faster (its blocks are four waves, the reductions here
    // run across up to sixteen), and at 32x32 the two-kernel path is, so this is the fallback between them.
    static const bool mid_off = [] { const char* e = getenv("DFH_GN_MID"); return e && e[0] == '0'; }();
    const int cpg = a.C / a.G, upp = cpg >> 2;
    if (!mid_off && !a.out8 && (cpg & 3) == 0 && a.HW <= 256 && a.G % 2 == 0) {
      for (int gq = 4; gq >= 2; gq >>= 1) {
        if (a.G % gq || (long)a.B * (a.G / gq) < 128 || gq * upp > 256) continue;
        const int uppb = gq * upp;
        int threads = 256;
        auto units_at = [&](int t) { return t / uppb < 1 ? 1 << 30 : (a.HW + t / uppb - 1) / (t / uppb); };
        while (threads < 1024 && units_at(threads) > 24) threads *= 2;
        // 1024-thread blocks hold at most 16 units per thread in registers (128 VGPRs); 17..32 units run as 512-thread blocks
        if (threads == 1024 && units_at(1024) > 16) threads = 512;
        const int ppb = threads / uppb;
        if (ppb < 1) continue;
        const int units = (a.HW + ppb - 1) / ppb;
        if (units > 32) continue;
        ProfScope ps(PC_GNORM, 0.0, 4.0 * a.B * (double)a.HW * a.C, stream);
        const dim3 grid(a.G / gq, a.B), block(threads);
        census(CK_GN_MID);
        if (units <= 4) hipLaunchKernelGGL(gn_mid_kernel<4>, grid, block, 0, stream, a, gq, ppb);
        else if (units <= 8) hipLaunchKernelGGL(gn_mid_kernel<8>, grid, block, 0, stream, a, gq, ppb);
        else if (units <= 16) hipLaunchKernelGGL(gn_mid_kernel<16>, grid, block, 0, stream, a, gq, ppb);
        else if (units <= 24) hipLaunchKernelGGL((gn_mid_kernel<24, 512>), grid, block, 0, stream, a, gq, ppb);      // threads <= 512 here
        else hipLaunchKernelGGL((gn_mid_kernel<32, 512>), grid, block, 0, stream, a, gq, ppb);
        return check_launch("gn_mid_kernel");
      }
    }
  }
  int block, achunks;
  gn_geometry(a, &block, &achunks);
  DFH_REQUIRE(block <= 512, "block too large (more than 4096 channels)");
  const size_t lds = (size_t)a.PL * a.C * 2 * sizeof(float);
  DFH_REQUIRE(lds <= 64 * 1024, "GroupNorm LDS reduction too large");
  ProfScope ps(PC_GNORM, 0.0, (a.out8 ? 3.0 : 4.0) * a.B * (double)a.HW * a.C, stream);   // algorithmic: one read + one write (bf16)
  census(CK_GN_STATS);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(a.chunks, a.B), dim3(block), lds, stream, a);
  if (int rc = check_launch("gn_stats_kernel")) return rc;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(achunks, a.B), dim3(block), 0, stream, a);
  return check_launch("gn_apply_kernel");
}

int groupnorm_fold_launch(GnFoldArgs f, hipStream_t stream) {
  DFH_REQUIRE(f.x && f.gamma && f.beta && f.W && f.Wimg && f.rv && (f.pre || f.partial), "groupnorm fold: null pointer");
  DFH_REQUIRE(f.C % 8 == 0 && f.G > 0 && f.G <= 64 && f.C % f.G == 0 && f.C <= 4096 && f.ldw % 8 == 0 && f.ldw >= f.C && f.N > 0, "groupnorm fold: bad shape");
  const float* partial = f.pre;
  int chunks = f.pre_chunks;
  if (!partial) {
    GnArgs a; std::memset(&a, 0, sizeof(a));
    a.src0 = f.x; a.C0 = a.C = f.C; a.B = f.B; a.HW = f.HW; a.G = f.G; a.partial = f.partial;
    int block, achunks;
    gn_geometry(a, &block, &achunks);
    DFH_REQUIRE(block <= 512, "block too large (more than 4096 channels)");
    const size_t lds = (size_t)a.PL * a.C * 2 * sizeof(float);
    DFH_REQUIRE(lds <= 64 * 1024, "GroupNorm LDS reduction too large");
    ProfScope ps(PC_GNORM, 0.0, 2.0 * a.B * (double)a.HW * a.C, stream);        // one read of the tensor
    census(CK_GN_STATS);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a.chunks, a.B), dim3(block), lds, stream, a);
    if (int rc = check_launch("gn_stats_kernel")) return rc;
    partial = f.partial; chunks = a.chunks;
  } else {
    DFH_REQUIRE(chunks > 0 && chunks <= (int)GN_MAX_CHUNKS, "producer statistics: 1..64 chunks");
  }
  ProfScope ps(PC_GNORM, 0.0, (double)f.N * f.C * 2.0 * (1.0 + f.B), stream);
  census(CK_GN_FOLDED);
  hipLaunchKernelGGL(gn_fold_kernel, dim3((f.N + 7) / 8, f.B), dim3(256), 0, stream, f, partial, chunks);
  return check_launch("gn_fold_kernel");
}

int layernorm_launch(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int M, int C, float eps,
                     hipStream_t stream) {
  DFH_REQUIRE(C % 8 == 0 && C <= 8 * 64 * 4, "LayerNorm width must be a multiple of 8 and <= 2048");
  const dim3 block(256);
  ProfScope ps(PC_LNORM, 0.0, 4.0 * (double)M * C, stream);
  census(CK_LAYERNORM);
  if (C <= 512) hipLaunchKernelGGL((layernorm_kernel<1, 4>), dim3((M + 15) / 16), block, 0, stream, x, gamma, beta, y, M, C, eps);
  else if (C <= 1024) hipLaunchKernelGGL((layernorm_kernel<2, 2>), dim3((M + 7) / 8), block, 0, stream, x, gamma, beta, y, M, C, eps);
  else hipLaunchKernelGGL((layernorm_kernel<4, 1>), dim3((M + 3) / 4), block, 0, stream, x, gamma, beta, y, M, C, eps);
  return check_launch("layernorm_kernel");
}

}  // namespace dfh
