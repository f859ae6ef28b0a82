// bf16 implicit-GEMM on MFMA for gfx950: every conv3x3 / 1x1 conv / linear on the DiFashion
// U-Net path (reference call sites: DiFashion/models/difashion.py:249-253,518-523 -> diffusers
// ResnetBlock2D / Transformer2DModel / Attention projections / GEGLU feed-forward).
//
// Design (MI355X-first, see DESIGN.md "Kernels/gemm"):
//   * activations are NHWC, so a 3x3 conv is a GEMM whose A rows are gathered per tap: nine
//     K-segments of Cin, each tap shifting the pixel the row reads (zero page for padding).
//     stride-2 (downsample), nearest-2x upsample and the skip-concat are pure address arithmetic.
//   * both operands are K-contiguous ([M][K] pixels, [N][K] weights) -> identical staging and
//     identical ds_read_b128 fragment reads for A and B of v_mfma_f32_16x16x32_bf16.
//   * global -> LDS by global_load_lds_dwordx4 (no VGPR round trip); LDS image is lane-linear,
//     bank conflicts are removed by XOR-swizzling the 16-byte slot on the SOURCE address and on
//     the fragment read (guide rule 21): slot' = slot ^ ((row >> 1) & 7) for 128-byte rows.
//   * the weight operand is fed as MFMA "A" so each lane ends up with 4 consecutive output
//     channels of one pixel -> 8-byte packed bf16 stores and float4 bias loads.
//   * 64-wide wavefronts, 4 waves as 2x2, tile 128 x {160,128,64}; 72 KiB LDS (double buffer)
//     -> 2 workgroups per CU; XCD-aware tile order; split-K through fp32 slabs + reduce kernel
//     for the 8x8 / 16x16 levels whose M is too small to fill 256 CUs.
#include "gemm.h"

#include <algorithm>
#include <string>

namespace {

constexpr int BK = 64;            // k-step depth (bf16 elements) = one 128-byte LDS row
constexpr int NW = 4;             // waves per workgroup
constexpr int NT = NW * 64;

struct KIter {                    // which 64-deep slice of which K segment a k-step covers
  int seg;                        // 0..ntaps-1 conv taps, then plain segments
  int c0;                         // channel offset inside the segment
  int wcol;                       // column of W where this slice starts
  int seglen;                     // channels in the current segment
};

DFH_DEVICE int cdiv64(int x) { return (x + BK - 1) / BK; }

DFH_DEVICE int seg_len(const GemmArgs& a, int seg) {
  return seg < a.ntaps ? a.conv_c : a.p_c[seg - a.ntaps];
}

DFH_DEVICE KIter kiter_at(const GemmArgs& a, int kstep) {
  KIter it;
  int seg = 0, base = 0;
  const int nseg = a.ntaps + a.nplain;
  for (;;) {
    const int len = seg_len(a, seg);
    const int n = cdiv64(len);
    if (kstep < n || seg == nseg - 1) { it.seglen = len; break; }
    kstep -= n; base += len; ++seg;
  }
  it.seg = seg; it.c0 = kstep * BK; it.wcol = base + it.c0;
  return it;
}

DFH_DEVICE void kiter_next(const GemmArgs& a, KIter& it) {
  it.c0 += BK; it.wcol += BK;
  if (it.c0 >= it.seglen) {
    it.wcol -= it.c0 - it.seglen;        // next segment starts right after this one's real length
    it.c0 = 0; ++it.seg;
    if (it.seg < a.ntaps + a.nplain) it.seglen = seg_len(a, it.seg);
  }
}

// ---------------------------------------------------------------------------------------------
template <int BM, int BN, bool GLDS>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_kernel(const GemmArgs a) {
  constexpr int WM = 2, WN = 2;
  constexpr int TM = BM / WM, TN = BN / WN;       // per-wave output tile
  constexpr int FM = TM / 16, FN = TN / 16;       // 16x16 fragments per wave
  constexpr int IA = BM / (8 * NW), IB = BN / (8 * NW);  // 1-KiB staging pieces per wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = (a.N + BN - 1) / BN;
  const int ntm = (a.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  // k-step range of this split
  const int per = (a.ksteps + a.ksplit - 1) / a.ksplit;
  const int ks_begin = blockIdx.z * per;
  const int ks_end = min(a.ksteps, ks_begin + per);
  const int nk = ks_end - ks_begin;

  // ---- per-thread staging bookkeeping: piece i of this wave covers tile rows (i*NW+wave)*8 + lane/8,
  //      16-byte slot lane%8; the source slot is swizzled, the LDS image stays lane-linear.
  const int srow = lane >> 3;
  const int sslot = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));   // == slot ^ ((row>>1)&7)
  int a_pix[IA], a_y[IA], a_x[IA];            // conv: batch pixel base, (oy*stride-1, ox*stride-1); plain: row
  const int HWo = a.Hout * a.Wout;
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int m = m0 + (i * NW + wave) * 8 + srow;
    if (m < a.M) {
      a_pix[i] = m;
      if (a.ntaps) {
        const int b = m / HWo, rem = m - b * HWo;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        a_y[i] = oy * a.stride - 1; a_x[i] = ox * a.stride - 1;   // plain segments still index by output row m
      } else { a_y[i] = 0; a_x[i] = 0; }
    } else { a_pix[i] = -1; a_y[i] = 0; a_x[i] = 0; }
  }
  int a_bbase[IA];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    a_bbase[i] = 0;
    if (a.ntaps && a_pix[i] >= 0) a_bbase[i] = (a_pix[i] / HWo) * a.Hin * a.Win;
  }
  long w_row[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int n = n0 + (i * NW + wave) * 8 + srow;
    w_row[i] = (n < a.N) ? (long)n * a.ldw : -1;
  }
  const int Hv = a.ups ? a.Hin * 2 : a.Hin, Wv = a.ups ? a.Win * 2 : a.Win;  // virtual (upsampled) input dims

  uint4 stage_regs[GLDS ? 1 : IA + IB];

  auto issue_stage = [&](const KIter& it, int buf) {
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + A_BYTES;
    const int ch = it.c0 + sslot * 8;                 // channel of this lane's 16-byte chunk
    const bool kin = ch < it.seglen;
    int ky = 0, kx = 0;
    if (it.seg < a.ntaps) { ky = it.seg / 3; kx = it.seg - ky * 3; }
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const bf16_t* src = a.zero;
      if (kin && a_pix[i] >= 0) {
        if (it.seg < a.ntaps) {
          const int yy = a_y[i] + ky, xx = a_x[i] + kx;
          if ((unsigned)yy < (unsigned)Hv && (unsigned)xx < (unsigned)Wv) {
            const int sy = a.ups ? (yy >> 1) : yy, sx = a.ups ? (xx >> 1) : xx;
            src = a.conv_src + ((long)(a_bbase[i] + sy * a.Win + sx) * a.conv_c + ch);
          }
        } else {
          const int ps = it.seg - a.ntaps;
          src = a.p_src[ps] + ((long)a_pix[i] * a.p_c[ps] + ch);
        }
      }
      if constexpr (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(As + (i * NW + wave) * 1024),
                                         16, 0, 0);
      } else {
        stage_regs[i] = *(const uint4*)src;
      }
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const bf16_t* src = a.zero;
      if (kin && w_row[i] >= 0) src = a.W + (w_row[i] + it.wcol + sslot * 8);
      if constexpr (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(Bs + (i * NW + wave) * 1024),
                                         16, 0, 0);
      } else {
        stage_regs[IA + i] = *(const uint4*)src;
      }
    }
  };
  auto commit_stage = [&](int buf) {   // register-staged variant only: write the pieces to LDS
    if constexpr (!GLDS) {
      unsigned char* As = smem + buf * STAGE;
      unsigned char* Bs = As + A_BYTES;
#pragma unroll
      for (int i = 0; i < IA; ++i) *(uint4*)(As + (i * NW + wave) * 1024 + lane * 16) = stage_regs[i];
#pragma unroll
      for (int i = 0; i < IB; ++i) *(uint4*)(Bs + (i * NW + wave) * 1024 + lane * 16) = stage_regs[IA + i];
    }
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;

  if (nk > 0) {
    KIter it = kiter_at(a, ks_begin);
    issue_stage(it, 0);
    commit_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
      const bool more = (t + 1 < nk);
      if (more) { kiter_next(a, it); issue_stage(it, cur ^ 1); }
      const unsigned char* As = smem + cur * STAGE;
      const unsigned char* Bs = As + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm * TM + i * 16 + fr;
          af[i] = *(const bf16x8_t*)(As + row * 128 + (((ks * 4 + fg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wn * TN + j * 16 + fr;
          bfr[j] = *(const bf16x8_t*)(Bs + row * 128 + (((ks * 4 + fg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            // weights as MFMA-A: D[row = channel (fg*4+r)][col = pixel (fr)]
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      if (more) commit_stage(cur ^ 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---------------------------------------------------------------- epilogue
  const bool partial = a.ksplit > 1;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * TM + i * 16 + fr;
    if (m >= a.M) continue;
    const int b = (a.rowvec || a.out_mode == OUT_BF16_T || a.out_mode == OUT_F32_T) ? m / a.rows_per_b : 0;
    if (a.act == ACT_GEGLU && !partial) {
#pragma unroll
      for (int j = 0; j + 1 < FN; j += 2) {
        const int n = n0 + wn * TN + j * 16 + fg * 4;      // packed column of the "value" half
        if (n >= a.N) continue;
        const float4 bv = a.bias ? *(const float4*)(a.bias + n) : float4{0, 0, 0, 0};
        const float4 bg = a.bias ? *(const float4*)(a.bias + n + 16) : float4{0, 0, 0, 0};
        float v[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
        float g[4] = {acc[i][j + 1][0] + bg.x, acc[i][j + 1][1] + bg.y, acc[i][j + 1][2] + bg.z, acc[i][j + 1][3] + bg.w};
        const int oc = ((n0 + wn * TN) >> 1) + (j >> 1) * 16 + fg * 4;
        uint2 o;
        o.x = pack2bf(v[0] * gelu_erf_f(g[0]), v[1] * gelu_erf_f(g[1]));
        o.y = pack2bf(v[2] * gelu_erf_f(g[2]), v[3] * gelu_erf_f(g[3]));
        *(uint2*)((bf16_t*)a.out + (long)m * a.ld_out + oc) = o;
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * TN + j * 16 + fg * 4;
      if (n >= a.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (partial) {
        *(float4*)(a.partial + ((long)blockIdx.z * a.M + m) * a.N + n) = float4{v[0], v[1], v[2], v[3]};
        continue;
      }
      if (a.bias) {
        const float4 bv = *(const float4*)(a.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (a.rowvec) {
        const float4 rv = *(const float4*)(a.rowvec + (long)b * a.rv_ld + a.rv_off + n);
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
      }
      if (a.act == ACT_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
      } else if (a.act == ACT_LEAKY) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r];
      } else if (a.act == ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
      }
      if (a.resid) {
        const uint2 rr = *(const uint2*)(a.resid + (long)m * a.ld_res + n);
        v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
        v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
      }
      if (a.out_mode == OUT_BF16) {
        uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
        *(uint2*)((bf16_t*)a.out + (long)m * a.ld_out + n) = o;
      } else if (a.out_mode == OUT_F32) {
        *(float4*)((float*)a.out + (long)m * a.ld_out + n) = float4{v[0], v[1], v[2], v[3]};
      } else if (a.out_mode == OUT_BF16_T) {
        const int mm = m - b * a.rows_per_b;
        bf16_t* o = (bf16_t*)a.out + ((long)b * a.N + n) * a.ld_out + mm;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(long)r * a.ld_out] = f2bf(v[r]);
      } else {  // OUT_F32_T
        const int mm = m - b * a.rows_per_b;
        float* o = (float*)a.out + ((long)b * a.N + n) * a.ld_out + mm;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(long)r * a.ld_out] = v[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// split-K second pass: sum the fp32 slabs in a fixed order (deterministic) and run the epilogue.
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const GemmArgs a) {
  const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long total = (long)a.M * a.N;
  if (idx >= total) return;
  const int m = (int)(idx / a.N), n = (int)(idx - (long)m * a.N);
  float4 s = *(const float4*)(a.partial + idx);
  for (int z = 1; z < a.ksplit; ++z) {
    const float4 p = *(const float4*)(a.partial + (long)z * total + idx);
    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
  }
  float v[4] = {s.x, s.y, s.z, s.w};
  const int b = (a.rowvec || a.out_mode == OUT_BF16_T || a.out_mode == OUT_F32_T) ? m / a.rows_per_b : 0;
  if (a.bias) {
    const float4 bv = *(const float4*)(a.bias + n);
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  if (a.rowvec) {
    const float4 rv = *(const float4*)(a.rowvec + (long)b * a.rv_ld + a.rv_off + n);
    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
  }
  if (a.act == ACT_SILU) { for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]); }
  else if (a.act == ACT_LEAKY) { for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r]; }
  else if (a.act == ACT_TANH) { for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]); }
  if (a.resid) {
    const uint2 rr = *(const uint2*)(a.resid + (long)m * a.ld_res + n);
    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
  }
  if (a.out_mode == OUT_BF16) {
    uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
    *(uint2*)((bf16_t*)a.out + (long)m * a.ld_out + n) = o;
  } else if (a.out_mode == OUT_F32) {
    *(float4*)((float*)a.out + (long)m * a.ld_out + n) = float4{v[0], v[1], v[2], v[3]};
  } else if (a.out_mode == OUT_BF16_T) {
    const int mm = m - b * a.rows_per_b;
    bf16_t* o = (bf16_t*)a.out + ((long)b * a.N + n) * a.ld_out + mm;
    for (int r = 0; r < 4; ++r) o[(long)r * a.ld_out] = f2bf(v[r]);
  } else {
    const int mm = m - b * a.rows_per_b;
    float* o = (float*)a.out + ((long)b * a.N + n) * a.ld_out + mm;
    for (int r = 0; r < 4; ++r) o[(long)r * a.ld_out] = v[r];
  }
}

template <int BM, int BN, bool GLDS>
int launch_tile(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN, GLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, 1, a.ksplit);
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, GLDS>), grid, dim3(NT), lds, stream, a);
  return dfh::check_launch("gemm_bf16_kernel");
}

}  // namespace

namespace dfh {

int gemm_count_ksteps(const GemmArgs& a) {
  int n = a.ntaps * ((a.conv_c + BK - 1) / BK);
  for (int i = 0; i < a.nplain; ++i) n += (a.p_c[i] + BK - 1) / BK;
  return n;
}

// tile ids: 0 = 128x160, 1 = 128x128, 2 = 128x64
int gemm_pick_split(const GemmArgs& a, int* tile_out) {
  int tile;
  if (a.act == ACT_GEGLU) tile = 1;
  else if (a.N % 160 == 0) tile = 0;
  else if (a.N % 128 == 0 || a.N > 320) tile = 1;
  else if (a.N <= 64) tile = 2;
  else tile = (a.N <= 128) ? 1 : 0;
  const int bn = tile == 0 ? 160 : (tile == 1 ? 128 : 64);
  const int blocks = ((a.M + 127) / 128) * ((a.N + bn - 1) / bn);
  const int ksteps = gemm_count_ksteps(a);
  int split = 1;
  if (a.act != ACT_GEGLU && blocks < 192 && ksteps >= 16) {
    split = std::min({(512 + blocks - 1) / blocks, ksteps / 8, 64});
    if (split < 1) split = 1;
  }
  if (tile_out) *tile_out = tile;
  return split;
}

size_t gemm_partial_floats(const GemmArgs& a) {
  int tile;
  const int s = gemm_pick_split(a, &tile);
  return s > 1 ? (size_t)s * a.M * a.N : 0;
}

int gemm_launch(GemmArgs a, hipStream_t stream, int force_tile, int force_split, int force_glds) {
  DFH_REQUIRE(a.M > 0 && a.N > 0, "empty GEMM");
  DFH_REQUIRE(a.N % 4 == 0, "N must be a multiple of 4");
  DFH_REQUIRE(a.ntaps == 0 || a.ntaps == 9, "ntaps must be 0 or 9");
  DFH_REQUIRE(a.ntaps + a.nplain >= 1, "no K segment");
  DFH_REQUIRE(a.ntaps == 0 || a.conv_c % 8 == 0, "conv channels must be a multiple of 8");
  for (int i = 0; i < a.nplain; ++i) DFH_REQUIRE(a.p_c[i] % 8 == 0, "segment length must be a multiple of 8");
  DFH_REQUIRE(a.zero != nullptr, "zero page missing");
  if (a.rows_per_b <= 0) a.rows_per_b = a.M;
  a.ksteps = gemm_count_ksteps(a);
  int tile;
  int split = gemm_pick_split(a, &tile);
  if (force_tile > 0) tile = force_tile - 1;
  if (force_split > 0) split = force_split;
  if (a.act == ACT_GEGLU) {
    DFH_REQUIRE(a.N % 32 == 0 && tile != 0 && split == 1, "GEGLU needs N % 32 == 0, a 2^k tile and no split-K");
    DFH_REQUIRE(a.out_mode == OUT_BF16 && !a.resid && !a.rowvec, "GEGLU epilogue is bias-only, bf16 out");
  }
  split = std::min(split, a.ksteps);
  a.ksplit = split;
  if (split > 1) DFH_REQUIRE(a.partial != nullptr, "split-K needs a partial buffer");
  const bool glds = force_glds < 0 ? true : (force_glds != 0);
  int rc;
  {
  // algorithmic work of this launch: 2*M*N*K over the REAL K (padding excluded); bytes = each operand once + output
  double kreal = (double)a.ntaps * a.conv_c;
  double abytes = a.ntaps ? (double)(a.M / (a.Hout * a.Wout)) * a.Hin * a.Win * a.conv_c * 2.0 : 0.0;
  for (int i = 0; i < a.nplain; ++i) { kreal += a.p_c[i]; abytes += (double)a.M * a.p_c[i] * 2.0; }
  const double obytes = (double)a.M * (a.act == ACT_GEGLU ? a.N / 2 : a.N) * ((a.out_mode == OUT_F32 || a.out_mode == OUT_F32_T) ? 4.0 : 2.0);
  ProfScope ps(a.ntaps ? PC_CONV3 : PC_LINEAR, 2.0 * a.M * a.N * kreal, abytes + (double)a.N * kreal * 2.0 + obytes, stream);
  if (glds) {
    rc = tile == 0 ? launch_tile<128, 160, true>(a, stream)
       : tile == 1 ? launch_tile<128, 128, true>(a, stream)
                   : launch_tile<128, 64, true>(a, stream);
  } else {
    rc = tile == 0 ? launch_tile<128, 160, false>(a, stream)
       : tile == 1 ? launch_tile<128, 128, false>(a, stream)
                   : launch_tile<128, 64, false>(a, stream);
  }
  }
  if (rc) return rc;
  if (split > 1) {
    const long total4 = ((long)a.M * a.N) / 4;
    ProfScope ps(PC_SPLITK, 0.0, (double)split * a.M * a.N * 4.0 + (double)a.M * a.N * 2.0, stream);
    hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, a);
    return check_launch("gemm_splitk_reduce");
  }
  return 0;
}

}  // namespace dfh
