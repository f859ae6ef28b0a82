// Wave-specialised implicit GEMM (same GemmArgs / K-segment semantics as gemm.hip): loader waves and matrix waves.
//
// Why: in gemm.hip / gemm_wide.hip every wave both stages and multiplies.  Issuing one 1-KiB LDS-DMA piece costs a wave ~100
// cycles during which it issues no MFMA, and a 256 x 160 tile needs one piece per 6 MFMAs (16x16x32) -- the staging issue
// time of a wave is as long as its matrix time, which is why those kernels sit at ~40 % matrix-pipe busy (profiles/r01).  The
// co-issue probe (scripts/probes/coissue.hip, profiles/r02/coissue.txt) shows what the hardware does allow: a wave that issues
// NOTHING but MFMAs keeps its SIMD's matrix pipe saturated (32.0 cycles per v_mfma_f32_32x32x16_bf16) while a second wave on
// the same SIMD does other work, unaffected.  So:
//   * 8 waves = 512 threads, ONE workgroup per CU: waves 0-3 ("matrix", one per SIMD) do ds_read_b128 + MFMA and nothing else
//     in the k-loop; waves 4-7 ("loader", one per SIMD) do all the addressing and the global -> LDS DMA of both operands;
//   * 256 x BN tile (BN = 160 or 128), a matrix wave owns 64 pixels x BN channels = 2 x (BN / 32) accumulator blocks of
//     v_mfma_f32_32x32x16_bf16 (1024 flop / cycle against the 820 of the 16x16x32 form -- profiles/r02/mfma_rate2.txt);
//   * 64-deep k-steps (128-byte rows: the fast LDS-DMA gather), 3 stages of 52 KiB = 156 KiB of the CU's 160 KiB LDS;
//     16-byte slots XOR-swizzled by (row >> 1) & 7 on the SOURCE address and the fragment read (conflict-free for the 32-row
//     fragments, derivation in the comment of the read below);
//   * one raw s_barrier per k-step for all eight waves: a loader waits (counted vmcnt) until its pieces of stage t have landed,
//     meets the matrix waves at the barrier -- which also tells it that stage t-1 has been consumed -- and refills that buffer;
//   * weights are the MFMA A operand, so a lane ends with 4 consecutive output channels of one pixel; epilogue as in
//     gemm_wide.hip: 64-row passes through an fp32 LDS tile (bias / time-embedding row / SiLU / residual on full rows, 16-byte
//     stores), GEGLU on the (value, gate) halves of each 32-row accumulator block in registers.
// Used for bf16-output launches without split-K whose grid gives every CU a tile (gemm_ws_pick).
#include "gemm.h"
#include "gemm_kiter.h"

#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int N> DFH_DEVICE void ws_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int WS_BM = 256, WS_NST = 3;

template <int BN>
__global__ __launch_bounds__(512, 2) void gemm_ws_kernel(const GemmArgs a) {
  constexpr int BM = WS_BM, NST = WS_NST;
  constexpr int NCB = BN / 32;                       // 32-channel accumulator blocks per matrix wave
  constexpr int PA = BM / 8, PB = BN / 8;            // 1-KiB staging pieces (8 rows x 128 B) per stage
  constexpr int IA = PA / 4, IB = PB / 4;            // per loader wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NPW = IA + IB;                       // LDS-DMA pieces per loader wave and stage
  static_assert(PA % 4 == 0 && PB % 4 == 0 && BN % 32 == 0, "tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int rw = wave & 3;                           // index inside the role
  const int ql = lane & 31, kh = lane >> 5;

  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  int mt_, nt_;
  tile_coords(blockIdx.x, ntm, ntn, a.n_major, a.tm_xm, a.tm_gm, mt_, nt_);
  const int m0 = mt_ * BM, n0 = nt_ * BN;
  const int nk = a.ksteps;

  f32x16_t acc[NCB][2];
#pragma unroll
  for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
    for (int pj = 0; pj < 2; ++pj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ci][pj][r] = 0.f;

  if (loader) {
    // ================================================================= loader waves: addressing + LDS-DMA, nothing else
    // piece p = i * 4 + rw covers tile rows p*8 + lane/8, 16-byte slot lane%8; the source slot is swizzled, the LDS image stays
    // lane-linear: slot s of row r holds chunk s ^ ((r >> 1) & 7); (r >> 1) & 7 = ((p & 1) << 2) | (srow >> 1), p & 1 = rw & 1
    const int srow = lane >> 3;
    const int sslot = (lane & 7) ^ (((rw & 1) << 2) | (lane >> 4));
    int a_pix[IA], a_y[IA], a_x[IA], a_bbase[IA];
    const int HWo = a.Hout * a.Wout;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int m = m0 + (i * 4 + rw) * 8 + srow;
      a_pix[i] = -1; a_y[i] = 0; a_x[i] = 0; a_bbase[i] = 0;
      if (m < a.M) {
        a_pix[i] = m;
        if (a.ntaps) {
          const int b = m / HWo, rem = m - b * HWo;
          const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
          a_y[i] = oy * a.stride - (a.pad0 ? 0 : 1); a_x[i] = ox * a.stride - (a.pad0 ? 0 : 1);
          a_bbase[i] = b * a.Hin * a.Win;
        }
      }
    }
    int w_row[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int n = n0 + (i * 4 + rw) * 8 + srow;
      w_row[i] = (n < a.N) ? n * a.ldw : -1;
    }
    const int Hv = a.ups ? a.Hin * 2 : a.Hin, Wv = a.ups ? a.Win * 2 : a.Win;
    const unsigned cc = (unsigned)a.conv_c;
    // lean tap staging (stride-1 3x3 convs over whole 64-channel slices): centre-pixel offset + 9-bit tap-validity mask per piece
    const bool leanc = a.ntaps == 9 && a.stride == 1 && a.ups == 0 && !a.pad0 && (a.conv_c % BK) == 0;
    unsigned c_pre[IA], c_mask[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      c_pre[i] = 0; c_mask[i] = 0;
      if (leanc && a_pix[i] >= 0) {
        c_pre[i] = (unsigned)(a_bbase[i] + (a_y[i] + 1) * a.Win + (a_x[i] + 1)) * cc + (unsigned)sslot * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int yy = a_y[i] + t / 3, xx = a_x[i] + t % 3;
          if ((unsigned)yy < (unsigned)a.Hin && (unsigned)xx < (unsigned)a.Win) c_mask[i] |= 1u << t;
        }
      }
    }
    const bf16_t* psrc0 = a.p_src[0];
    const bf16_t* psrc1 = a.p_src[1];
    asm volatile("" : "+s"(psrc0), "+s"(psrc1));      // keep them in SGPRs (see gemm.hip)
    auto glds = [&](const bf16_t* src, unsigned char* dst) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue_stage = [&](const KIter& it, int buf) {
      unsigned char* As = smem + buf * STAGE + rw * 1024;
      unsigned char* Bs = As + A_BYTES;
      const int ch = it.c0 + sslot * 8;                 // channel of this lane's 16-byte chunk
      const bool kin = ch < it.seglen;
      if (it.seg < a.ntaps && leanc) {
        const int ky = it.seg / 3, kx = it.seg - ky * 3;
        const unsigned delta = (unsigned)(((ky - 1) * a.Win + (kx - 1)) * (int)cc + it.c0);     // wave-uniform
        const unsigned bit = 1u << it.seg;
#pragma unroll
        for (int i = 0; i < IA; ++i)
          glds((c_mask[i] & bit) ? a.conv_src + (c_pre[i] + delta) : a.zero, As + i * 4096);
      } else if (it.seg < a.ntaps) {
        const int ky = it.seg / 3, kx = it.seg - ky * 3;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
          const int yy = a_y[i] + ky, xx = a_x[i] + kx;
          const bool ok = kin & (a_pix[i] >= 0) & ((unsigned)yy < (unsigned)Hv) & ((unsigned)xx < (unsigned)Wv) &
                          ((a.ups != 2) | (((yy | xx) & 1) == 0));
          const int sy = a.ups ? (yy >> 1) : yy, sx = a.ups ? (xx >> 1) : xx;
          const unsigned off = (unsigned)(a_bbase[i] + sy * a.Win + sx) * cc + (unsigned)ch;
          glds(ok ? a.conv_src + off : a.zero, As + i * 4096);
        }
      } else {
        const bf16_t* base = it.seg == a.ntaps ? psrc0 : psrc1;
        const unsigned pc = (unsigned)it.seglen;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
          const bool ok = kin & (a_pix[i] >= 0);
          const unsigned off = (unsigned)a_pix[i] * pc + (unsigned)ch;
          glds(ok ? base + off : a.zero, As + i * 4096);
        }
      }
      const unsigned wc = (unsigned)(it.wcol + sslot * 8);
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const bool ok = kin & (w_row[i] >= 0);
        glds(ok ? a.W + ((unsigned)w_row[i] + wc) : a.zero, Bs + i * 4096);
      }
    };
    if (nk > 0) {
      KIter it = kiter_at(a, 0);
      int issued = 0;
#pragma unroll
      for (int s = 0; s < NST - 1; ++s) {
        if (s < nk) {
          if (s) kiter_next(a, it);
          issue_stage(it, s);
          ++issued;
        }
      }
      int buf = 0;
      for (int t = 0; t < nk; ++t) {
        const int ahead = issued - 1 - t;               // stages after t already in flight: they may stay in flight
        if (ahead == 0) ws_vmcnt<0>();
        else if (ahead == 1) ws_vmcnt<NPW>();
        else ws_vmcnt<2 * NPW>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // stage t visible to the matrix waves; stage t-1 consumed by all of them
        asm volatile("" ::: "memory");
        if (issued < nk) {
          int nb = buf - 1; if (nb < 0) nb += NST;
          kiter_next(a, it);
          issue_stage(it, nb);
          ++issued;
        }
        if (++buf == NST) buf = 0;
      }
    }
  } else {
    // ================================================================= matrix waves: ds_read_b128 + MFMA
    // fragment of k-step ks: lane (row ql, half kh) reads the 16 bytes of chunk 2 ks + kh of its row: row * 128 + ((chunk ^ sw) << 4),
    // sw = (row >> 1) & 7 = (ql >> 1) & 7 (block bases are multiples of 32).  Bank (address / 4) % 64 = (row & 1) * 32 + slot * 4: the
    // 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) hold 8 even and 8 odd rows whose (row >> 1) & 7 are all different.
    const int sw = (ql >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = ((2 * ks + kh) ^ sw) << 4;
    const int x_row = (rw * 64 + ql) * 128;
    const int w_row = A_BYTES + ql * 128;
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned char* S = smem + buf * STAGE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8_t xf[2], wf[NCB];
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) xf[pj] = *(const bf16x8_t*)(S + x_row + pj * 32 * 128 + foff[ks]);
#pragma unroll
        for (int ci = 0; ci < NCB; ++ci) wf[ci] = *(const bf16x8_t*)(S + w_row + ci * 32 * 128 + foff[ks]);
#pragma unroll
        for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
          for (int pj = 0; pj < 2; ++pj)
            // weights as MFMA-A: D[row = channel][col = pixel]
            acc[ci][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ci], xf[pj], acc[ci][pj], 0, 0, 0);
      }
      if (++buf == NST) buf = 0;
    }
  }

  // ---------------------------------------------------------------- epilogue (all eight waves)
  // lane (pixel ql of block pj, half kh) of matrix wave rw holds channels ci*32 + 8 (r >> 2) + 4 kh + (r & 3) of pixel rw*64 + pj*32 + ql
  if (a.act == ACT_GEGLU) {
    // rows 0..15 of a 32-row accumulator block are values, 16..31 the gates of the same 16 hidden units (packed in 16-row blocks):
    // bias, exact-erf GELU and the product on the accumulators; the bf16 result (BN / 2 columns) is staged for full-row stores
    constexpr int RSG = BN + 16;                     // bf16 row stride of the staged [BM][BN / 2] tile (bytes)
    static_assert(BM * RSG <= NST * STAGE, "staged GEGLU tile must fit the pipeline buffers");
    __syncthreads();
    if (!loader) {
#pragma unroll
      for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int nv = n0 + ci * 32 + 8 * g + 4 * kh;
          float4 bv = float4{0, 0, 0, 0}, bg = bv;
          if (a.bias && nv + 16 < a.N) { bv = *(const float4*)(a.bias + nv); bg = *(const float4*)(a.bias + nv + 16); }
#pragma unroll
          for (int pj = 0; pj < 2; ++pj) {
            const f32x16_t& c = acc[ci][pj];
            uint2 o;
            o.x = pack2bf((c[4 * g] + bv.x) * gelu_erf_f(c[8 + 4 * g] + bg.x), (c[4 * g + 1] + bv.y) * gelu_erf_f(c[9 + 4 * g] + bg.y));
            o.y = pack2bf((c[4 * g + 2] + bv.z) * gelu_erf_f(c[10 + 4 * g] + bg.z), (c[4 * g + 3] + bv.w) * gelu_erf_f(c[11 + 4 * g] + bg.w));
            *(uint2*)(smem + (rw * 64 + pj * 32 + ql) * RSG + (ci * 16 + 8 * g + 4 * kh) * 2) = o;
          }
        }
    }
    __syncthreads();
    constexpr int CPRG = BN / 16;                    // 16-byte chunks per output row of the tile
    for (int c = tid; c < BM * CPRG; c += 512) {
      const int row = c / CPRG, cc = c - row * CPRG;
      const int m = m0 + row, oc = (n0 >> 1) + cc * 8;
      if (m < a.M && oc < (a.N >> 1))
        *(uint4*)((bf16_t*)a.out + (long)m * a.ld_out + oc) = *(const uint4*)(smem + row * RSG + cc * 16);
    }
    return;
  }

  constexpr int RSF = BN * 4 + 16;                   // fp32 row stride of a 64-row pass (bytes)
  constexpr int CPR = BN / 8;                        // 8-column chunks per row
  constexpr int EPI = (64 * CPR + 511) / 512;        // chunks per thread and pass
  constexpr int BIAS_OFF = 64 * RSF;
  static_assert(BIAS_OFF + BN * 4 <= NST * STAGE, "epilogue tile must fit the pipeline buffers");
  uint4 rnext[EPI];
  auto fetch_resid = [&](int q) {                    // residual rows one pass ahead (see gemm_wide.hip)
#pragma unroll
    for (int e = 0; e < EPI; ++e) {
      const int c = tid + e * 512;
      const int row = c / CPR, cchunk = c - row * CPR;
      const int m = m0 + q * 64 + row, n = n0 + cchunk * 8;
      rnext[e] = make_uint4(0u, 0u, 0u, 0u);
      if (c < 64 * CPR && m < a.M && n < a.N) rnext[e] = *(const uint4*)(a.resid + (long)m * a.ld_res + n);
    }
  };
  const bool has_resid = a.resid != nullptr;
  if (has_resid) fetch_resid(0);
  float4 bias_reg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias && tid < BN / 4 && n0 + tid * 4 < a.N) bias_reg = *(const float4*)(a.bias + n0 + tid * 4);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    __syncthreads();                                 // pipeline buffers / previous pass no longer read
    if (!loader && rw == q) {
#pragma unroll
      for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
        for (int pj = 0; pj < 2; ++pj)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4_t v = f32x4_t{acc[ci][pj][4 * g], acc[ci][pj][4 * g + 1], acc[ci][pj][4 * g + 2], acc[ci][pj][4 * g + 3]};
            *(f32x4_t*)(smem + (pj * 32 + ql) * RSF + (ci * 32 + 8 * g + 4 * kh) * 4) = v;
          }
    }
    if (q == 0 && tid < BN / 4) *(float4*)(smem + BIAS_OFF + tid * 16) = bias_reg;
    __syncthreads();
    uint4 rcur[EPI];
#pragma unroll
    for (int e = 0; e < EPI; ++e) rcur[e] = rnext[e];
    if (has_resid && q + 1 < 4) fetch_resid(q + 1);
#pragma unroll
    for (int e = 0; e < EPI; ++e) {
      const int c = tid + e * 512;
      const int row = c / CPR, cchunk = c - row * CPR;
      const int m = m0 + q * 64 + row, n = n0 + cchunk * 8;
      if (c >= 64 * CPR || m >= a.M || n >= a.N) continue;
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
        const float* rv = a.rowvec + (long)(m / a.rows_per_b) * a.rv_ld + a.rv_off + n;
        const float4 r0 = *(const float4*)rv, r1 = *(const float4*)(rv + 4);
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
        unpack8(rcur[e], f);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += f[r];
      }
      *(uint4*)((bf16_t*)a.out + (long)m * a.ld_out + n) = pack8(v);
    }
  }
}

template <int BN>
int ws_launch_t(GemmArgs a, hipStream_t s) {
  constexpr int lds = WS_NST * (WS_BM + BN) * BK * 2;
  static_assert(lds <= 160 * 1024, "LDS ring exceeds the CU's 160 KiB");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_ws_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  a.ksteps = dfh::gemm_count_ksteps(a);
  a.ksplit = 1;
  const int tiles = ((a.M + WS_BM - 1) / WS_BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_ws_kernel<BN>), dim3(tiles), dim3(512), lds, s, a);
  return dfh::check_launch("gemm_ws_kernel");
}

}  // namespace

namespace dfh {

// 0 = not eligible, 160 / 128 = column tile.  bf16 row-major output, no split-K; GEGLU needs whole 32-row (value, gate) blocks.
int gemm_ws_pick(const GemmArgs& a, int min_tiles) {
  if (a.out_mode != OUT_BF16) return 0;
  if ((a.N & 7) || (a.ld_out & 7) || (a.resid && (a.ld_res & 7))) return 0;
  if (a.act == ACT_GEGLU && ((a.N % 32) || a.resid || a.rowvec)) return 0;
  int bn;
  if (a.N % 160 == 0) bn = 160;
  else if (a.N % 128 == 0) bn = 128;
  else if (a.N > 640) bn = 160;                        // ragged last column tile: < 1/5 of the columns padded
  else return 0;
  const long tiles = (long)((a.M + WS_BM - 1) / WS_BM) * ((a.N + bn - 1) / bn);
  return tiles >= min_tiles ? bn : 0;
}

int gemm_ws_launch(GemmArgs a, hipStream_t s, int bn) {
  return bn == 128 ? ws_launch_t<128>(a, s) : ws_launch_t<160>(a, s);
}

}  // namespace dfh
