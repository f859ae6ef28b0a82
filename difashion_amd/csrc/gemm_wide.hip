// Wide-tile variant of the implicit GEMM (same GemmArgs / K-segment semantics as gemm.hip).
//
// Why it exists: on MI355X the 128 x {64,128,160} tiles of gemm.hip run at a throughput PROPORTIONAL to their
// arithmetic intensity against the global->LDS staging path (43 / 64 / 71 flop per staged byte -> 527 / 792 / 852
// TFLOP/s on a 65536 x 1280 x 1280 GEMM, scripts/gemm_tiles_probe.py): the LDS-DMA path, not the MFMA pipe, is the
// limiter.  This kernel raises the intensity to 98 flop/B while keeping two workgroups per CU:
//   * 256 x 160 block tile, 4 waves as 2 x 2, each 128 x 80 (8 x 5 MFMA fragments, 160 accumulator VGPRs);
//   * 32-deep k-steps (one v_mfma_f32_16x16x32_bf16 per fragment pair), 26 KB per stage, 3 stages in flight
//     (78 KB), one barrier per 40 MFMAs;
//   * 64-byte LDS rows, 16-byte slots swizzled by (row >> 1) & 3 on the SOURCE address and on the fragment read
//     (conflict-free for the ds_read_b128 lane groups; linear rows are 2-way conflicted);
//   * epilogue in four 64-row passes through an fp32 LDS tile: bias / time-embedding row / activation / residual are
//     applied on full rows, so residual reads and output writes are 16-byte pieces of contiguous rows and the result is
//     rounded to bf16 exactly once.
// Used for bf16-output launches without GEGLU / split-K whose grid fills the chip (gemm_wide_eligible).
#include "gemm.h"
#include "gemm_wide_epilogue.h"

#include <algorithm>

namespace {

constexpr int BKW = 32;

struct KIterW { int seg, c0, wcol, seglen; };

DFH_DEVICE int seg_len_w(const GemmArgs& a, int seg) {
  return seg < a.ntaps ? a.conv_c : (seg == a.ntaps ? a.p_c[0] : a.p_c[1]);
}
// channel-chunk-major over the conv taps, see gemm.hip
DFH_DEVICE KIterW kiterw_at(const GemmArgs& a, int kstep) {
  KIterW it;
  const int conv_steps = a.ntaps * ((a.conv_c + BKW - 1) / BKW);
  if (kstep < conv_steps) {
    const int cc = kstep / a.ntaps;
    it.seg = kstep - cc * a.ntaps; it.c0 = cc * BKW; it.seglen = a.conv_c; it.wcol = it.seg * a.conv_c + it.c0;
    return it;
  }
  kstep -= conv_steps;
  int seg = a.ntaps, base = a.ntaps * a.conv_c;
  const int nseg = a.ntaps + a.nplain;
  for (;;) {
    const int len = seg_len_w(a, seg);
    const int n = (len + BKW - 1) / BKW;
    if (kstep < n || seg == nseg - 1) { it.seglen = len; break; }
    kstep -= n; base += len; ++seg;
  }
  it.seg = seg; it.c0 = kstep * BKW; it.wcol = base + it.c0;
  return it;
}
DFH_DEVICE void kiterw_next(const GemmArgs& a, KIterW& it) {
  if (it.seg < a.ntaps) {
    ++it.seg; it.wcol += a.conv_c;
    if (it.seg < a.ntaps) return;
    it.seg = 0; it.c0 += BKW; it.wcol = it.c0;
    if (it.c0 < a.conv_c) return;
    it.seg = a.ntaps; it.c0 = 0; it.wcol = a.ntaps * a.conv_c;
    if (a.nplain > 0) it.seglen = seg_len_w(a, it.seg);
    return;
  }
  it.c0 += BKW; it.wcol += BKW;
  if (it.c0 >= it.seglen) {
    it.wcol -= it.c0 - it.seglen;
    it.c0 = 0; ++it.seg;
    if (it.seg < a.ntaps + a.nplain) it.seglen = seg_len_w(a, it.seg);
  }
}
template <int N> DFH_DEVICE void ww_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define WRD_(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// WN = 2: four waves (2 x 2), two workgroups per CU, 3 stages.  WN = 4: eight waves (2 x 4) on a 256 x 320 tile, ONE workgroup
// per CU, 4 stages (144 KB): 142 flop per staged byte.
// ABL (probe only, tile ids 13..19 = ABL 1..7; results are garbage): bit 0 drops the LDS-DMA staging of the k-loop, bit 1 the
// fragment reads, bit 2 the MFMAs -- what is left is timed by scripts/gemm_ablate_probe.py to see which resource paces the loop.
template <int BM, int BN, int WN, int NSTAGE, int ABL = 0>
__global__ __launch_bounds__(WN * 128, WN == 2 ? 2 : 1) void gemm_wide_kernel(const GemmArgs a) {
  constexpr int NWV = 2 * WN;
  constexpr int TM = BM / 2, TN = BN / WN;           // per-wave output tile
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int PA = BM / 16, PB = BN / 16;          // 1-KiB staging pieces (16 rows x 64 B) per stage
  constexpr int IA = PA / NWV, IB = (PB + NWV - 1) / NWV;
  constexpr int A_BYTES = BM * BKW * 2, B_BYTES = BN * BKW * 2, STAGE = A_BYTES + B_BYTES;
  static_assert(PA % NWV == 0 && TM % 16 == 0 && TN % 16 == 0, "tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  int mt_, nt_;
  tile_coords(blockIdx.x, ntm, ntn, a.n_major, a.tm_xm, a.tm_gm, mt_, nt_);
  const int m0 = mt_ * BM, n0 = nt_ * BN;
  const int nk = a.ksteps;                           // 32-deep steps (filled by the launcher)
  // folded LayerNorm (consumer, GEGLU epilogue below): this thread's row statistics, the wave's oldest vector-memory operations
  float2 lnmr = float2{0.f, 1.f};
  if (a.ln_stat != nullptr && tid < BM) lnmr = ln_row_stats(a, m0 + tid);

  // staging geometry: piece p = i*4 + wave holds tile rows p*16 + lane/4; the LDS image is lane-linear, the SOURCE
  // 16-byte chunk is swizzled: slot s of row r holds channel chunk s ^ ((r >> 1) & 3)
  const int srow = lane >> 2;
  const int schunk = (lane & 3) ^ ((lane >> 3) & 3);
  int a_pix[IA], a_y[IA], a_x[IA], a_bbase[IA];
  const int HWo = a.Hout * a.Wout;
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int m = m0 + (i * NWV + wave) * 16 + srow;
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
    const int n = n0 + (i * NWV + wave) * 16 + srow;
    w_row[i] = (n < a.N) ? n * a.ldw : -1;
  }
  const int Hv = a.ups ? a.Hin * 2 : a.Hin, Wv = a.ups ? a.Win * 2 : a.Win;
  constexpr int N_LO = IA + PB / NWV, N_HI = N_LO + 1, PB_REM = PB % NWV;
  const bool hi_wave = wave < PB_REM;

  const unsigned cc = (unsigned)a.conv_c;
  // Lean tap staging (stride-1 3x3 convs over whole 32-channel slices: every resnet conv): the centre-pixel offset and a
  // 9-bit tap-validity mask per staging piece are fixed for the kernel, so a k-step is one add, one bit test and the
  // zero-page select per piece instead of the coordinate arithmetic and four bounds compares of the general path.
  const bool leanc = a.ntaps == 9 && a.stride == 1 && a.ups == 0 && !a.pad0 && (a.conv_c % BKW) == 0;
  unsigned c_pre[IA], c_mask[IA];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    c_pre[i] = 0; c_mask[i] = 0;
    if (leanc && a_pix[i] >= 0) {
      const int cy = a_y[i] + 1, cx = a_x[i] + 1;                    // the output pixel = centre tap
      c_pre[i] = (unsigned)(a_bbase[i] + cy * a.Win + cx) * cc + (unsigned)schunk * 8;
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
  auto issue_stage = [&](const KIterW& it, int buf) {
    unsigned char* As = smem + buf * STAGE + wave * 1024;
    unsigned char* Bs = As + A_BYTES;
    const int ch = it.c0 + schunk * 8;
    const bool kin = ch < it.seglen;
    if (it.seg < a.ntaps && leanc) {
      const int ky = it.seg / 3, kx = it.seg - ky * 3;
      const unsigned delta = (unsigned)(((ky - 1) * a.Win + (kx - 1)) * (int)cc + it.c0);     // wave-uniform
      const unsigned bit = 1u << it.seg;
#pragma unroll
      for (int i = 0; i < IA; ++i)
        glds((c_mask[i] & bit) ? a.conv_src + (c_pre[i] + delta) : a.zero, As + i * NWV * 1024);
    } else if (it.seg < a.ntaps) {
      const int ky = it.seg / 3, kx = it.seg - ky * 3;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const int yy = a_y[i] + ky, xx = a_x[i] + kx;
        const bool ok = kin & (a_pix[i] >= 0) & ((unsigned)yy < (unsigned)Hv) & ((unsigned)xx < (unsigned)Wv) &
                        ((a.ups != 2) | (((yy | xx) & 1) == 0));
        const int sy = a.ups ? (yy >> 1) : yy, sx = a.ups ? (xx >> 1) : xx;
        const unsigned off = (unsigned)(a_bbase[i] + sy * a.Win + sx) * cc + (unsigned)ch;
        glds(ok ? a.conv_src + off : a.zero, As + i * NWV * 1024);
      }
    } else {
      const bf16_t* base = it.seg == a.ntaps ? psrc0 : psrc1;
      const unsigned pc = (unsigned)it.seglen;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const bool ok = kin & (a_pix[i] >= 0);
        const unsigned off = (unsigned)a_pix[i] * pc + (unsigned)ch;
        glds(ok ? base + off : a.zero, As + i * NWV * 1024);
      }
    }
    const unsigned wc = (unsigned)(it.wcol + schunk * 8);
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      if (i * NWV + wave >= PB) continue;             // wave-uniform
      const bool ok = kin & (w_row[i] >= 0);
      glds(ok ? a.W + ((unsigned)w_row[i] + wc) : a.zero, Bs + i * NWV * 1024);
    }
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  // fragment read offsets inside a stage: row * 64 + ((fg ^ ((row >> 1) & 3)) << 4); (row >> 1) & 3 depends on fr only
  const int fslot = (fg ^ ((fr >> 1) & 3)) << 4;
  const unsigned a_off = (wm * TM + fr) * 64 + fslot;
  const unsigned b_off = A_BYTES + (wn * TN + fr) * 64 + fslot;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  static_assert((FM == 8 || FM == 4) && (FN == 5 || FN == 4), "the hand-scheduled k-step is written for {8,4} x {5,4} fragments");

  if (nk > 0) {
    KIterW it = kiterw_at(a, 0);
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
      if (s < nk) {
        if (s) kiterw_next(a, it);
        issue_stage(it, s);
        ++issued;
      }
    }
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
      const int ahead = issued - 1 - t;
      if (ahead == 0) ww_vmcnt<0>();
      else if (ahead == 1) { if (hi_wave) ww_vmcnt<N_HI>(); else ww_vmcnt<N_LO>(); }
      else if (ahead == 2 || NSTAGE == 3) { if (hi_wave) ww_vmcnt<2 * N_HI>(); else ww_vmcnt<2 * N_LO>(); }
      else { if (hi_wave) ww_vmcnt<3 * N_HI>(); else ww_vmcnt<3 * N_LO>(); }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (issued < nk) {
        kiterw_next(a, it);
        int nb = buf - 1; if (nb < 0) nb += NSTAGE;
        if constexpr (!(ABL & 1)) issue_stage(it, nb);
        ++issued;
      }
      // Fragment reads software-pipelined by hand (inline asm + counted lgkmcnt): the A fragment of row i+2 is in
      // flight while row i's five MFMAs run, so no MFMA group waits for a full LDS round trip (left to hipcc the reads
      // are issued right in front of an s_waitcnt lgkmcnt(0) five times per step).
      const unsigned sa = lds0 + buf * STAGE + a_off, sb = lds0 + buf * STAGE + b_off;
      u32x4_t b[FN], a0, a1, a2;
      if constexpr (ABL & 2) {
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "=v"(b[j]));
        asm volatile("" : "=v"(a0)); asm volatile("" : "=v"(a1)); asm volatile("" : "=v"(a2));
      }
      __builtin_amdgcn_sched_barrier(0);
#define WRD(dst, addr, off) do { if constexpr (!(ABL & 2)) WRD_(dst, addr, off); } while (0)
      WRD(b[0], sb, 0); WRD(b[1], sb, 1024); WRD(b[2], sb, 2048); WRD(b[3], sb, 3072);
      if constexpr (FN == 5) WRD(b[4], sb, 4096);
      WRD(a0, sa, 0); WRD(a1, sa, 1024);
      if constexpr (FN == 5) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(a0));
      else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(a0));
#define WROW(i, ar)                                                                                                    \
      if constexpr (!(ABL & 4)) _Pragma("unroll") for (int j = 0; j < FN; ++j)                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[j]), __builtin_bit_cast(bf16x8_t, ar), \
                                                            acc[i][j], 0, 0, 0);
#define WNEXT(rd, off, wt) WRD(rd, sa, off); asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wt));
      // weights as MFMA-A: D[row = channel (fg*4+r)][col = pixel (fr)]
      if constexpr (FM == 8) {
        WROW(0, a0) WNEXT(a2, 2048, a1)
        WROW(1, a1) WNEXT(a0, 3072, a2)
        WROW(2, a2) WNEXT(a1, 4096, a0)
        WROW(3, a0) WNEXT(a2, 5120, a1)
        WROW(4, a1) WNEXT(a0, 6144, a2)
        WROW(5, a2) WNEXT(a1, 7168, a0)
        WROW(6, a0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a1));
        WROW(7, a1)
      } else {
        WROW(0, a0) WNEXT(a2, 2048, a1)
        WROW(1, a1) WNEXT(a0, 3072, a2)
        WROW(2, a2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0));
        WROW(3, a0)
      }
#undef WROW
#undef WNEXT
#undef WRD
      __builtin_amdgcn_sched_barrier(0);
      if (++buf == NSTAGE) buf = 0;
    }
  }

  // ---------------------------------------------------------------- GEGLU epilogue in registers (column tiles whose waves own
  // whole (value, gate) block pairs: FN even).  The packed weight rows interleave values and gates in 16-row blocks, so
  // acc[i][j] / acc[i][j + 1] hold value and gate of the SAME four hidden units of one pixel in one lane: bias, exact-erf GELU and
  // the product run on the accumulators, and only the bf16 result (half the columns) is staged through LDS for full-row
  // 16-byte stores.  (The four fp32 passes below cost 22k of the 48k cycles of a K = 320 tile -- more than its k-loop.)
  if constexpr (FN % 2 == 0) {
    if (a.act == ACT_GEGLU) {
      constexpr int RSG = BN + 16;                     // bf16 row stride of the staged [BM][BN / 2] tile (bytes)
      constexpr int LNROWG = BM * RSG;                 // folded LayerNorm (gemm.h): (mean, rstd) of the tile's rows behind the staged tile
      static_assert(LNROWG + BM * 8 <= NSTAGE * STAGE, "staged GEGLU tile + row statistics must fit the pipeline buffers");
      static_assert(BM <= NWV * 64, "one thread per tile row");
      const bool lnf = a.ln_stat != nullptr;
      __syncthreads();                                 // every wave is done reading the last pipeline stage
      if (lnf) {
        if (tid < BM) *(float2*)(smem + LNROWG + tid * 8) = lnmr;
        __syncthreads();
      }
      const GeluK gk = gelu_consts();
      float4 bv[FN / 2], bg[FN / 2], sv[FN / 2], sg[FN / 2];
#pragma unroll
      for (int jj = 0; jj < FN / 2; ++jj) {
        const int n = n0 + wn * TN + jj * 32 + fg * 4;
        bv[jj] = float4{0, 0, 0, 0}; bg[jj] = float4{0, 0, 0, 0}; sv[jj] = bv[jj]; sg[jj] = bv[jj];
        if (a.bias && n + 16 < a.N) { bv[jj] = *(const float4*)(a.bias + n); bg[jj] = *(const float4*)(a.bias + n + 16); }
        if (lnf && n + 16 < a.N) { sv[jj] = *(const float4*)(a.ln_s + n); sg[jj] = *(const float4*)(a.ln_s + n + 16); }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * TM + i * 16 + fr;
        float2 mr = float2{0.f, 1.f};
        if (lnf) mr = *(const float2*)(smem + LNROWG + row * 8);
        const float ms = -mr.x * mr.y;                 // rstd * (acc - mean * s) + b' == rstd * acc + (b' - rstd * mean * s)
#pragma unroll
        for (int jj = 0; jj < FN / 2; ++jj) {
          // value / gate with their additive constants (bias, or the folded-LayerNorm fix-up: two FMAs per element), GELU and product on
          // packed fp32 pairs: this epilogue is VALU-bound, every instruction shows in the launch time (dfh_common.h geglu4)
          uint2 o;
          if constexpr ((ABL & 8) != 0) {                // training: the pre-activations leave as a second output (GemmArgs::pre_out)
            uint2 pv, pg;
            o = geglu4p(acc[i][2 * jj], acc[i][2 * jj + 1], bv[jj], bg[jj], gk, pv, pg);
            const int m = m0 + row, n = n0 + wn * TN + jj * 32 + fg * 4;
            if (m < a.M && n + 16 < a.N) {
              bf16_t* pr = (bf16_t*)a.pre_out + (long)m * a.ld_pre + n;
              *(uint2*)pr = pv; *(uint2*)(pr + 16) = pg;
            }
          } else {
            o = geglu4(acc[i][2 * jj], acc[i][2 * jj + 1], bv[jj], bg[jj], sv[jj], sg[jj], lnf, mr.y, ms, gk);
          }
          const int ocl = ((wn * TN) >> 1) + jj * 16 + fg * 4;           // output column inside the tile
          *(uint2*)(smem + row * RSG + ocl * 2) = o;
        }
      }
      __syncthreads();
      constexpr int CPRG = BN / 16;                    // 16-byte chunks per output row of the tile
      for (int c = tid; c < BM * CPRG; c += NWV * 64) {
        const int row = c / CPRG, cc = c - row * CPRG;
        const int m = m0 + row, oc = (n0 >> 1) + cc * 8;
        if (m < a.M && oc < (a.N >> 1))
          *(uint4*)((bf16_t*)a.out + (long)m * a.ld_out + oc) = *(const uint4*)(smem + row * RSG + cc * 16);
      }
      return;
    }
  }

  // ---------------------------------------------------------------- epilogue: four 64-row passes through fp32 LDS
  wide_epilogue<BM, BN, WN, NSTAGE * STAGE>(a, acc, smem, tid, wm, wn, fr, fg, m0, n0);
}

}  // namespace

namespace dfh {

int gemm_wide_ksteps(const GemmArgs& a) {
  int n = a.ntaps * ((a.conv_c + BKW - 1) / BKW);
  for (int i = 0; i < a.nplain; ++i) n += (a.p_c[i] + BKW - 1) / BKW;
  return n;
}

// bf16 row-major output, no GEGLU / split-K, 16-byte aligned rows, and enough tiles to give every CU its two workgroups
// 1 = 256 x 160, 4 = 256 x 128 (N a multiple of 128 but not of 160), 0 = not eligible; 2 / 3 are experiment variants
int gemm_wide_pick(const GemmArgs& a) {
  if (a.out_mode != OUT_BF16) return 0;
  if (a.ln_stat && !(a.act == ACT_GEGLU && a.N % 128 == 0)) return 0;   // only the in-register GEGLU epilogue implements the LayerNorm fix-up
  if (a.act == ACT_GEGLU && (a.resid || a.rowvec)) return 0;
  if ((a.N & 7) || (a.ld_out & 7) || (a.resid && (a.ld_res & 7))) return 0;
  // GEGLU: the 256 x 128 sibling keeps whole (value, gate) block pairs inside a wave -> epilogue in registers
  if (a.act == ACT_GEGLU && a.N % 128 == 0) return (long)((a.M + 255) / 256) * (a.N / 128) >= 448 ? 4 : 0;
  if (a.act == ACT_GEGLU && a.N % 160 != 0) return 0;
  if (a.N % 160 != 0 && a.N % 128 == 0) {       // 128 / 256 / 512 / 1024 output channels (the VAE): the 256 x 128 sibling
    return (long)((a.M + 255) / 256) * (a.N / 128) >= 448 ? 4 : 0;
  }
  if (a.N % 160 != 0 && a.N < 640) return 0;
  // plain linears / 1x1 convs (no taps): the eight-wave 128 x 160 kernel of gemm.hip is as fast or faster since its epilogue
  // stopped serialising the bias loads and its residual loads moved behind the prologue (scripts/gemm_shortk2_probe.py,
  // profiles/r02/gemm_shortk2_probe.txt: 65536 x 960 x 320 72.7 -> 60.2 us, 65536 x 320 x 1280 + residual 75.3 -> 67.7 us,
  // 16384 x 1920 x 640 63.4 -> 47.6 us); the 3x3 convs keep the wide tile, and so do the K = 320 linears with a residual, whose
  // coalesced one-pass-ahead residual reads win (same-box A/B: 31.4 vs 33.7 us)
  // (with row statistics for a folded LayerNorm the eight-wave kernel wins again: the 256-row epilogue pays ~4.5 us per launch for them)
  if (a.ntaps == 0 && a.N % 160 == 0 && !(a.resid && a.nplain == 1 && a.p_c[0] <= 320 && !a.rowstat)) return 0;
  const long nt = (a.N + 159) / 160;
  // the 128-row sibling (variant 2) is kept for experiments only: at equal tile size the 64-deep two-stage kernel of
  // gemm.hip wins (848 vs 724 TFLOP/s on conv 320->320 @64): the gain of this file is the larger tile
  return (long)((a.M + 255) / 256) * nt >= 448 ? 1 : 0;
}

bool gemm_wide_eligible(const GemmArgs& a) {
  if (a.out_mode != OUT_BF16) return false;
  if (a.act == ACT_GEGLU && (a.N % 160 != 0 || a.resid || a.rowvec)) return false;
  if ((a.N & 7) || (a.ld_out & 7) || (a.resid && (a.ld_res & 7))) return false;
  if (a.N % 160 != 0 && a.N < 640) return false;
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 159) / 160);
  return tiles >= 448;
}

template <int BM, int BN, int WN, int NSTAGE, int ABL = 0>
static int wide_launch_t(GemmArgs a, hipStream_t s) {
  constexpr int lds = NSTAGE * (BM + BN) * BKW * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<BM, BN, WN, NSTAGE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  a.ksteps = gemm_wide_ksteps(a);
  a.ksplit = 1;
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_wide_kernel<BM, BN, WN, NSTAGE, ABL>), dim3(tiles), dim3(WN * 128), lds, s, a);
  return check_launch("gemm_wide_kernel");
}

int gemm_wide_launch(GemmArgs a, hipStream_t s, int variant) {
  if (variant == 4) return wide_launch_t<256, 128, 2, 3>(a, s);
  if (variant == 5) return wide_launch_t<256, 128, 2, 3, 8>(a, s);      // GEGLU with the pre-activations as a second output (training)
#ifdef DFH_PROBES   // experiment tiles (eight-wave 256 x 320, 128-row sibling) and the k-loop ablations: probe builds only
  if (variant == 3) return wide_launch_t<256, 320, 4, 4>(a, s);
  if (variant == 2) return wide_launch_t<128, 160, 2, 3>(a, s);
  switch (variant) {                                 // ablation probes (see the kernel's ABL note)
    case 8: return wide_launch_t<256, 160, 2, 3, 1>(a, s);
    case 9: return wide_launch_t<256, 160, 2, 3, 2>(a, s);
    case 10: return wide_launch_t<256, 160, 2, 3, 3>(a, s);
    case 11: return wide_launch_t<256, 160, 2, 3, 4>(a, s);
    case 12: return wide_launch_t<256, 160, 2, 3, 5>(a, s);
    case 13: return wide_launch_t<256, 160, 2, 3, 6>(a, s);
    default: break;
  }
#else
  DFH_REQUIRE(variant == 1, "wide-kernel variants 2 / 3 / 8-13 are probe instantiations (make -C scripts/probes)");
#endif
  return wide_launch_t<256, 160, 2, 3>(a, s);
}

}  // namespace dfh
