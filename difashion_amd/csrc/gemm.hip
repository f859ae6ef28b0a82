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
//   * N-stage LDS ring: loads for k-step t+NSTAGE-1 are issued while k-step t computes; a wave waits
//     with a COUNTED s_waitcnt vmcnt (never a full drain in steady state) and the workgroup meets at one
//     raw s_barrier per k-step, so LDS-DMA stays in flight across barriers (guide T3/T4).
//     Main tile 256 x {160,128} x 64 with 8 waves (4x2) and 3 stages = 156/144 KiB of the CU's 160 KiB;
//     a 128-row / 4-wave / 2-stage variant (72 KiB, 2 workgroups per CU) covers narrow or tiny shapes.
//   * the weight operand is fed as MFMA "A" so each lane ends up with 4 consecutive output
//     channels of one pixel -> 8-byte packed bf16 stores and float4 bias loads.
//   * XCD-aware tile order; split-K through fp32 slabs + a deterministic reduce kernel for the
//     8x8 / 16x16 levels whose M is too small to fill 256 CUs.
#include "gemm.h"
#include "gemm_kiter.h"
#include "gemm_wide_epilogue.h"
#include "norm.h"   // GN_MAX_CHUNKS: what the consuming GroupNorm kernels accept per (image, group)
#ifdef DFH_PROBES
#include "token_linear.h"
#endif

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {

// counted wait on this wave's outstanding vector-memory operations (LDS-DMA pieces)
template <int N> DFH_DEVICE void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------
// LEAN: the launch has ONE plain K segment whose length is a multiple of 64 (every linear / 1x1 of the U-Net): a k-step's
// sources are the previous k-step's plus 128 bytes, so the k-loop keeps one 64-bit pointer per staging piece and adds a
// constant -- no segment iterator, no per-piece predicates / selects / multiplies (the generic loop spends ~190 SALU and
// ~125 VALU instructions per k-step beside 20 MFMAs; with 2 waves per SIMD that, not the MFMA pipe, paces the loop).
// WEPI: the tile's epilogue is the 256-row one of gemm_wide_epilogue.h (four 64-row passes through an fp32 LDS tile) -- the 256 x 320
// eight-wave tile, whose bf16 staging tile would not fit beside nothing (168 KB).  That instantiation is the "big" tile: 64-deep
// k-steps of 128-byte rows (LDS-DMA gathers 128-byte rows at twice the rate of the 64-byte rows of gemm_wide.hip), 128 x 80 per wave
// (13 fragment reads per 40 MFMAs), 72 staging pieces per 640 MFMAs of a k-step (0.11 per MFMA against 0.16 for 256 x 160 x 32), two
// stages = 144 KB, one workgroup per CU.
template <int BM, int BN, int WM, int WN, int NSTAGE, bool LEAN = false, bool WEPI = false>
// second launch-bound = waves per SIMD the register allocation must leave room for: the eight-wave 128-row tiles run TWO workgroups
// per CU (4 waves per SIMD, <= 128 VGPRs); without the bound the allocator settles at 130 and silently halves the occupancy
// (+35 % on every 32x32 / 16x16-level conv, measured)
// (only where TWO workgroups fit the CU's LDS: the 4-stage ring of the same tile is 144 KB = one workgroup per CU = 2 waves per SIMD,
// and under the 128-VGPR bound it spilled 20 registers into its k-loop for an occupancy it can never have)
__global__ __launch_bounds__(WM* WN * 64, (WM * WN == 8 && BM == 128 && 2 * NSTAGE * (BM + BN) * BK * 2 <= 160 * 1024) ? 4 : (WM * WN == 8 ? 2 : 1))
void gemm_bf16_kernel(const GemmArgs a) {
  constexpr int NWV = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;       // per-wave output tile
  constexpr int FM = TM / 16, FN = TN / 16;       // 16x16 fragments per wave
  constexpr int PA = BM / 8, PB = BN / 8;         // 1-KiB staging pieces (8 rows x 128 B) per stage
  constexpr int IA = (PA + NWV - 1) / NWV, IB = (PB + NWV - 1) / NWV;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  static_assert(TM % 16 == 0 && TN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0 && NWV % 2 == 0 && NSTAGE >= 2 && NSTAGE <= 4, "tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int ntn = (a.N + BN - 1) / BN;
  const int ntm = (a.M + BM - 1) / BM;
  // xcd_remap hands each XCD (own L2) a contiguous range of tile ids.  Which operand should that range share?
  // m-major ids (default): an XCD owns a few pixel tiles x ALL column tiles -> it streams every weight row once per
  // group of pixel tiles: fine while the activations are the big operand (64x64 / 32x32 levels).  At the deep levels the
  // WEIGHTS are the big operand (1280 x 11520 bf16 = 29 MB against 10 MB of pixels): n-major ids give an XCD whole column
  // tiles, so each weight row is fetched by one XCD only (a.n_major, set by the launcher).
  int mt_, nt_;
  tile_coords(blockIdx.x, ntm, ntn, a.n_major, a.tm_xm, a.tm_gm, mt_, nt_);
  const int m0 = mt_ * BM, n0 = nt_ * BN;

  // folded LayerNorm (consumer): this thread's row statistics, requested before anything else so that the loads are the oldest
  // vector-memory operations of the wave (they retire first; the k-loop's counted waits are unaffected)
  float2 lnmr = float2{0.f, 1.f};
  if (a.ln_stat != nullptr && tid < BM) lnmr = ln_row_stats(a, m0 + tid);

  // k-step range of this split
  const int per = (a.ksteps + a.ksplit - 1) / a.ksplit;
  const int ks_begin = blockIdx.z * per;
  const int ks_end = min(a.ksteps, ks_begin + per);
  const int nk = ks_end - ks_begin;

  // ---- per-thread staging bookkeeping: piece p = i*NWV + wave covers tile rows p*8 + lane/8,
  //      16-byte slot lane%8; the source slot is swizzled, the LDS image stays lane-linear.
  const int srow = lane >> 3;
  const int sslot = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));   // == slot ^ ((row>>1)&7), NWV even
  int a_pix[IA], a_y[IA], a_x[IA], a_bbase[IA];   // conv: (oy*stride-1, ox*stride-1), batch pixel base; plain: row
  const int HWo = a.Hout * a.Wout;
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int m = m0 + (i * NWV + wave) * 8 + srow;
    a_pix[i] = -1; a_y[i] = 0; a_x[i] = 0; a_bbase[i] = 0;
    if (m < a.M && i * NWV + wave < PA) {
      a_pix[i] = m;
      if (!LEAN && a.ntaps) {                     // LEAN instantiations only ever see plain segments (lean_plain): no pixel decode
        const int b = m / HWo, rem = m - b * HWo;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        a_y[i] = oy * a.stride - (a.pad0 ? 0 : 1); a_x[i] = ox * a.stride - (a.pad0 ? 0 : 1);
        a_bbase[i] = b * a.Hin * a.Win;
      }
    }
  }
  int w_row[IB];                                  // element offset of the W row (fits 31 bits), -1 = beyond N
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int n = n0 + (i * NWV + wave) * 8 + srow;
    w_row[i] = (n < a.N) ? n * a.ldw : -1;
  }
  const int Hv = a.ups ? a.Hin * 2 : a.Hin, Wv = a.ups ? a.Win * 2 : a.Win;  // virtual (upsampled) input dims
  // LDS-DMA pieces per stage and wave: waves below PB % NWV issue one more B piece (wave-uniform)
  constexpr int N_LO = PA / NWV + PB / NWV, N_HI = N_LO + 1, PB_REM = PB % NWV;
  static_assert(PA % NWV == 0, "A pieces must divide evenly over the waves");
  const bool hi_wave = wave < PB_REM;

  const unsigned cc = (unsigned)a.conv_c;
  // Lean tap staging (stride-1 3x3 convs over whole 64-channel slices), see gemm_wide.hip: centre-pixel offset + 9-bit
  // tap-validity mask per staging piece, fixed for the kernel
  const bool leanc = !LEAN && (a.ntaps == 9 || a.phase2x) && a.stride == 1 && a.ups == 0 && !a.pad0 && (a.conv_c % BK) == 0;
  // phase-decomposed upsample conv (GemmArgs::phase2x): segment s of this plane is 3x3-tap position ((s >> 1) + py, (s & 1) + px)
  const int ppy = (!LEAN && a.phase2x) ? (int)(blockIdx.y >> 1) : 0, ppx = (!LEAN && a.phase2x) ? (int)(blockIdx.y & 1) : 0;
  // phase2x == 2 (the DATA GRADIENT of that conv, training): plane (py, px) convolves ITS image of output-gradient pixels (conv_src + plane *
  // a_bs) with the transposed phase weights, tap s at the mirrored position (2 - py - (s >> 1), 2 - px - (s & 1)); plain output rows per plane
  auto tap_of = [&](int seg) {
    return a.phase2x == 1 ? ((seg >> 1) + ppy) * 3 + (seg & 1) + ppx : (a.phase2x == 2 ? (2 - ppy - (seg >> 1)) * 3 + 2 - ppx - (seg & 1) : seg);
  };
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
  // Plain-segment pointers pinned in SGPRs: left to itself hipcc re-loads them from the kernel-argument segment with an
  // s_load + s_waitcnt lgkmcnt(0) in EVERY k-step of a linear layer (it selects the argument offset, not the value).
  // batched launch (GemmArgs::nbatch): grid.y selects the operand / output planes
  const long bz = (long)blockIdx.y;
  const bf16_t* psrc0 = a.p_src[0] + bz * a.a_bs;
  const bf16_t* csrc = a.conv_src + (a.phase2x == 2 ? bz * a.a_bs : 0);
  const bf16_t* psrc1 = a.p_src[1];
  const bf16_t* Wb = a.W + bz * a.w_bs + (a.w_img_bs ? (long)(m0 / a.rows_per_b) * a.w_img_bs : 0L);
  bf16_t* const outb = (bf16_t*)a.out + bz * a.o_bs;
  asm volatile("" : "+s"(psrc0), "+s"(psrc1), "+s"(Wb));
  auto glds = [&](const bf16_t* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // One stage = this wave's pieces of the A tile (gathered) and the W tile.  Straight-line code:
  // predicates select between the real address and the zero page, the segment kind is the only branch.
  auto issue_stage = [&](const KIter& it, int buf) {
    unsigned char* As = smem + buf * STAGE + wave * 1024;
    unsigned char* Bs = As + A_BYTES;
    const int ch = it.c0 + sslot * 8;                 // channel of this lane's 16-byte chunk
    const bool kin = ch < it.seglen;
    if (it.seg < a.ntaps && leanc) {
      const int tp = tap_of(it.seg);
      const int ky = tp / 3, kx = tp - ky * 3;
      const unsigned delta = (unsigned)(((ky - 1) * a.Win + (kx - 1)) * (int)cc + it.c0);     // wave-uniform
      const unsigned bit = 1u << tp;
#pragma unroll
      for (int i = 0; i < IA; ++i)
        glds((c_mask[i] & bit) ? csrc + (c_pre[i] + delta) : a.zero, As + i * NWV * 1024);
    } else if (it.seg < a.ntaps) {
      const int tp = tap_of(it.seg);
      const int ky = tp / 3, kx = tp - ky * 3;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const int yy = a_y[i] + ky, xx = a_x[i] + kx;
        // ups == 1: nearest-2x upsampled input; ups == 2: zero-inserted input (only even virtual pixels exist) --
        // the data-gradient of a stride-2 conv is a stride-1 conv of the zero-inserted output gradient
        const bool ok = kin & (a_pix[i] >= 0) & ((unsigned)yy < (unsigned)Hv) & ((unsigned)xx < (unsigned)Wv) &
                        ((a.ups != 2) | (((yy | xx) & 1) == 0));
        const int sy = a.ups ? (yy >> 1) : yy, sx = a.ups ? (xx >> 1) : xx;
        const unsigned off = (unsigned)(a_bbase[i] + sy * a.Win + sx) * cc + (unsigned)ch;
        glds(ok ? csrc + off : a.zero, As + i * NWV * 1024);
      }
    } else {
      const int ps = it.seg - a.ntaps;
      const bf16_t* base = ps == 0 ? psrc0 : psrc1;
      const unsigned pc = (unsigned)it.seglen;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const bool ok = kin & (a_pix[i] >= 0);
        const unsigned off = (unsigned)a_pix[i] * pc + (unsigned)ch;
        glds(ok ? base + off : a.zero, As + i * NWV * 1024);
      }
    }
    const unsigned wc = (unsigned)(it.wcol + sslot * 8);
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      if (i * NWV + wave >= PB) continue;             // wave-uniform
      const bool ok = kin & (w_row[i] >= 0);
      glds(ok ? Wb + ((unsigned)w_row[i] + wc) : a.zero, Bs + i * NWV * 1024);
    }
  };

  // LEAN staging: per-piece 32-bit BYTE OFFSETS from two block-uniform (SGPR) bases -- the A segment being walked and the W plane -- so a
  // piece costs one VGPR, not a 64-bit pointer plus a stride (the pointer form spilled 12-20 VGPRs of the eight-wave 128 x 160 tile into
  // its k-loop under the 128-VGPR bound of two workgroups per CU).  Rows beyond M / N are CLAMPED to the last valid row instead of
  // being pointed at the zero page: they produce finite garbage in accumulator rows / columns that no epilogue stores (every store
  // path tests m < M and n < N; statistics are only written for rows < M on whole column tiles).  The launcher checks that both operands
  // span less than 4 GB.  One or two plain segments of whole 64-channel slices: the W columns of the two are contiguous, the A base is
  // switched once, where the first segment ends (the [GEGLU output | h2] operand of the folded ff.net.2 . proj_out linear)
  unsigned lo_a[IA], lo_w[IB];
  const bf16_t* abase = psrc0;
  unsigned a_step = BK * 2, w_step = BK * 2;         // bytes per k-step (uniform)
  int lean_left = 0x7fffffff;                        // k-steps left in the segment the A offsets walk
  if (LEAN) {
    const int steps0 = a.p_c[0] / BK;
    const bool in0 = ks_begin < steps0 || a.nplain < 2;
    const unsigned ka = (unsigned)(in0 ? ks_begin : ks_begin - steps0) * BK + (unsigned)sslot * 8;
    const unsigned k0 = (unsigned)ks_begin * BK + (unsigned)sslot * 8;
    if (in0 && a.nplain == 2) lean_left = steps0 - ks_begin;
    abase = in0 ? psrc0 : psrc1;
    const unsigned pc = (unsigned)(in0 ? a.p_c[0] : a.p_c[1]);
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const unsigned row = (unsigned)min(m0 + (i * NWV + wave) * 8 + srow, a.M - 1);
      lo_a[i] = (row * pc + ka) * 2u;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int n = min(n0 + (i * NWV + wave) * 8 + srow, a.N - 1);
      lo_w[i] = ((unsigned)n * (unsigned)a.ldw + k0) * 2u;
      if (a.w_blocked)                                 // [N / 16][ldw / 64][16][64]: the next k-step of a row is 2 KB further on
        lo_w[i] = (((unsigned)(n >> 4) * (unsigned)(a.ldw >> 6) + (unsigned)ks_begin) * 1024u + (unsigned)((n & 15) * 64 + sslot * 8)) * 2u;
    }
    if (a.w_blocked) w_step = 2048;
  }
  asm volatile("" : "+s"(abase));
  auto issue_lean = [&](int buf) {
    unsigned char* As = smem + buf * STAGE + wave * 1024;
    unsigned char* Bs = As + A_BYTES;
    if (lean_left == 0) {                            // block-uniform: the second segment starts here
      abase = psrc1;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const unsigned row = (unsigned)min(m0 + (i * NWV + wave) * 8 + srow, a.M - 1);
        lo_a[i] = (row * (unsigned)a.p_c[1] + (unsigned)sslot * 8) * 2u;
      }
      lean_left = 0x7fffffff;
    }
    --lean_left;
#pragma unroll
    for (int i = 0; i < IA; ++i) { glds((const bf16_t*)((const char*)abase + lo_a[i]), As + i * NWV * 1024); lo_a[i] += a_step; }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      if (i * NWV + wave >= PB) continue;             // wave-uniform
      glds((const bf16_t*)((const char*)Wb + lo_w[i]), Bs + i * NWV * 1024); lo_w[i] += w_step;
    }
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;

  // Staged epilogue (plain bf16 outputs): the residual tile is fetched into registers while the k-loop runs and the output
  // tile goes through LDS so that every global store is a full 16-byte piece of a contiguous row (the fragment layout
  // alone writes 32-byte runs of 128-byte lines).  The residual loads are issued AFTER the pipeline prologue: vmcnt retires
  // in order, so issued in front of it they sat between the first k-step and its operands (in-kernel stamps: 5.8k instead of
  // 1.6k cycles from the prologue to the first MFMA); behind it only the first wait has to let them stay in flight (+NRES).
  constexpr int RS = BN * 2 + 16;                  // LDS row stride of the staged output tile (bytes)
  constexpr int NRES = FM * FN;                    // residual loads per lane
  static_assert(WEPI || BM * RS <= 2 * STAGE, "staged output tile must fit the pipeline buffers");
  static_assert((NSTAGE - 1) * N_HI + NRES <= 63, "vmcnt is a 6-bit counter");
  // ... and the per-batch TRANSPOSED bf16 output (attention's V^T) as well when the tile lies inside one batch element: the tile is
  // staged transposed and leaves as 16-byte pieces of the [channel][pixel] rows (the fragment layout alone writes 2-byte elements:
  // 40 store instructions of 128 bytes per lane set)
  const int tb_ = m0 / a.rows_per_b;
  // second destination (GemmArgs::out2): the column tiles from n_split on are a transposed output of their own
  const bool part2 = a.out2 != nullptr && n0 >= a.n_split;                      // block-uniform
  const int omode = part2 ? (int)OUT_BF16_T : a.out_mode;
  bf16_t* const tbase = part2 ? (bf16_t*)a.out2 : (bf16_t*)a.out;               // base / row pitch / channel count / first channel of the
  const int tld = part2 ? a.ld_out2 : a.ld_out;                                  // transposed destination this tile writes (if any)
  const int tN = part2 ? a.N - a.n_split : a.N, tn0 = part2 ? a.n_split : 0;
  const bool tr = !WEPI && a.ksplit == 1 && omode == OUT_BF16_T && a.act != ACT_GEGLU && !a.resid && (tld & 7) == 0 &&
                  (a.rows_per_b & 7) == 0 && tb_ == (min(m0 + BM, a.M) - 1) / a.rows_per_b;
  const bool staged = tr || (!WEPI && a.ksplit == 1 && omode == OUT_BF16 && a.act != ACT_GEGLU && (a.N & 7) == 0 && (a.ld_out & 7) == 0);
  const bool staged_geglu = !WEPI && a.ksplit == 1 && a.act == ACT_GEGLU && (a.ld_out & 7) == 0;
  const bool res_pre = staged && a.resid != nullptr;
  uint2 rpre[FM][FN];
  auto fetch_resid = [&]() {
    // SGPR base + 32-bit lane offset (the residual tensor is < 4 GB, checked by the launcher): ten 64-bit lane addresses would
    // cost the 8 VGPRs that decide between one and two workgroups per CU.  Out-of-range lanes read element 0 (never stored).
    const unsigned ldr2 = (unsigned)a.ld_res * 2u;
    const unsigned roff0 = (unsigned)(m0 + wm * TM + fr) * ldr2 + (unsigned)(n0 + wn * TN + fg * 4) * 2u;
    const char* rbase = (const char*)a.resid;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int m = m0 + wm * TM + i * 16 + fr, n = n0 + wn * TN + j * 16 + fg * 4;
        const unsigned off = (m < a.M && n < a.N) ? roff0 + (unsigned)(i * 16) * ldr2 + (unsigned)(j * 32) : 0u;
        rpre[i][j] = *(const uint2*)(rbase + off);
      }
  };

  if (nk > 0) {
    KIter it;
    if (!LEAN) it = kiter_at(a, ks_begin);
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
      if (s < nk) {
        if (LEAN) issue_lean(s);
        else {
          if (s) kiter_next(a, it);
          issue_stage(it, s);
        }
        ++issued;
      }
    }
    // the first wait below counts the residual loads as the NEWEST operations in flight (+NRES): pin them behind the prologue's
    // LDS-DMA pieces -- plain global loads and global_load_lds do not alias, so nothing else keeps the scheduler from hoisting them
    __builtin_amdgcn_sched_barrier(0);
    if (res_pre) fetch_resid();
    __builtin_amdgcn_sched_barrier(0);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
      // stages beyond t already in flight may stay in flight: wait only for stage t's pieces
      const int ahead = issued - 1 - t;               // 0 .. NSTAGE-2, block-uniform
      if (t == 0 && res_pre) {                        // ... and so may the residual loads issued behind the prologue
        if (ahead == 0) wait_vmcnt<NRES>();
        else if (ahead == 1) { if (hi_wave) wait_vmcnt<N_HI + NRES>(); else wait_vmcnt<N_LO + NRES>(); }
        else if (ahead == 2 || NSTAGE < 4) { if (hi_wave) wait_vmcnt<2 * N_HI + NRES>(); else wait_vmcnt<2 * N_LO + NRES>(); }
        else { if (hi_wave) wait_vmcnt<3 * N_HI + NRES>(); else wait_vmcnt<3 * N_LO + NRES>(); }
      } else if (ahead == 0) wait_vmcnt<0>();
      else if (ahead == 1) { if (hi_wave) wait_vmcnt<N_HI>(); else wait_vmcnt<N_LO>(); }
      else if (ahead == 2 || NSTAGE < 4) { if (hi_wave) wait_vmcnt<2 * N_HI>(); else wait_vmcnt<2 * N_LO>(); }
      else { if (hi_wave) wait_vmcnt<3 * N_HI>(); else wait_vmcnt<3 * N_LO>(); }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();          // stage t visible to all waves; all waves done with k-step t-1
      asm volatile("" ::: "memory");
      if (issued < nk) {                      // refill the buffer k-step t-1 just released
        int nb = buf - 1; if (nb < 0) nb += NSTAGE;
        if (LEAN) issue_lean(nb);
        else { kiter_next(a, it); issue_stage(it, nb); }
        ++issued;
      }
      const unsigned char* As = smem + buf * STAGE;
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
      if (++buf == NSTAGE) buf = 0;
    }
  }

  // ---------------------------------------------------------------- epilogue
  if constexpr (WEPI) {      // the launcher sends only single-pass bf16 launches here (gemm_big_pick)
    if constexpr (FN % 2 == 0) {
      // GEGLU in registers (the epilogue of gemm_wide.hip's 256 x 128 tile, here for the 256 x 256 eight-wave tile): the packed weight rows
      // interleave values and gates in 16-row blocks, so acc[i][j] / acc[i][j + 1] hold value and gate of the SAME four hidden units of
      // one pixel in one lane; bias (or the folded-LayerNorm fix-up), exact-erf GELU and the product run on the accumulators and only the
      // bf16 result (half the columns) is staged through LDS for full-row 16-byte stores.
      if (a.act == ACT_GEGLU) {
        constexpr int RSG = BN + 16;                     // bf16 row stride of the staged [BM][BN / 2] tile (bytes)
        constexpr int LNROWG = BM * RSG;
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
            const uint2 o = geglu4(acc[i][2 * jj], acc[i][2 * jj + 1], bv[jj], bg[jj], sv[jj], sg[jj], lnf, mr.y, ms, gk);
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
    wide_epilogue<BM, BN, WN, NSTAGE * STAGE>(a, acc, smem, tid, wm, wn, fr, fg, m0, n0);
    return;
  }
  const bool partial = a.ksplit > 1;
  if (staged) {
    if (nk <= 0 && res_pre) fetch_resid();
    // bias / time-embedding slices of this wave's TN columns go through a wave-private LDS strip behind the output tile: ONE
    // global load per lane, then unconditional ds_reads per fragment.  Loaded per fragment inside `if (a.bias)`, hipcc kept
    // every load behind its branch: ten dependent L2 round trips, 4.4-5.5k cycles of a tile whose K = 320 loop takes 6.6k
    // (in-kernel stamps); batched into registers they cost 8 VGPRs too many for two workgroups per CU.
    constexpr int STRIP = 2 * TN * 4;                // bias | rowvec (or the folded-LayerNorm s slice), fp32
    constexpr int RST = BM * 2 + 16;                 // row stride of the TRANSPOSED staged tile ([BN][BM] bf16)
    constexpr int TILE_B = BM * RS > BN * RST ? BM * RS : BN * RST;
    constexpr int LNROW = TILE_B + NWV * STRIP;      // folded LayerNorm: (mean, rstd) of the tile's rows, fp32 pairs
    static_assert(WEPI || LNROW + BM * 8 <= NSTAGE * STAGE, "bias strips + row statistics must fit behind the staged output tile");
    float* strip = (float*)(smem + TILE_B + wave * STRIP);
    const bool lnf = a.ln_stat != nullptr;
    // the time-embedding row is per image: through the strip when the whole tile lies in one image, else per fragment
    const bool rv_lds = a.rowvec != nullptr && (a.rv_ld == 0 || (m0 / a.rows_per_b) == ((min(m0 + BM, a.M) - 1) / a.rows_per_b));   // rv_ld == 0: one row for every image (cached timestep row)
    float4 bq = float4{0.f, 0.f, 0.f, 0.f}, rq = bq;
    if (lane < TN / 4) {
      const int n = n0 + wn * TN + lane * 4;
      if (a.bias && n < a.N) bq = *(const float4*)(a.bias + n);
      if (rv_lds && n < a.N) rq = *(const float4*)(a.rowvec + (long)(m0 / a.rows_per_b) * a.rv_ld + a.rv_off + n);
      if (lnf && n < a.N) rq = *(const float4*)(a.ln_s + n);
    }
    __syncthreads();                                 // every wave is done reading the last pipeline stage
    if (lane < TN / 4) { *(float4*)(strip + lane * 4) = bq; *(float4*)(strip + TN + lane * 4) = rq; }
    if (lnf) {
      if (tid < BM) *(float2*)(smem + LNROW + tid * 8) = lnmr;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm * TM + i * 16 + fr, m = m0 + row;
      const int b = a.rowvec ? min(m, a.M - 1) / a.rows_per_b : 0;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = wn * TN + j * 16 + fg * 4, n = n0 + col;
        const float4 bv = *(const float4*)(strip + j * 16 + fg * 4);
        float v[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
        if (lnf) {                                   // rstd * (acc - mean * s) + b'
          const float2 mr = *(const float2*)(smem + LNROW + row * 8);
          const float4 sv = *(const float4*)(strip + TN + j * 16 + fg * 4);
          const float ms = -mr.x * mr.y;
          v[0] = fmaf(mr.y, acc[i][j][0], fmaf(ms, sv.x, bv.x)); v[1] = fmaf(mr.y, acc[i][j][1], fmaf(ms, sv.y, bv.y));
          v[2] = fmaf(mr.y, acc[i][j][2], fmaf(ms, sv.z, bv.z)); v[3] = fmaf(mr.y, acc[i][j][3], fmaf(ms, sv.w, bv.w));
        }
        if (a.rowvec) {
          float4 rv;
          if (rv_lds) rv = *(const float4*)(strip + TN + j * 16 + fg * 4);
          else rv = n < a.N ? *(const float4*)(a.rowvec + (long)b * a.rv_ld + a.rv_off + n) : float4{0.f, 0.f, 0.f, 0.f};
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
          const uint2 rr = rpre[i][j];
          v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
          v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
        }
        if (tr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) *(bf16_t*)(smem + (col + r) * RST + row * 2) = f2bf(v[r]);
        } else {
          uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
          *(uint2*)(smem + row * RS + col * 2) = o;
        }
      }
    }
    __syncthreads();
    if (tr) {
      constexpr int CPT = BM / 8;                    // 16-byte chunks per transposed row (one output channel, BM pixels)
      const int mm0 = m0 - tb_ * a.rows_per_b;
      for (int c = tid; c < BN * CPT; c += NWV * 64) {
        const int col = c / CPT, cc = c - col * CPT;
        const int n = n0 + col, m = m0 + cc * 8;
        if (n < a.N && m < a.M)                      // M and rows_per_b are multiples of 8 here: whole chunks
          *(uint4*)(tbase + ((long)tb_ * tN + (n - tn0)) * tld + mm0 + cc * 8) = *(const uint4*)(smem + col * RST + cc * 16);
      }
      return;
    }
    constexpr int CPR = BN / 8;                      // 16-byte chunks per tile row
    for (int c = tid; c < BM * CPR; c += NWV * 64) {
      const int row = c / CPR, cc = c - row * CPR;
      const int m = m0 + row, n = n0 + cc * 8;
      long orow = m;
      if (!LEAN && a.phase2x == 1) {                 // source pixel (b, y, x) -> pixel (2y + py, 2x + px) of the 2H x 2W output
        const int b = m / HWo, rem = m - b * HWo;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        orow = (long)b * 4 * HWo + (long)(2 * oy + ppy) * (2 * a.Wout) + 2 * ox + ppx;
      }
      if (m < a.M && n < a.N)
        *(uint4*)(outb + orow * a.ld_out + n) = *(const uint4*)(smem + row * RS + cc * 16);
    }
    if (a.rowstat) {
      // Row statistics of this tile's bf16-rounded outputs for a LayerNorm folded into the consumer (gemm.h): TPR adjacent lanes share a
      // row, two passes over the staged tile (mean, then the centred squares: no cancellation), fixed orders.  The launcher guarantees
      // N % BN == 0 (whole column tiles).
      constexpr int TPR = NWV * 64 / BM;             // 4 (eight-wave 128-row tile) or 2
      static_assert(TPR == 2 || TPR == 4, "threads per row");
      const int row = tid / TPR, part = tid % TPR;
      const unsigned char* src = smem + row * RS;
      float sum = 0.f;
      for (int c = part; c < CPR; c += TPR) {
        float f[8]; unpack8(*(const uint4*)(src + c * 16), f);
#pragma unroll
        for (int r = 0; r < 8; ++r) sum += f[r];
      }
      sum += __shfl_xor(sum, 1, 64);
      if (TPR == 4) sum += __shfl_xor(sum, 2, 64);
      const float mean = sum * (1.0f / BN);
      float m2 = 0.f;
      for (int c = part; c < CPR; c += TPR) {
        float f[8]; unpack8(*(const uint4*)(src + c * 16), f);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const float d = f[r] - mean; m2 = fmaf(d, d, m2); }
      }
      m2 += __shfl_xor(m2, 1, 64);
      if (TPR == 4) m2 += __shfl_xor(m2, 2, 64);
      const int m = m0 + row;
      if (part == 0 && m < a.M) *(float2*)(a.rowstat + ((long)nt_ * a.M + m) * 2) = float2{mean, m2};
    }
    if constexpr (BM == 128 && BN == 160 && !WEPI && WM * WN == 8) {      // tid < 2 * BN needs the eight-wave block (320 <= 512 threads)
      static_assert(WM * WN * 64 >= 2 * BN, "the statistics epilogue indexes 2 * BN threads");
      if (a.gstat) {
        // GroupNorm statistics of this tile's bf16-rounded outputs for the consumer (gemm.h GemmArgs::gstat; round 5: the 128-row tile too,
        // so that the 32x32-level producers leave them and gn_stats_kernel disappears there).  As in the 256-row epilogue: per column the
        // sum / sum of squares of two row halves, the halves in order, then the columns of a group in order -- fixed orders, reruns are
        // bit-identical.  The launcher guarantees whole tiles inside one image and BN % cpg == 0.
        constexpr int GST_OFF = LNROW + BM * 8;
        static_assert(GST_OFF + 3 * BN * 8 <= NSTAGE * STAGE, "statistics scratch must fit behind the staged output tile");
        float* qrt = (float*)(smem + GST_OFF);         // [2 row halves][BN][2]
        float* cst = qrt + 2 * BN * 2;                 // [BN][2]
        if (tid < 2 * BN) {
          const int rq = tid / BN, col = tid - rq * BN;
          const unsigned char* src = smem + (rq * (BM / 2)) * RS + col * 2;
          float ss = 0.f, qq = 0.f;
          for (int r = 0; r < BM / 2; ++r) {
            const float v = bf2f(*(const bf16_t*)(src + r * RS));
            ss += v; qq = fmaf(v, v, qq);
          }
          qrt[(rq * BN + col) * 2] = ss; qrt[(rq * BN + col) * 2 + 1] = qq;
        }
        __syncthreads();
        if (tid < BN) {
          cst[tid * 2] = qrt[tid * 2] + qrt[(BN + tid) * 2];
          cst[tid * 2 + 1] = qrt[tid * 2 + 1] + qrt[(BN + tid) * 2 + 1];
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
    return;
  }
  if (staged_geglu) {
    // value * gelu(gate) on 16-column value/gate blocks: the tile's BN packed columns give BN/2 outputs per row
    const bool lnf = a.ln_stat != nullptr;
    constexpr int LNROWG = BM * RS;
    static_assert(WEPI || LNROWG + BM * 8 <= NSTAGE * STAGE, "row statistics must fit behind the staged output tile");
    __syncthreads();
    if (lnf) {
      if (tid < BM) *(float2*)(smem + LNROWG + tid * 8) = lnmr;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm * TM + i * 16 + fr;
      float2 mr = float2{0.f, 1.f};
      if (lnf) mr = *(const float2*)(smem + LNROWG + row * 8);
      const float ms = -mr.x * mr.y;
#pragma unroll
      for (int j = 0; j + 1 < FN; j += 2) {
        const int n = n0 + wn * TN + j * 16 + fg * 4;      // packed column of the "value" half
        float4 bv = float4{0, 0, 0, 0}, bg = float4{0, 0, 0, 0};
        if (a.bias && n + 16 < a.N) { bv = *(const float4*)(a.bias + n); bg = *(const float4*)(a.bias + n + 16); }
        if (lnf && n + 16 < a.N) {                         // rstd * (acc - mean * s) + b' == rstd * acc + (b' - rstd * mean * s)
          const float4 sv = *(const float4*)(a.ln_s + n), sg = *(const float4*)(a.ln_s + n + 16);
          bv.x = fmaf(ms, sv.x, bv.x); bv.y = fmaf(ms, sv.y, bv.y); bv.z = fmaf(ms, sv.z, bv.z); bv.w = fmaf(ms, sv.w, bv.w);
          bg.x = fmaf(ms, sg.x, bg.x); bg.y = fmaf(ms, sg.y, bg.y); bg.z = fmaf(ms, sg.z, bg.z); bg.w = fmaf(ms, sg.w, bg.w);
        }
        const float v[4] = {fmaf(mr.y, acc[i][j][0], bv.x), fmaf(mr.y, acc[i][j][1], bv.y), fmaf(mr.y, acc[i][j][2], bv.z), fmaf(mr.y, acc[i][j][3], bv.w)};
        const float g[4] = {fmaf(mr.y, acc[i][j + 1][0], bg.x), fmaf(mr.y, acc[i][j + 1][1], bg.y), fmaf(mr.y, acc[i][j + 1][2], bg.z),
                            fmaf(mr.y, acc[i][j + 1][3], bg.w)};
        const int ocl = ((wn * TN) >> 1) + (j >> 1) * 16 + fg * 4;      // output column inside the tile
        uint2 o;
        o.x = pack2bf(v[0] * gelu_erf_f(g[0]), v[1] * gelu_erf_f(g[1]));
        o.y = pack2bf(v[2] * gelu_erf_f(g[2]), v[3] * gelu_erf_f(g[3]));
        *(uint2*)(smem + row * RS + ocl * 2) = o;
      }
    }
    __syncthreads();
    constexpr int CPRG = BN / 16;                    // 16-byte chunks per output row of the tile
    for (int c = tid; c < BM * CPRG; c += NWV * 64) {
      const int row = c / CPRG, cc = c - row * CPRG;
      const int m = m0 + row, oc = (n0 >> 1) + cc * 8;
      if (m < a.M && oc < (a.N >> 1))
        *(uint4*)((bf16_t*)a.out + (long)m * a.ld_out + oc) = *(const uint4*)(smem + row * RS + cc * 16);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * TM + i * 16 + fr;
    if (m >= a.M) continue;
    const int b = (a.rowvec || omode == OUT_BF16_T || omode == OUT_F32_T) ? m / a.rows_per_b : 0;
    float2 lmr = float2{0.f, 1.f};
    if (a.ln_stat) lmr = ln_row_stats(a, m);              // the launcher refuses ln_stat with split-K or the GEGLU branch below
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
      if (a.ln_stat) {
        const float4 sv = *(const float4*)(a.ln_s + n);
        const float ms = -lmr.x * lmr.y;
        v[0] = fmaf(lmr.y, v[0], ms * sv.x); v[1] = fmaf(lmr.y, v[1], ms * sv.y);
        v[2] = fmaf(lmr.y, v[2], ms * sv.z); v[3] = fmaf(lmr.y, v[3], ms * sv.w);
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
      if (omode == OUT_BF16) {
        uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
        *(uint2*)((bf16_t*)a.out + (long)m * a.ld_out + n) = o;
      } else if (omode == OUT_F32) {
        *(float4*)((float*)a.out + (long)m * a.ld_out + n) = float4{v[0], v[1], v[2], v[3]};
      } else if (omode == OUT_BF16_T) {
        const int mm = m - b * a.rows_per_b;
        bf16_t* o = tbase + ((long)b * tN + (n - tn0)) * tld + mm;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(long)r * tld] = f2bf(v[r]);
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

template <int BM, int BN, int WM, int WN, int NSTAGE, bool LEAN = false, bool WEPI = false>
int launch_tile(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = NSTAGE * (BM + BN) * BK * 2;
  static_assert(lds <= 160 * 1024, "LDS ring exceeds the CU's 160 KiB");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN, WM, WN, NSTAGE, LEAN, WEPI>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, a.nbatch > 1 ? a.nbatch : 1, a.ksplit);
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, NSTAGE, LEAN, WEPI>), grid, dim3(WM * WN * 64), lds, stream, a);
  return dfh::check_launch("gemm_bf16_kernel");
}

struct TileInfo { int bm, bn; };
// variant ids (force_tile - 1):
//   0: 256x160 8 waves 3 stages   1: 256x128 8 waves 3 stages   2: 128x64 4 waves 3 stages
//   3: 128x160 4 waves 2 stages   4: 128x128 4 waves 2 stages
//   5: 128x160 EIGHT waves (4 x 2, 32 x 80 each) 2 stages, external id 10 (ids 6-9 are the wide kernel) -- the default
//      128 x 160 kernel: issuing a k-step's 36 LDS-DMA pieces costs a wave ~100 cycles apiece during which it issues no
//      MFMA, so with one wave per SIMD (a 4-wave workgroup alone on its CU: the 256-tile launches of the 16x16 level) a
//      k-step takes 0.93 us against 0.29 us of MFMA work; two waves per SIMD overlap the two (0.75 us; 12.6 / 25.0 / 69.2
//      vs 15.0 / 28.3 / 78.5 us on 4096 x 1280 x {320, 1280, 5120}), and at two workgroups per CU (122 VGPRs: 16 waves
//      fit) it still wins 4-5 % (scripts/gemm_deepring_probe.py).  A 4-stage ring on the 4-wave tile gained nothing.
constexpr TileInfo kTiles[6] = {{256, 160}, {256, 128}, {128, 64}, {128, 160}, {128, 128}, {128, 160}};
constexpr int kNumTiles = 5;                      // ids 1..5 of force_tile; the eight-wave 128 x 160 variant is id 10
constexpr int kEightWave = 5;

// one plain K segment of a multiple of 64 channels, W rows long enough: the LEAN k-loop applies
bool lean_plain(const GemmArgs& a) {
  // same-box A/B (scripts/gemm_lean_probe.py): 4096 x 1280 x {1280, 5120} 26.8 -> 25.7 / 77 -> 71 us (one workgroup per CU),
  // 0-3 % at two workgroups per CU -- the LDS-DMA issue itself (~100 cycles per 1-KB piece), not its address arithmetic,
  // is what paces the loop
  if (a.ntaps != 0 || a.nplain < 1 || a.p_c[0] % BK != 0 || a.p_c[0] <= 0) return false;
  // 32-bit byte offsets from the segment / weight-plane bases inside the kernel
  const double amax = (double)a.M * std::max(a.p_c[0], a.nplain > 1 ? a.p_c[1] : 0) * 2.0, wmax = (double)a.N * a.ldw * 2.0;
  if (amax >= 4.0e9 || wmax >= 4.0e9) return false;
  return a.nplain == 1 || (a.p_c[1] % BK == 0 && a.p_c[1] > 0);
}

// the "big" tile: 256 x 320, eight waves, 64-deep k-steps (see the kernel's WEPI note)
// the GEGLU projections on a 256 x 256 eight-wave tile (128 x 64 per wave: whole (value, gate) block pairs inside a wave)
int launch_big_geglu(const GemmArgs& a, hipStream_t s) {
  return lean_plain(a) ? launch_tile<256, 256, 2, 4, 2, true, true>(a, s) : launch_tile<256, 256, 2, 4, 2, false, true>(a, s);
}

int launch_big(const GemmArgs& a, hipStream_t s) {
  // (a 128 x 320 sibling for the 32x32 level -- one tile per CU there too -- was measured and dropped: equal to the eight-wave
  //  128 x 160 kernel in isolation, conv3x3 class +0.3 ms per step in situ: profiles/r03/big_tile_probe.txt)
  return lean_plain(a) ? launch_tile<256, 320, 2, 4, 2, true, true>(a, s) : launch_tile<256, 320, 2, 4, 2, false, true>(a, s);
}

int launch_variant(int tile, const GemmArgs& a, hipStream_t s) {
  switch (tile) {
    case 0: return launch_tile<256, 160, 4, 2, 3>(a, s);
    case 1: return launch_tile<256, 128, 4, 2, 3>(a, s);
    case 2: return launch_tile<128, 64, 2, 2, 3>(a, s);
    case 3: return launch_tile<128, 160, 2, 2, 2>(a, s);
    case 5: {
      // launches of at most ~one workgroup per CU (the 8x8-level Winograd GEMM: 16 planes x 256 rows = 256 workgroups; the 16x16-level token
      // linears), each a chain of k-steps that wait for lines requested one stage ahead: a 4-stage ring keeps three stages in flight.
      // DFH_DEEP4=0 turns it off (A/B).
      static const bool deep4_off = [] { const char* e = getenv("DFH_DEEP4"); return e && e[0] == '0'; }();
      static const bool deep4_all = [] { const char* e = getenv("DFH_DEEP4"); return !(e && e[0] == '1'); }();     // 1: batched launches only (A/B: 16.32 -> 16.25 ms with single launches too)
      const long wgs = (long)((a.M + 127) / 128) * ((a.N + 159) / 160) * (a.nbatch > 1 ? a.nbatch : 1) * a.ksplit;
      if (!deep4_off && (a.nbatch > 1 || deep4_all) && wgs <= 320) return lean_plain(a) ? launch_tile<128, 160, 4, 2, 4, true>(a, s) : launch_tile<128, 160, 4, 2, 4>(a, s);
      return lean_plain(a) ? launch_tile<128, 160, 4, 2, 2, true>(a, s) : launch_tile<128, 160, 4, 2, 2>(a, s);
    }
    default: return launch_tile<128, 128, 2, 2, 2>(a, s);
  }
}

}  // namespace

namespace dfh {

int gemm_count_ksteps(const GemmArgs& a) {
  int n = a.ntaps * ((a.conv_c + BK - 1) / BK);
  for (int i = 0; i < a.nplain; ++i) n += (a.p_c[i] + BK - 1) / BK;
  return n;
}

int gemm_pick_split(const GemmArgs& a, int* tile_out) {
  // column tile: 160 when it divides N (320/640/1280/...), else 128 (GEGLU needs 32-aligned pairs), 64 for N <= 64
  const bool geglu = a.act == ACT_GEGLU;
  const bool n160 = !geglu && (a.N % 160 == 0 || (a.N % 128 != 0 && a.N > 128));
  // Measured on MI355X (scripts/gemm_microbench.py): the 128-row / 4-wave / 2-stage variants at two
  // workgroups per CU beat the 256-row / 8-wave / 3-stage ring on every U-Net shape except the 8x8 level.
  int tile;
  if (a.N <= 64 && !geglu) tile = 2;
  else if (a.M > 128 && a.M <= 1024 && gemm_count_ksteps(a) >= 64) tile = n160 ? 0 : 1;
  else tile = n160 ? kEightWave : 4;   // same-box A/B over the whole step: linear class 7.65 -> 7.33 ms, conv3x3 6.92 -> 6.87 ms
  const TileInfo ti = kTiles[tile];
  const int blocks = ((a.M + ti.bm - 1) / ti.bm) * ((a.N + ti.bn - 1) / ti.bn);
  const int ksteps = gemm_count_ksteps(a);
  int split = 1;
  if (!geglu && blocks < 384 && ksteps >= (a.ntaps ? 64 : 160)) {     // plain K = 5120 (80 k-steps) loses: 67 -> 72 us
    // deep-K launches that leave CUs idle or at one workgroup each (16x16 level at batch 16: 256 tiles; 8x8 level: 64-128):
    // split K until about 512 workgroups are resident.  The slab round trip pays for itself on the 3x3 convs
    // (M=4096: 205 -> 149 us with 2 slices; M=2048: 111 -> 78 us with 4; scripts/gemm_split_probe*.py)
    const int target = ti.bm == 256 ? 256 : 512;        // the 8-wave 256-row tiles run one workgroup per CU
    split = std::max(1, std::min((target + blocks / 2) / blocks, ksteps / 16));
  } else if (!geglu && blocks < 160 && ksteps >= 16) {
    // shallow K (1x1 / linear): more than two slices cost more in slab traffic than they win
    split = std::min({(256 + blocks - 1) / blocks, ksteps / 8, 64});
    if (split < 1) split = 1;
  }
  if (tile_out) *tile_out = tile;
  return split;
}

size_t gemm_partial_floats(const GemmArgs& a) {
  if (a.nbatch > 1) return 0;     // batched launches never split K
  int tile;
  const int s = gemm_pick_split(a, &tile);
  return s > 1 ? (size_t)s * a.M * a.N : 0;
}

// Tile order of a launch (dfh_common.h tile_coords) for a bm x bn tile.  DFH_TMAP="xm,gm" pins it for every launch (probe).
// The rules are the measured ones (scripts/pmc_traffic_calib.sh + scripts/tile_order_probe.py, profiles/r02): launch TIME does
// not depend on the order (+-2 %: the over-fetched bytes come out of the Infinity Cache), fabric traffic does --
//   * many column tiles over a weight matrix that cannot stay in one XCD's L2 (GEGLU projections at the 32x32 / 16x16 levels:
//     40 / 80 column tiles, 6.5 / 26 MB of weights): groups of 8 row tiles walked column by column, FETCH 11-12 x -> 5 x the
//     algorithmic bytes;
//   * single-pass 3x3 convs with few column tiles and big weights (32x32 level: 128 x 4 tiles, 7-22 MB): a 4 x 2 grid of XCDs
//     (each XCD streams half of the weights instead of all of them), 3.0 / 4.2 x -> 2.5 / 3.1 x; not at the 64x64 level, where
//     splitting the column tiles over XCDs doubles the (20 x larger) pixel traffic.
static void gemm_pick_tile_order(GemmArgs& a, int split, int bm, int bn) {
  static const int env_xm = [] { const char* e = getenv("DFH_TMAP"); return e ? atoi(e) : -1; }();
  static const int env_gm = [] { const char* e = getenv("DFH_TMAP"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }();
  a.tm_xm = 0; a.tm_gm = 0;
  if (env_xm >= 0) {            // probe knob; xm must divide the 8 XCDs (anything else would enumerate some tiles twice and others never)
    a.tm_xm = (env_xm == 1 || env_xm == 2 || env_xm == 4 || env_xm == 8) ? env_xm : 0; a.tm_gm = env_gm; return;
  }
  if (split > 1 || a.n_major) return;
  double kk = (double)a.ntaps * a.conv_c;
  for (int i = 0; i < a.nplain; ++i) kk += a.p_c[i];
  const double w_bytes = (double)a.N * kk * 2.0;
  const double a_bytes = a.ntaps ? (double)(a.M / (a.Hout * a.Wout)) * a.Hin * a.Win * a.conv_c * 2.0 : (double)a.M * kk * 2.0;
  const int ntm = (a.M + bm - 1) / bm, ntn = (a.N + bn - 1) / bn;
  if (w_bytes <= 2.0e6 || ntm < 16) return;          // the weights stay resident in every L2: nothing to order
  if (ntn >= 16) { a.tm_gm = 8; return; }
  if (!a.ntaps) return;
  // few column tiles: an xm x (8 / xm) grid of XCDs fetches (8 / xm) x the pixels + xm x the weights in total (xm = 8 is the
  // legacy order: every XCD streams all the weights).  Measured 4 x 2 against 8 x 1: better on the 32x32-level convs (21 + 8 x 7
  // MB -> 2 x 21 + 4 x 7), WORSE at 64x64 where the pixels outweigh the weights 20 : 1 -- so pick the minimum of the model.
  int best = 8; double cost = a_bytes + 8.0 * w_bytes;
  for (int xm = 4; xm >= 2; xm >>= 1) {
    if (ntm % xm || ntn % (8 / xm)) continue;
    const double c = (8.0 / xm) * a_bytes + xm * w_bytes;
    if (c < 0.95 * cost) { cost = c; best = xm; }
  }
  if (best != 8) { a.tm_xm = best; a.tm_gm = 8; }
}

// Launches for the 256 x 320 tile: one workgroup per CU, so the grid has to come out at whole rounds of the 256 CUs (a 257th tile
// would run alone for a whole round).  DFH_GEMM_BIG=0 turns it off, =2 also sends the plain linears there (A/B).
int gemm_big_pick(const GemmArgs& a) {
  static const int mode = [] { const char* e = getenv("DFH_GEMM_BIG"); return e ? atoi(e) : 2; }();   // 0 off, 1 convs, 2 + deep linears, 3 + all linears (A/B)
  if (mode == 0) return 0;
  if (a.out_mode != OUT_BF16 || a.act == ACT_GEGLU || a.ln_stat) return 0;
  if (a.N % 320 != 0 || (a.ld_out & 7) || (a.resid && (a.ld_res & 7))) return 0;
  const int ksteps = gemm_count_ksteps(a);
  if (ksteps < 16 && (a.ntaps || mode < 3)) return 0;   // conv_in (K = 72): prologue + four-pass epilogue outweigh two k-steps (34.6 vs 24.3 us)
  if (a.ntaps == 0 && mode < 2) return 0;
  const long tiles = (long)((a.M + 255) / 256) * (a.N / 320), rem = tiles % 256;
  return (tiles >= 224 && (rem == 0 || rem >= 224 || tiles >= 1024)) ? 1 : 0;
}

// GEGLU projections for the 256 x 256 tile: whole rounds of the CUs, as above.  DFH_GEMM_BIGG=0 turns it off (A/B).
int gemm_big_geglu_pick(const GemmArgs& a) {
  static const int mode = [] { const char* e = getenv("DFH_GEMM_BIGG"); return e ? atoi(e) : 1; }();
  if (mode == 0 || a.act != ACT_GEGLU || a.out_mode != OUT_BF16 || a.resid || a.rowvec) return 0;
  if (a.N % 256 != 0 || (a.ld_out & 7)) return 0;
  // (16x16 level: 640 tiles = 2.5 rounds, still 5 % ahead of the 256 x 128 tile in isolation: profiles/r03/geglu_tile_probe.txt)
  const long tiles = (long)((a.M + 255) / 256) * (a.N / 256), rem = tiles % 256;
  return (tiles >= 224 && (rem == 0 || rem >= 224 || tiles >= 512)) ? 1 : 0;
}

bool wino_blocked(int N, int C) {
  static const bool off = [] { const char* e = getenv("DFH_W_BLOCKED"); return e && e[0] == '0'; }();      // A/B
  return !off && N % 160 == 0 && C % 64 == 0;          // the batched launch then runs on a LEAN instantiation (128 x 160 eight-wave / 256 x 320)
}

int wino_gemm_tile(const GemmArgs& a) {
  static const int pin = [] { const char* e = getenv("DFH_WINO_TILE"); return e ? atoi(e) : -1; }();      // probe
  if (pin >= 0) return pin;
  return 0;
}

// Batched launches (Winograd planes, phase planes of an upsample conv) for the 256 x 320 eight-wave tile: planes of at least 512 rows whose
// tiles together make whole rounds of the CUs (16x16-level Winograd: 16 planes x 16 tiles = 256 workgroups, 78.7 us on the 128 x 160 tile -> 67.1 us).
// DFH_BATCH_BIG=0 turns it off (A/B).
static bool batched_big_pick(const GemmArgs& a) {
  static const bool off = [] { const char* e = getenv("DFH_BATCH_BIG"); return e && e[0] == '0'; }();
  if (off || a.nbatch <= 1 || a.M < 512 || a.N % 320 != 0) return false;
  const long tiles = (long)((a.M + 255) / 256) * (a.N / 320) * a.nbatch, rem = tiles % 256;
  return tiles >= 224 && (rem == 0 || rem >= 224 || tiles >= 1024);
}

bool gemm_out2_ok(GemmArgs a) {
  if (a.rows_per_b <= 0) a.rows_per_b = a.M;
  a.ksteps = gemm_count_ksteps(a);
  int tile;
  if (gemm_pick_split(a, &tile) != 1) return false;
  return a.out2 != nullptr && a.n_split > 0 && a.n_split < a.N && a.n_split % kTiles[tile].bn == 0 && a.act != ACT_GEGLU &&
         a.out_mode == OUT_BF16 && !a.resid;
}

bool gemm_ln_consumer_ok(GemmArgs a) {
  if (a.rows_per_b <= 0) a.rows_per_b = a.M;
  a.ksteps = gemm_count_ksteps(a);
  int tile;
  if (gemm_pick_split(a, &tile) != 1) return false;                     // the split-K reduce does not implement the fix-up
  if (a.act == ACT_GEGLU && ((a.ld_out & 7) || a.N % 32)) return false;  // ... nor does the unstaged GEGLU branch
  return a.ntaps == 0 && a.resid == nullptr && a.rowvec == nullptr;
}

int gemm_launch(GemmArgs a, hipStream_t stream, int force_tile, int force_split, int force_order, int* gstat_rows, int* rowstat_bn) {
  if (gstat_rows) *gstat_rows = 0;
  if (rowstat_bn) *rowstat_bn = 0;
  DFH_REQUIRE(a.M > 0 && a.N > 0, "empty GEMM");
  DFH_REQUIRE(a.N % 4 == 0, "N must be a multiple of 4");
  DFH_REQUIRE(a.ntaps == 0 || a.ntaps == 9 || (a.ntaps == 4 && a.phase2x), "ntaps must be 0 or 9 (4 for the phase planes of an upsample conv)");
  DFH_REQUIRE(a.ntaps + a.nplain >= 1, "no K segment");
  DFH_REQUIRE(a.ntaps == 0 || a.conv_c % 8 == 0, "conv channels must be a multiple of 8");
  for (int i = 0; i < a.nplain; ++i) DFH_REQUIRE(a.p_c[i] % 8 == 0, "segment length must be a multiple of 8");
  DFH_REQUIRE(a.zero != nullptr, "zero page missing");
  if (a.rows_per_b <= 0) a.rows_per_b = a.M;
#ifdef DFH_PROBES
  if (force_tile == 30) {           // probe kernel: the launch as a register-resident token linear (scripts/probes/kernels/token_linear.hip)
    if (rowstat_bn && a.rowstat) *rowstat_bn = a.N;
    return token_linear_from_gemm(a, stream);
  }
#endif
  a.ksteps = gemm_count_ksteps(a);
  int tile;
  int split = gemm_pick_split(a, &tile);
  // microbench / tests: tile id 6 = the 256 x 160 wide kernel, 7 = its 128 x 160 sibling
  const bool force_deep = force_tile == 10;
  if (force_deep) { tile = kEightWave; force_tile = 0; }
  int force_wide = force_tile > kNumTiles ? force_tile - kNumTiles : 0;
  if (force_wide) force_tile = 0;
  if (force_tile > 0) { DFH_REQUIRE(force_tile <= kNumTiles, "unknown tile variant"); tile = force_tile - 1; }
  if (force_split > 0) split = force_split;
  if (a.act == ACT_GEGLU) {
    DFH_REQUIRE(a.N % 32 == 0 && kTiles[tile].bn != 160 && split == 1,
                "GEGLU needs N % 32 == 0, a 64/128-wide tile and no split-K");
    DFH_REQUIRE(a.out_mode == OUT_BF16 && !a.resid && !a.rowvec, "GEGLU epilogue is bias-only, bf16 out");
  }
  split = std::min(split, a.ksteps);
  a.ksplit = split;
  {
    double kk = (double)a.ntaps * a.conv_c;
    for (int i = 0; i < a.nplain; ++i) kk += a.p_c[i];
    const double w_bytes = (double)a.N * kk, a_bytes = (double)a.M * (a.ntaps ? (double)a.conv_c : kk);
    // measured (scripts/gemm_nmajor_probe.py): +12 % / +6 % on the 16x16-level 3x3 convs, -4 % on the linear shapes -> convs only
    // batched planes (Winograd): each plane is its own weight-heavy GEMM -- same rule (DFH_BATCH_NMAJOR=0: m-major, A/B)
    static const bool bn_off = [] { const char* e = getenv("DFH_BATCH_NMAJOR"); return e && e[0] == '0'; }();
    const bool conv_like = a.ntaps || (a.nbatch > 1 && !bn_off);
    a.n_major = (force_order == 2 || (force_order < 0 && conv_like && w_bytes > a_bytes && a.N > 160)) ? 1 : 0;   // force_order 2 / 3 pin it (probe)
    if (force_order == 3) a.n_major = 0;
  }
  if (a.nbatch > 1) {
    // the split heuristic sees one plane's tiles: a batched launch has nbatch times as many, and its planes are independent problems
    split = 1;
    DFH_REQUIRE(((a.ntaps == 0 && a.nplain == 1) || (a.phase2x && a.nplain == 0)) && a.out_mode == OUT_BF16 && a.act != ACT_GEGLU &&
                !a.resid && !a.rowvec && !a.out2 && !a.rowstat && !a.ln_stat && (a.N & 7) == 0 && (a.ld_out & 7) == 0 && force_split <= 1,
                "batched launch: one plain segment (or the four phase planes of an upsample conv), plain bf16 row-major output, no split-K");
    a.gstat = nullptr; a.ksplit = 1;
    // one plane alone would pick the 256-row tile at the 8x8 level (M <= 1024): the planes together fill the chip with the 128-row tiles
    if (force_tile == 0 && !force_deep) tile = (a.N % 160 == 0) ? kEightWave : 4;
  }
  DFH_REQUIRE(!a.phase2x || (a.nbatch == 4 && a.ntaps == 4 && a.stride == 1 && a.ups == 0 && !a.pad0 && a.Hin == a.Hout && a.Win == a.Wout &&
                             a.M % (a.Hout * a.Wout) == 0), "phase planes of an upsample conv: four planes over the source image");
  if (split > 1) DFH_REQUIRE(a.partial != nullptr, "split-K needs a partial buffer");
  if (a.ln_stat) {
    DFH_REQUIRE(split == 1 && a.ln_parts > 0 && a.ln_cnt > 0 && a.ln_s != nullptr && !a.rowvec && !a.resid,
                "folded LayerNorm: single-pass launches without rowvec / residual only (gemm_ln_consumer_ok)");
    DFH_REQUIRE(a.act != ACT_GEGLU || (a.ld_out & 7) == 0, "folded LayerNorm + GEGLU needs 16-byte aligned output rows");
  }
  if (a.w_img_bs) {
    split = 1; a.ksplit = 1;
    DFH_REQUIRE(a.ntaps == 0 && a.nbatch <= 1 && !a.w_blocked && a.rows_per_b % kTiles[tile].bm == 0 && a.M % a.rows_per_b == 0,
                "per-image weights: plain segments, images of whole row tiles");
  }
  if (a.w_blocked) DFH_REQUIRE(lean_plain(a) && a.N % 16 == 0 && a.ldw % 64 == 0 && a.N % 160 == 0 && force_tile == 0 && split == 1,
                               "blocked W: LEAN launches (plain 64-multiple segments) on the 128 x 160 / 256 x 320 tiles only");
  if (a.out2) DFH_REQUIRE(split == 1 && a.n_split > 0 && a.n_split % kTiles[tile].bn == 0 && a.out_mode == OUT_BF16 && a.act != ACT_GEGLU &&
                          !a.resid && force_tile == 0, "second destination: single pass, n_split a multiple of the column tile (gemm_out2_ok)");
  if (a.resid) DFH_REQUIRE((double)a.M * a.ld_res * 2.0 < 4.0e9, "residual tensor must be smaller than 4 GB (32-bit lane offsets)");
  int rc;
  {
    // algorithmic work of this launch: 2*M*N*K over the REAL K (padding excluded); bytes = each operand once + output
    double kreal = (double)a.ntaps * a.conv_c;
    double abytes = a.ntaps ? (double)(a.M / (a.Hout * a.Wout)) * a.Hin * a.Win * a.conv_c * 2.0 : 0.0;
    for (int i = 0; i < a.nplain; ++i) { kreal += a.p_c[i]; abytes += (double)a.M * a.p_c[i] * 2.0; }
    const double obytes = (double)a.M * (a.act == ACT_GEGLU ? a.N / 2 : a.N) * ((a.out_mode == OUT_F32 || a.out_mode == OUT_F32_T) ? 4.0 : 2.0) +
                          (a.resid ? (double)a.M * a.N * 2.0 : 0.0) +     // the residual is an operand too: read once
                          (a.pre_out ? (double)a.M * a.N * 2.0 : 0.0);
    const double planes = a.nbatch > 1 ? (double)a.nbatch : 1.0;       // a phase launch reads its source image once for all four planes
    // algorithmic multiply-adds = the reference algorithm's: the four phase planes of an upsample conv stand for the
    // 3x3 conv over the upsampled image (9 taps per output pixel, of which the planes execute 4)
    const double flops = a.prof_flops > 0.0 ? a.prof_flops : (a.phase2x ? 9.0 / 4.0 : 1.0) * planes * 2.0 * a.M * a.N * kreal;
    prof_note_saved(flops - planes * 2.0 * a.M * a.N * kreal);
    ProfScope ps((a.ntaps || a.prof_flops > 0.0) ? PC_CONV3 : PC_LINEAR, flops,
                 (a.phase2x == 1 ? abytes : planes * abytes) + planes * ((double)a.N * kreal * 2.0 + obytes), stream);
    const bool wide_ok0 = split == 1 && a.out2 == nullptr && a.out_mode == OUT_BF16 && (a.act != ACT_GEGLU || a.N % 160 == 0 || a.N % 128 == 0) && (a.N & 7) == 0 &&
                         (a.ld_out & 7) == 0 && (!a.resid || (a.ld_res & 7) == 0);
    const bool wide_ok = wide_ok0 && a.nbatch <= 1 && !a.w_img_bs;       // batched launches / per-image weights: gemm_bf16_kernel tiles only (the 256 x 320 one when pinned)
    // tile id 21 pins the 256 x 320 tile (launches it cannot take -- fp32 / transposed outputs, GEGLU, N % 8 -- fall back to the
    // heuristic tile, like the forced wide ids); otherwise gemm_big_pick decides
    const bool force_big = force_wide == 16 ||     // id 21
                           (a.nbatch > 1 && force_tile == 0 && force_wide == 0 && !force_deep && batched_big_pick(a));
    if (force_big) force_wide = 0;
    // tile id 23: the 256 x 256 GEGLU tile
    const bool force_bigg = force_wide == 18;
    if (force_bigg) force_wide = 0;
    // tile id 24: the persistent 128 x 160 kernel (gemm_persist.hip) for every launch it can take; otherwise persist_pick decides
    const bool force_persist = force_wide == 19;
    if (force_persist) { force_wide = 0; tile = kEightWave; }
    const bool pre = a.pre_out != nullptr;      // training GEGLU with its pre-activations as a second output: the 256 x 128 wide tile only
    if (pre) DFH_REQUIRE(wide_ok && a.act == ACT_GEGLU && a.N % 128 == 0 && !a.ln_stat && !a.resid && !a.rowvec && (a.ld_pre & 7) == 0 && a.ld_pre >= a.N,
                         "pre_out: GEGLU launches with N % 128 == 0, no split-K, no folded LayerNorm");
    const bool bigg = !pre && wide_ok && !force_deep && !force_persist && a.act == ACT_GEGLU && a.N % 32 == 0 && !a.resid && !a.rowvec &&
                      (force_bigg || (!force_wide && force_tile == 0 && force_split == 0 && gemm_big_geglu_pick(a)));
    const bool big_ok = wide_ok0 && !force_deep && a.act != ACT_GEGLU && !a.ln_stat && (a.nbatch <= 1 || force_big) && !a.w_img_bs;
    const bool big = big_ok && !force_persist && (force_big || (!force_wide && force_tile == 0 && force_split == 0 && gemm_big_pick(a)));
    int wide = pre ? 5 : (!wide_ok || force_deep || big || force_big || bigg || force_bigg || force_persist) ? 0 : (force_wide ? force_wide : ((force_tile == 0 && force_split == 0) ? gemm_wide_pick(a) : 0));
    int ws = 0; bool halo = false;
#ifdef DFH_PROBES
    // Probe builds only (scripts/probes/Makefile): tile ids 11 / 12 = the wave-specialised kernel (scripts/probes/kernels/gemm_ws.hip, opt-in
    // DFH_GEMM_WS=1), tile id 20 = the halo-patch conv kernel (scripts/probes/kernels/gemm_halo.hip, DFH_GEMM_HALO=1).  Both lost their A/B
    // (profiles/r02/gemm_ws_probe.txt; conv3x3 class 6.96 -> 6.93 ms) and are kept as measurements, not as product code.
    static const bool ws_off = [] { const char* e = getenv("DFH_GEMM_WS"); return !(e && e[0] == '1'); }();
    if (wide_ok && !force_deep) {
      if (force_wide == 6 || force_wide == 7) ws = gemm_ws_pick(a, 1) ? (force_wide == 6 ? 160 : 128) : 0;
      else if (!force_wide && force_tile == 0 && force_split == 0 && !ws_off) ws = gemm_ws_pick(a, 224);
    }
    static const bool halo_on = [] { const char* e = getenv("DFH_GEMM_HALO"); return e && e[0] == '1'; }();
    halo = wide_ok && !force_deep && gemm_halo_eligible(a) && (force_wide == 15 || (wide == 1 && !force_wide && halo_on));
#else
    DFH_REQUIRE(force_wide != 6 && force_wide != 7 && force_wide != 15 && !(force_wide >= 8 && force_wide <= 14),
                "tile ids 11-20 are probe kernels: build scripts/probes (make -C scripts/probes) and load it with DFH_LIB");
#endif
    if (bigg) gemm_pick_tile_order(a, split, 256, 256);
    else if (big) gemm_pick_tile_order(a, split, 256, 320);
    else if (ws) gemm_pick_tile_order(a, split, 256, ws);
    else if (wide) gemm_pick_tile_order(a, split, wide == 2 ? 128 : 256, wide == 3 ? 320 : ((wide == 4 || wide == 5) ? 128 : 160));
    else gemm_pick_tile_order(a, split, kTiles[tile].bm, kTiles[tile].bn);
    if (force_wide == 6 || force_wide == 7) wide = 0;                      // not eligible (odd N, transposed / fp32 output): the default tile runs
    if (force_wide == 15) wide = halo ? 1 : 0;
    // output statistics for the consuming GroupNorm: the 256-row epilogues (256 x 160 wide, 256 x 320) and, since round 5, the staged
    // epilogue of the eight-wave 128 x 160 tile write them, on full tiles inside one image; *gstat_rows = pixel rows per statistics chunk
    const int gbn = big ? 320 : 160;
    const bool gst256 = (halo || big || (wide == 1 && !ws)) && a.M % 256 == 0 && a.gstat_hw % 256 == 0 && a.gstat_hw / 256 <= (int)GN_MAX_CHUNKS;
    static const bool gst128_off = [] { const char* e = getenv("DFH_GSTAT128"); return e && e[0] == '0'; }();      // A/B
    const bool gst128 = !gst128_off && !halo && !big && !bigg && !wide && !ws && tile == kEightWave && split == 1 && a.out_mode == OUT_BF16 && (a.N & 7) == 0 &&
                        (a.ld_out & 7) == 0 && a.M % 128 == 0 && a.gstat_hw % 128 == 0 && a.nbatch <= 1 && !a.phase2x &&
                        a.gstat_hw / 128 <= (int)GN_MAX_CHUNKS &&
                        a.out2 == nullptr;           // the transposed / out2 column tiles return before the statistics block
    const bool gst_ok = a.gstat && (gst256 || gst128) && a.gstat_cpg > 0 && gbn % a.gstat_cpg == 0 && a.N % gbn == 0 &&
                        a.N % a.gstat_cpg == 0 && a.act != ACT_GEGLU;
    if (!gst_ok) a.gstat = nullptr;
    else if (gstat_rows) *gstat_rows = gst256 ? 256 : 128;
    // per-row output statistics for a LayerNorm folded into the consumer: the staged bf16 epilogue of gemm_bf16_kernel and the 256-row
    // epilogue write them, on whole column tiles
    {
      const int bn = big ? 320 : (wide == 1 ? 160 : ((wide == 4 || wide == 5) ? 128 : (wide || ws ? 0 : kTiles[tile].bn)));
      const bool staged_ok = split == 1 && a.out_mode == OUT_BF16 && a.act != ACT_GEGLU && (a.N & 7) == 0 && (a.ld_out & 7) == 0;
      if (!(a.rowstat && bn > 0 && staged_ok && a.N % bn == 0 && !halo)) a.rowstat = nullptr;
      else if (rowstat_bn) *rowstat_bn = bn;
    }
    bool persist = false;
#ifdef DFH_PROBES
    // Probe builds only: the persistent 128 x 160 kernel (scripts/probes/kernels/gemm_persist.hip; tile id 24, or DFH_PERSIST=1 / 2 for every
    // eligible launch with >= 512 / >= 1 tiles).  Bit-identical to the tile kernel and 1.2-1.5 x slower: profiles/r06/persistent_lean_gemm.md.
    static const int persist_mode = [] { const char* e = getenv("DFH_PERSIST"); return e ? atoi(e) : 0; }();
    persist = !bigg && !big && !wide && !halo && !ws && tile == kEightWave && !force_deep && (force_tile == 0 || force_persist) &&
              gemm_persist_ok(a) && (force_persist || (persist_mode != 0 && (long)(a.M / 128) * (a.N / 160) >= (persist_mode >= 2 ? 1 : 512)));
    if (force_persist) DFH_REQUIRE(persist, "tile id 24: this launch cannot run on the persistent kernel (gemm_persist_ok)");
#else
    DFH_REQUIRE(!force_persist, "tile id 24 is a probe kernel: build scripts/probes (make -C scripts/probes) and load it with DFH_LIB");
#endif
#ifdef DFH_PROBES
    if (halo) rc = gemm_halo_launch(a, stream);
    else if (ws) rc = gemm_ws_launch(a, stream, ws);
    else
    if (bigg && force_tile == 0 && !force_bigg && gemm_geglu_rows_ok(a)) rc = gemm_geglu_rows_launch(a, stream);   // opt-in DFH_GEGLU_ROWS=1
    else
#endif
    if (bigg) rc = launch_big_geglu(a, stream);
    else if (big) rc = launch_big(a, stream);
    else if (wide) rc = gemm_wide_launch(a, stream, wide);
#ifdef DFH_PROBES
    else if (persist) rc = gemm_persist_launch(a, stream);
#endif
    else rc = launch_variant(tile, a, stream);
    if (persist) census(CK_GEMM_PERSIST);
    census((big || bigg) ? CK_GEMM_ROW : wide ? CK_GEMM_WIDE : ((halo || ws) ? CK_GEMM_OTHER : (tile == kEightWave ? (lean_plain(a) ? CK_GEMM_LEAN : CK_GEMM_8WAVE) : CK_GEMM_OTHER)));
    if (a.gstat) census(CK_GSTAT_WRITTEN);
    if (a.phase2x) census(CK_CONV_PHASE);
  }
  if (rc) return rc;
  if (split > 1) {
    const long total4 = ((long)a.M * a.N) / 4;
    ProfScope ps(PC_SPLITK, 0.0, (double)split * a.M * a.N * 4.0 + (double)a.M * a.N * 2.0, stream);
    census(CK_SPLITK);
    hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, a);
    return check_launch("gemm_splitk_reduce");
  }
  return 0;
}

}  // namespace dfh
