// 3x3 convolution as an implicit GEMM whose pixel operand is staged ONCE per channel slice (halo patch), not once per tap.
//
// Why: the wide kernel (gemm_wide.hip) is paced by its LDS-DMA staging -- 26 KB per 32-deep k-step, 258 us of staging against
// 185 us of MFMAs on the 64x64-level conv 960 -> 320 (scripts/gemm_ablate_probe.py) -- and 16 of those 26 KB are the SAME
// pixels shifted by one tap.  K is already walked channel-slice-major (the nine taps of a 32-channel slice are consecutive
// k-steps, gemm_kiter.h), so here a slice's pixels are staged once: the 256-pixel tile is R = 256 / W whole image rows, its
// patch the (R + 2) x (W + 2) pixels around them (zero page outside the image), 25 pieces of 16 pixels x 64 B, double
// buffered; the nine taps read their A fragments from the patch at a per-tap offset.  Per k-step that is 2.8 KB of pixels +
// 10 KB of weights instead of 16 + 10.
//   * LDS: 2 patch buffers x 25 KiB + a 3-deep ring of 10-KiB weight tiles = 80 KiB exactly -> two workgroups per CU.
//   * 16-byte slots of a patch pixel are swizzled by (pixel >> 2) & 3: any 16 consecutive pixels x one k-chunk hit 16
//     distinct 16-byte bank groups, whatever the tap shift.
//   * every wave issues the same number of LDS-DMA pieces in every k-step of a slice (2 or 3 weight pieces + one patch piece
//     on taps 0..6; waves with six patch pieces re-issue piece 24), so the in-order vmcnt waits depend on the tap only.
//   * the tile, the MFMA schedule (4 waves x 128 x 80, fragment reads software-pipelined by hand) and the epilogue are the
//     wide kernel's (gemm_wide_epilogue.h).
// Eligible: stride-1 3x3 convs without a fused plain segment, C_in % 32 == 0, W in {16, 32, 64}, H * W % 256 == 0.
#include "gemm.h"
#include "gemm_wide_epilogue.h"

#include <algorithm>

namespace {

constexpr int BKH = 32;
constexpr int HBM = 256, HBN = 160;
constexpr int PATCH_PIECES = 25;                         // 400 patch pixels >= (R + 2) * (W + 2) for R * W = 256, W >= 16
constexpr int PATCH_BYTES = PATCH_PIECES * 1024;
constexpr int HB_BYTES = HBN * BKH * 2;                  // one weight tile
constexpr int NSB = 3;                                   // weight ring depth
constexpr int HALO_LDS = 2 * PATCH_BYTES + NSB * HB_BYTES;
static_assert(HALO_LDS == 80 * 1024, "two workgroups per CU");
constexpr int APW = 7;                                   // patch pieces per wave and slice (taps 0..6 issue one each)

template <int N> DFH_DEVICE void hw_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
DFH_DEVICE void hw_wait(int n) {                         // wave-uniform: a scalar branch per k-step
  switch (n) {
    case 0: hw_vmcnt<0>(); break;
    case 1: hw_vmcnt<1>(); break;
    case 2: hw_vmcnt<2>(); break;
    case 3: hw_vmcnt<3>(); break;
    case 4: hw_vmcnt<4>(); break;
    default: hw_vmcnt<5>(); break;
  }
}
typedef __attribute__((ext_vector_type(4))) unsigned hu32x4_t;
#define HRD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

__global__ __launch_bounds__(256, 2) void gemm_halo_kernel(const GemmArgs a) {
  constexpr int WN = 2, NWV = 4;
  constexpr int TM = HBM / 2, TN = HBN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;              // 8 x 5 fragments per wave
  constexpr int PB = HBN / 16, IB = (PB + NWV - 1) / NWV, PB_REM = PB % NWV;
  static_assert(FM == 8 && FN == 5, "hand-scheduled k-step");

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int ntn = (a.N + HBN - 1) / HBN, ntm = a.M / HBM;
  int mt_, nt_;
  tile_coords(blockIdx.x, ntm, ntn, a.n_major, a.tm_xm, a.tm_gm, mt_, nt_);
  const int m0 = mt_ * HBM, n0 = nt_ * HBN;

  const int W = a.Win, H = a.Hin, HW = H * W, PW = W + 2;
  const int R = HBM / W;
  const int bimg = m0 / HW, y0 = (m0 - bimg * HW) / W;   // the tile = rows y0 .. y0 + R - 1 of image bimg
  const int npix = (R + 2) * PW;
  const unsigned cc = (unsigned)a.conv_c;
  const int S = a.conv_c / BKH, nk = S * 9;

  // ---- patch staging geometry: piece q = wave + 4 i covers patch pixels q*16 .. +15; lane -> (pixel q*16 + lane/4, slot lane&3)
  unsigned p_off[APW];                                   // element offset of the lane's 16 source bytes at channel slice 0
  unsigned p_ok = 0;                                     // bit i: inside the image
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int q = min(wave + NWV * i, PATCH_PIECES - 1);
    const int pidx = q * 16 + (lane >> 2);
    const int py = pidx / PW, px = pidx - py * PW;
    const int y = y0 - 1 + py, x = px - 1;
    const int chunk = (lane & 3) ^ ((pidx >> 2) & 3);
    const bool ok = pidx < npix && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    p_off[i] = ok ? (unsigned)((bimg * H + y) * W + x) * cc + (unsigned)chunk * 8u : 0u;
    p_ok |= ok ? (1u << i) : 0u;
  }
  // ---- weight staging geometry (as gemm_wide.hip: 16 rows x 64 B per piece, slots swizzled by (row >> 1) & 3)
  const int srow = lane >> 2;
  const int schunk = (lane & 3) ^ ((lane >> 3) & 3);
  int w_row[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int n = n0 + (i * NWV + wave) * 16 + srow;
    w_row[i] = (n < a.N) ? n * a.ldw : -1;
  }
  const bool hi_wave = wave < PB_REM;                    // issues IB weight pieces per k-step, the others IB - 1

  auto glds = [&](const bf16_t* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue_patch_piece = [&](int i, int c0, int buf) {
    const int q = min(wave + NWV * i, PATCH_PIECES - 1);
    glds(((p_ok >> i) & 1u) ? a.conv_src + (p_off[i] + (unsigned)c0) : a.zero, smem + buf * PATCH_BYTES + q * 1024);
  };
  auto issue_w = [&](int kstep, int ring) {              // weight tile of k-step (slice, tap) = (kstep / 9, kstep % 9)
    const int s = kstep / 9, tap = kstep - s * 9;
    const unsigned wc = (unsigned)(tap * a.conv_c + s * BKH + schunk * 8);
    unsigned char* Bs = smem + 2 * PATCH_BYTES + ring * HB_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      if (i * NWV + wave >= PB) continue;                // wave-uniform
      glds(w_row[i] >= 0 ? a.W + ((unsigned)w_row[i] + wc) : a.zero, Bs + i * NWV * 1024);
    }
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  // A fragment i of this wave = tile pixels wm*128 + i*16 .. +15 (one image row: 16 | W | 128, so fragment 0 starts in column 0 and
  // fragment i is i*16 pixels further plus two pad pixels per image row crossed: a wave-uniform offset).  pb0 = byte offset of
  // lane fr's pixel of fragment 0 in the patch at tap (0, 0); tap (ky, kx) adds (ky * PW + kx) * 64.
  const int lw = W == 64 ? 2 : (W == 32 ? 1 : 0);        // log2(fragments per image row)
  const unsigned pb0 = (unsigned)(((wm * TM) / W) * PW + fr) * 64u;
  auto frag_off = [&](int i) -> unsigned { return (unsigned)(i * 1024 + ((i >> lw) << 7)); };
  const unsigned fslot = (unsigned)((fg ^ ((fr >> 1) & 3)) << 4);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned b_off = lds0 + 2 * PATCH_BYTES + (unsigned)(wn * TN + fr) * 64u + fslot;
  const unsigned fgv = (unsigned)fg;
  unsigned pbx = pb0;                                    // made opaque once per k-step (below)
  auto a_addr = [&](int i, unsigned tbase) -> unsigned {           // tbase = LDS base of the patch buffer + tap offset
    const unsigned t1 = pbx + (tbase + frag_off(i));
    return t1 + ((fgv ^ ((t1 >> 8) & 3u)) << 4);
  };

  // ---- prologue: patch of slice 0, weight tiles of k-steps 0 and 1
#pragma unroll
  for (int i = 0; i < APW; ++i) issue_patch_piece(i, 0, 0);
  issue_w(0, 0);
  issue_w(min(1, nk - 1), 1);

  for (int s = 0; s < S; ++s) {
    const int c_next = min(s + 1, S - 1) * BKH;          // the last slice re-stages itself: keeps the issue counts uniform
    const int cur = s & 1;
#pragma unroll 1                                          // unrolled, hipcc needs 280 VGPRs for the nine bodies and spills
    for (int tap = 0; tap < 9; ++tap) {
      // in-order vmcnt: behind the weight tile of this k-step were issued the patch piece of step t-2, the weight tile of
      // step t+1 and the patch piece of step t-1
      const int extra = ((tap >= 1 && tap <= 7) ? 1 : 0) + ((tap >= 2 && tap <= 8) ? 1 : 0);
      if (hi_wave) hw_wait(IB + extra); else hw_wait(IB - 1 + extra);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_w(min(s * 9 + tap + 2, nk - 1), (tap + 2) % 3);
      if (tap < APW) issue_patch_piece(tap, c_next, cur ^ 1);

      const int ky = tap / 3, kx = tap - ky * 3;
      const unsigned tbase = lds0 + (unsigned)(cur * PATCH_BYTES) + (unsigned)((ky * PW + kx) * 64);
      const unsigned sb = b_off + (unsigned)((tap % 3) * HB_BYTES);
      hu32x4_t b[FN], a0, a1, a2;
      // the fragment addresses are four VALU ops each; left visible, hipcc hoists all 72 of a slice out of the loop and spills
      pbx = pb0;
      asm volatile("" : "+v"(pbx));
      __builtin_amdgcn_sched_barrier(0);
      HRD(b[0], sb, 0); HRD(b[1], sb, 1024); HRD(b[2], sb, 2048); HRD(b[3], sb, 3072); HRD(b[4], sb, 4096);
      { const unsigned ad = a_addr(0, tbase); HRD(a0, ad, 0); }
      { const unsigned ad = a_addr(1, tbase); HRD(a1, ad, 0); }
      asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(a0));
#define HROW(i, ar)                                                                                                   \
      _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[j]), __builtin_bit_cast(bf16x8_t, ar), \
                                                            acc[i][j], 0, 0, 0);
#define HNEXT(rd, i, wt) { const unsigned ad = a_addr(i, tbase); HRD(rd, ad, 0); } asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wt));
      HROW(0, a0) HNEXT(a2, 2, a1)
      HROW(1, a1) HNEXT(a0, 3, a2)
      HROW(2, a2) HNEXT(a1, 4, a0)
      HROW(3, a0) HNEXT(a2, 5, a1)
      HROW(4, a1) HNEXT(a0, 6, a2)
      HROW(5, a2) HNEXT(a1, 7, a0)
      HROW(6, a0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a1));
      HROW(7, a1)
#undef HROW
#undef HNEXT
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the re-issued pieces of the last k-steps must have landed before the epilogue reuses the buffers
  hw_vmcnt<0>();
  asm volatile("" ::: "memory");

  wide_epilogue<HBM, HBN, WN, HALO_LDS>(a, acc, smem, tid, wm, wn, fr, fg, m0, n0);
}

}  // namespace

namespace dfh {

bool gemm_halo_eligible(const GemmArgs& a) {
  if (a.ntaps != 9 || a.nplain != 0 || a.stride != 1 || a.ups != 0 || a.pad0) return false;
  if (a.conv_c % BKH != 0 || a.Hin != a.Hout || a.Win != a.Wout) return false;
  if (a.Win != 16 && a.Win != 32 && a.Win != 64) return false;
  if ((a.Hin * a.Win) % HBM != 0 || a.M % HBM != 0) return false;
  if (a.out_mode != OUT_BF16 || a.act == ACT_GEGLU) return false;
  if ((a.N & 7) || (a.ld_out & 7) || (a.resid && (a.ld_res & 7))) return false;
  if ((double)a.M / (a.Hout * a.Wout) * a.Hin * a.Win * a.conv_c * 2.0 >= 4.0e9) return false;      // 32-bit element offsets
  return true;
}

int gemm_halo_launch(GemmArgs a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
    attr_set = true;
  }
  a.ksteps = 9 * (a.conv_c / BKH);
  a.ksplit = 1;
  const int tiles = (a.M / HBM) * ((a.N + HBN - 1) / HBN);
  hipLaunchKernelGGL(gemm_halo_kernel, dim3(tiles), dim3(256), HALO_LDS, s, a);
  return check_launch("gemm_halo_kernel");
}

}  // namespace dfh
