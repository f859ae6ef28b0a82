// PERSISTENT short-K token linear: the 128 x 160 eight-wave LEAN tile of gemm.hip with its LDS ring running ACROSS tiles.
//
// Serves: the K = C projections of diffusers' Transformer2DModel / BasicTransformerBlock at the 64x64 and 32x32 levels (proj_in, q | k | V^T,
// attn1 / attn2 to_out, attn2.to_q, ff.net.2 . proj_out; reference call site DiFashion/models/difashion.py:518-523).  Those launches are five
// or ten k-steps per tile: in gemm_bf16_kernel<128,160,4,2,2,LEAN> every tile pays its own pipeline fill (the first stage's ~2 us round trip with
// nothing in flight behind it) and drains the ring into its epilogue -- rocprofv3 PMC (profiles/r05/pmc_short_k_linears.txt): 47 % of the wave
// cycles in s_waitcnt, MFMA busy 25 %, 2.5-3 TB/s.  Here ONE workgroup per CU walks its tiles (grid = min(tiles, 256), tile v -> v + grid:
// the same tile -> XCD map as the one-tile-per-workgroup launch), and the k-step sequence of ALL its tiles is one stream through a three-stage
// ring: two stages (72 KB) are in flight at every moment, including under every epilogue; the next tile's first two k-steps land while the
// current tile is staged and stored.
//
// The k-loop is software-pipelined (second form, see profiles/r06/persistent_lean_gemm.md): fragment reads of the next 32-deep half in front of the
// MFMAs of the current one, the workgroup's barrier in the middle of a k-step, stage g + 3 refilled behind it; the tile order advances without
// divisions.  MEASURED 1.1-1.5 x SLOWER than the tile kernel at two workgroups per CU: this file lives in the probe library only.
//
// What makes the ring survive the epilogue:
//   * the staged output tile has its own LDS region (43.5 KB behind the ring); the ring is never the epilogue's scratch;
//   * EVERYTHING the epilogue needs from global memory arrives by LDS-DMA issued at the tile's first k-step: the residual tile (42 pieces,
//     into the staging region in the staged layout: the epilogue adds it in place), the bias and row-vector / folded-LayerNorm column
//     slices and the LayerNorm row records (one 1-KB piece each).  No VGPR-returning load exists in the loop, so the compiler never
//     inserts an s_waitcnt vmcnt(0) that would drain the ring;
//   * every wait is a COUNTED s_waitcnt: vector-memory operations retire in order, so "stage g has landed" == "at most as many operations
//     are outstanding as this wave issued after stage g's pieces".  The kernel keeps that number in an SGPR (a running count of issued
//     operations and a mark per ring slot); the epilogue's sixteen-byte stores (exactly five per thread: whole tiles only) are counted like
//     the pieces.  The few conditional statistics stores are NOT counted: an under-count only makes a wait stricter, never unsafe.
// Results are bit-identical to gemm_bf16_kernel's staged epilogue (same MFMA order per tile, same epilogue arithmetic, same fixed-order
// statistics): tests/test_gpu_ops.py::test_gemm_persistent_matches_the_tile_kernel_bit_for_bit.
#include "gemm.h"
#include "gemm_kiter.h"
#include <cstdlib>

namespace {

constexpr int PBM = 128, PBN = 160, PWM = 4, PWN = 2, PNWV = 8, PNST = 3;
constexpr int PTM = PBM / PWM, PTN = PBN / PWN;        // 32 x 80 per wave
constexpr int PFM = PTM / 16, PFN = PTN / 16;          // 2 x 5 fragments
constexpr int P_A_BYTES = PBM * BK * 2, P_B_BYTES = PBN * BK * 2, P_STAGE = P_A_BYTES + P_B_BYTES;      // 16 KB + 20 KB
constexpr int P_PA = PBM / 8, P_PB = PBN / 8;          // 16 + 20 one-KB pieces per stage
constexpr int P_IA = P_PA / PNWV, P_IB = (P_PB + PNWV - 1) / PNWV, P_PB_REM = P_PB % PNWV;   // 2, 3, 4
constexpr int P_RS = PBN * 2 + 16;                     // staged output row (bytes): 336
constexpr int P_RST = PBM * 2 + 16;                    // transposed staged row: 272
constexpr int P_TILE_B = PBN * P_RST > PBM * P_RS ? PBN * P_RST : PBM * P_RS;     // 43,520
constexpr int P_EPI = PNST * P_STAGE;                  // 110,592
constexpr int P_AUX = P_EPI + P_TILE_B;                // 154,112: six 1-KB slots -- bias | strip2 | up to four LayerNorm record parts
constexpr int P_AUX_SLOTS = 6, P_MAX_LN_PARTS = 4;
constexpr int P_LDS = P_AUX + P_AUX_SLOTS * 1024;      // 160,256 of 163,840
constexpr int P_RES_PIECES = PBM * (P_RS / 16) / 64;   // 42 (21 chunks per staged row incl. the pad chunk)
constexpr int P_STORES = PBM * (PBN / 8) / (PNWV * 64);   // 5 sixteen-byte stores per thread
static_assert(PBM * (P_RS / 16) % 64 == 0 && PBM * (PBN / 8) % (PNWV * 64) == 0 && PBN * (PBM / 8) == PBM * (PBN / 8), "whole pieces / store rounds");
static_assert(P_LDS <= 160 * 1024, "LDS");

// s_waitcnt vmcnt(n) for a wave-uniform RUNTIME n (clamped down to 31: waiting for fewer outstanding operations than allowed is always safe)
DFH_DEVICE void wait_vmcnt_dyn(int n) {
#define DFH_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n < 31 ? n : 31) {
    DFH_W(0) DFH_W(1) DFH_W(2) DFH_W(3) DFH_W(4) DFH_W(5) DFH_W(6) DFH_W(7) DFH_W(8) DFH_W(9) DFH_W(10) DFH_W(11) DFH_W(12) DFH_W(13)
    DFH_W(14) DFH_W(15) DFH_W(16) DFH_W(17) DFH_W(18) DFH_W(19) DFH_W(20) DFH_W(21) DFH_W(22) DFH_W(23) DFH_W(24) DFH_W(25) DFH_W(26)
    DFH_W(27) DFH_W(28) DFH_W(29) DFH_W(30)
    default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
  }
#undef DFH_W
}

// workgroup barrier that orders LDS traffic ONLY: __syncthreads() carries a workgroup-scope fence, i.e. s_waitcnt vmcnt(0) -- it would drain
// the two ring stages in flight under every epilogue.  LDS writes are complete when lgkmcnt reaches zero.
DFH_DEVICE void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);                    // vmcnt(63) expcnt(7) lgkmcnt(0)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// 64 lanes x 16 bytes, global -> LDS, no VGPR round trip: lane l's chunk lands at lds_addr + 16 l; source = block-uniform base + per-lane 32-bit
// byte offset.  INLINE ASM on purpose: for the builtin the compiler tracks the LDS write and puts an s_waitcnt vmcnt(0) in front of every
// LDS access that might alias it -- which is every access of the epilogue (it cannot tell the staging region from the ring) -- draining
// the two stages this kernel exists to keep in flight.  Behind the asm it sees no load at all; every wait in this file is explicit.
DFH_DEVICE void glds(const char* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}

// a block-uniform pointer, pinned in SGPRs: the tile coordinates come out of integer divisions the compiler runs on the VALU, so pointers
// derived from them sit in VGPRs although every lane holds the same value; v_readfirstlane moves them where a scalar base belongs
DFH_DEVICE const char* uni_ptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(PNWV * 64, 2) void gemm_persist_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / PWN, wn = wave % PWN;
  const int fr = lane & 15, fg = lane >> 4;
  const int ntn = a.N / PBN, ntm = a.M / PBM, ntiles = ntm * ntn;
  const int nk = a.ksteps;
  const int P = gridDim.x;
  const int my_tiles = (ntiles - (int)blockIdx.x + P - 1) / P;
  const int G = my_tiles * nk;                          // this workgroup's k-steps, all tiles

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;      // LDS byte address of the dynamic segment
  const int srow = lane >> 3;
  const int sslot = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));      // == slot ^ ((row >> 1) & 7): the swizzle of gemm.hip
  const bool hi_wave = wave < P_PB_REM;
  const int n_stage = (P_IA + P_PB / PNWV) + (hi_wave ? 1 : 0);          // pieces this wave issues per stage: 4 or 5

  // ---- counted-wait bookkeeping (wave-uniform scalars)
  int issued = 0;                      // counted vector-memory operations this wave has issued
  int mark0 = 0, mark1 = 0, mark2 = 0; // `issued` right after the pieces of the stage in ring slot 0 / 1 / 2 (scalars, not an indexed array: no scratch)

  // ---- tile order without divisions in the loop.  Workgroup b owns the tile ids v = b + j P (P a multiple of 8: every one of them on XCD b & 7,
  //      dfh_common.h xcd_remap); in the legacy order (tm_gm == 0, required by gemm_persist_ok) the logical tile of id v is base(xcd) + (v >> 3), so
  //      consecutive tiles of a workgroup are P / 8 apart: (hi, lo) = divmod(tile, D) advances by divmod(P / 8, D) with one carry.  The divisions the
  //      compiler would run on the VALU (~40 instructions each, 8 waves) cost more than a k-step when done per tile.
  const int Dv = a.n_major ? ntm : ntn;                 // m-major: tile = mt * ntn + nt;  n-major: tile = nt * ntm + mt
  int q_hi, q_lo;                                       // divmod(logical tile, Dv) of the NEXT tile the producer will start
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int tile0 = (P & 7) ? (int)blockIdx.x : ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    q_hi = tile0 / Dv; q_lo = tile0 - q_hi * Dv;
  }
  const int st = (P & 7) ? P : (P >> 3);                // logical-tile distance between a workgroup's consecutive tiles
  const int st_hi = st / Dv, st_lo = st - st_hi * Dv;
  int q_m0 = 0, q_n0 = 0;                               // coordinates of the tile the producer started last: read by the consumer two k-steps later
  // residual pieces: staged-layout chunk c = p * 64 + lane -> (row, 16-byte chunk) once per kernel
  unsigned r_row[(P_RES_PIECES + PNWV - 1) / PNWV], r_q8[(P_RES_PIECES + PNWV - 1) / PNWV];
#pragma unroll
  for (int i = 0; i < (P_RES_PIECES + PNWV - 1) / PNWV; ++i) {
    const int c = (i * PNWV + wave) * 64 + lane, row = c / 21, qq = c - row * 21;
    r_row[i] = (unsigned)row; r_q8[i] = (unsigned)((qq < 20 ? qq : 0) * 8);
  }
  const bool need_tb = a.w_img_bs != 0 || a.rowvec != nullptr || a.out2 != nullptr || a.out_mode == OUT_BF16_T;

  // ---- producer side: the stream of (tile, k-step) pairs, up to three k-steps ahead of the consumer
  int is_t = 0, is_slot = 0;                            // next stage to issue: k-step inside its tile, ring slot
  int n_issued = 0;                                     // stages issued so far (all tiles)
  unsigned lo_a[P_IA], lo_w[P_IB];
  const char* is_abase = (const char*)a.p_src[0];
  const char* is_wb = (const char*)a.W;
  int is_left = 0x7fffffff, is_m0 = 0;
  const int steps0 = a.p_c[0] / BK;
  const unsigned pc0 = (unsigned)a.p_c[0], pc1 = (unsigned)a.p_c[1], ldw_u = (unsigned)a.ldw;

  f32x4_t acc[PFM][PFN];
#pragma unroll
  for (int i = 0; i < PFM; ++i)
#pragma unroll
    for (int j = 0; j < PFN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const bool lnf = a.ln_stat != nullptr;
  const bool have_strip2 = a.rowvec != nullptr || lnf;
  unsigned char* const epi = smem + P_EPI;
  unsigned char* const aux = smem + P_AUX;

  // One stage = this wave's pieces of the A and W tiles of the next (tile, k-step) of the stream
#define DFH_ISSUE_STAGE()                                                                                                                    \
  do {                                                                                                                                        \
    if (is_t == 0) {                                                                                                                          \
      const int mt = a.n_major ? q_lo : q_hi, nt = a.n_major ? q_hi : q_lo;                                                                   \
      q_lo += st_lo; q_hi += st_hi;                                                                                                           \
      if (q_lo >= Dv) { q_lo -= Dv; ++q_hi; }                                                                                                 \
      is_m0 = mt * PBM; q_m0 = is_m0; q_n0 = nt * PBN;                                                                                        \
      is_abase = (const char*)a.p_src[0];                                                                                                     \
      is_left = a.nplain == 2 ? steps0 : 0x7fffffff;                                                                                          \
      is_wb = (const char*)(a.W + (a.w_img_bs ? (long)(is_m0 / a.rows_per_b) * a.w_img_bs : 0L));                                              \
      _Pragma("unroll") for (int i = 0; i < P_IA; ++i) lo_a[i] = ((unsigned)(is_m0 + (i * PNWV + wave) * 8 + srow) * pc0 + (unsigned)sslot * 8) * 2u; \
      _Pragma("unroll") for (int i = 0; i < P_IB; ++i)                                                                                        \
        lo_w[i] = ((unsigned)min(q_n0 + (i * PNWV + wave) * 8 + srow, a.N - 1) * ldw_u + (unsigned)sslot * 8) * 2u;                            \
    }                                                                                                                                         \
    if (is_left == 0) {                                                                                                                       \
      is_abase = (const char*)a.p_src[1];                                                                                                     \
      _Pragma("unroll") for (int i = 0; i < P_IA; ++i) lo_a[i] = ((unsigned)(is_m0 + (i * PNWV + wave) * 8 + srow) * pc1 + (unsigned)sslot * 8) * 2u; \
      is_left = 0x7fffffff;                                                                                                                   \
    }                                                                                                                                         \
    --is_left;                                                                                                                                \
    const unsigned As_ = lds0 + is_slot * P_STAGE + wave * 1024, Bs_ = As_ + P_A_BYTES;                                                       \
    const char* ab = uni_ptr(is_abase);                                                                                                       \
    const char* wb = uni_ptr(is_wb);                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < P_IA; ++i) { glds(ab, lo_a[i], As_ + i * PNWV * 1024); lo_a[i] += BK * 2; }                          \
    _Pragma("unroll") for (int i = 0; i < P_IB; ++i) {                                                                                        \
      if (i * PNWV + wave >= P_PB) continue;                                                                                                  \
      glds(wb, lo_w[i], Bs_ + i * PNWV * 1024); lo_w[i] += BK * 2;                                                                            \
    }                                                                                                                                         \
    issued += n_stage;                                                                                                                        \
    if (is_slot == 0) mark0 = issued; else if (is_slot == 1) mark1 = issued; else mark2 = issued;                                             \
    if (++is_slot == PNST) is_slot = 0;                                                                                                       \
    if (++is_t == nk) is_t = 0;                                                                                                               \
    ++n_issued;                                                                                                                               \
  } while (0)

  // fragments of one 32-deep half of a k-step: ds_read_b128 with the swizzle of gemm.hip
#define DFH_LOAD_FRAGS(AF, BF, SLOT, KS)                                                                                                      \
  do {                                                                                                                                        \
    const unsigned char* As_ = smem + (SLOT) * P_STAGE;                                                                                       \
    const unsigned char* Bs_ = As_ + P_A_BYTES;                                                                                               \
    _Pragma("unroll") for (int i = 0; i < PFM; ++i) {                                                                                         \
      const int row = wm * PTM + i * 16 + fr;                                                                                                 \
      AF[i] = *(const bf16x8_t*)(As_ + row * 128 + ((((KS) * 4 + fg) ^ ((row >> 1) & 7)) << 4));                                               \
    }                                                                                                                                         \
    _Pragma("unroll") for (int jj = 0; jj < PFN; ++jj) {                                                                                      \
      const int row = wn * PTN + jj * 16 + fr;                                                                                                \
      BF[jj] = *(const bf16x8_t*)(Bs_ + row * 128 + ((((KS) * 4 + fg) ^ ((row >> 1) & 7)) << 4));                                              \
    }                                                                                                                                         \
  } while (0)
#define DFH_MFMAS(AF, BF)                                                                                                                      \
  _Pragma("unroll") for (int i = 0; i < PFM; ++i)                                                                                             \
    _Pragma("unroll") for (int jj = 0; jj < PFN; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[jj], AF[i], acc[i][jj], 0, 0, 0)

  // ---- consumer side: k-step t of the current tile, ring slot buf; its tile coordinates
  int t = 0, buf = 0;
  int m0 = 0, n0 = 0, nt_ = 0, tb_ = 0;
  bool part2 = false, tr = false;
  bf16x8_t afA[PFM], bfA[PFN], afB[PFM], bfB[PFN];      // fragment sets of the two halves of a k-step: one is read while the other feeds the MFMAs

  // pipeline fill, once per launch: all three ring slots, then the first half-step's fragments
  if (G > 0) DFH_ISSUE_STAGE();
  if (G > 1) DFH_ISSUE_STAGE();
  if (G > 2) DFH_ISSUE_STAGE();
  if (G > 0) {
    wait_vmcnt_dyn(issued - mark0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DFH_LOAD_FRAGS(afA, bfA, 0, 0);
  }
  // SOFTWARE-PIPELINED k-loop: the fragment reads of the NEXT 32-deep half are issued in front of the MFMAs of the current one, and the
  // workgroup meets in the MIDDLE of a k-step (when its second half sits in registers, i.e. when nobody reads the stage any more).  One
  // workgroup per CU moves through its phases in lock step -- without this, "all waves read LDS" (875 cycles per k-step of this tile) and "all
  // waves issue MFMAs" (640) add up instead of overlapping (profiles/r06/persistent_lean_gemm.md).
  for (int g = 0; g < G; ++g) {
    const int nbuf = buf == PNST - 1 ? 0 : buf + 1;
    DFH_LOAD_FRAGS(afB, bfB, buf, 1);                    // second half of stage g -> registers ...
    __builtin_amdgcn_sched_barrier(0);
    DFH_MFMAS(afA, bfA);                                 // ... under the MFMAs of its first half
    __builtin_amdgcn_sched_barrier(0);
    // lgkmcnt(0) through the BUILTIN: the compiler's own wait insertion sees it and knows the second-half fragments are in registers; behind an
    // asm wait it re-waits in front of their MFMAs -- by then the next half's reads are in flight, and it waits for THOSE (in-order counter)
    __builtin_amdgcn_s_waitcnt(0xC07F);                  // vmcnt(63) expcnt(7) lgkmcnt(0): this wave no longer reads stage g
    asm volatile("" ::: "memory");
    if (g + 1 < G) wait_vmcnt_dyn(issued - (nbuf == 0 ? mark0 : (nbuf == 1 ? mark1 : mark2)));      // stage g + 1 landed (this wave's pieces)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // stage g + 1 visible to every wave; every wave is done with stage g (and with the last epilogue's LDS reads)
    asm volatile("" ::: "memory");
    if (n_issued < G) DFH_ISSUE_STAGE();                 // refill the slot of stage g with stage g + 3
    if (t == 0) {
      // tile start.  The producer started this tile's first stage three k-steps ago and will not start the next tile before k-step nk - 3 of
      // this one (nk >= 4): q_m0 / q_n0 still hold what it saved then.
      m0 = q_m0; n0 = q_n0;
      nt_ = n0 / PBN;
      tb_ = need_tb ? m0 / a.rows_per_b : 0;
      part2 = a.out2 != nullptr && n0 >= a.n_split;      // block-uniform: a transposed tile of the second destination
      tr = part2 || a.out_mode == OUT_BF16_T;
      // the epilogue's operands, by LDS-DMA: residual tile -> staging region (staged layout, in place), column slices and LayerNorm records
      if (a.resid && !tr) {
        const char* rb = uni_ptr(a.resid);
#pragma unroll
        for (int i = 0; i < (P_RES_PIECES + PNWV - 1) / PNWV; ++i) {
          const int p = i * PNWV + wave;
          if (p >= P_RES_PIECES) continue;               // wave-uniform
          const unsigned off = (((unsigned)m0 + r_row[i]) * (unsigned)a.ld_res + (unsigned)n0 + r_q8[i]) * 2u;
          glds(rb, off, lds0 + P_EPI + p * 1024);
        }
        issued += (P_RES_PIECES / PNWV) + (wave < P_RES_PIECES % PNWV ? 1 : 0);
      }
      // aux slot w is issued by wave w: 0 bias, 1 row vector / folded-LayerNorm s slice, 2.. LayerNorm records of the tile's rows
      const int cl = lane < 40 ? lane : 39;              // 160 floats = 40 chunks; the upper lanes repeat the last one inside the slot
      if (wave == 0 && a.bias) { glds(uni_ptr(a.bias + n0), cl * 16, lds0 + P_AUX); issued += 1; }
      if (wave == 1 && have_strip2) {
        const float* s2 = lnf ? a.ln_s + n0 : a.rowvec + (long)tb_ * a.rv_ld + a.rv_off + n0;
        glds(uni_ptr(s2), cl * 16, lds0 + P_AUX + 1024); issued += 1;
      }
      if (lnf && wave >= 2 && wave - 2 < a.ln_parts) {
        glds(uni_ptr(a.ln_stat + ((long)(wave - 2) * a.M + m0) * 2), lane * 16, lds0 + P_AUX + wave * 1024); issued += 1;
      }
    }
    if (g + 1 < G) DFH_LOAD_FRAGS(afA, bfA, nbuf, 0);    // first half of stage g + 1 -> registers ...
    __builtin_amdgcn_sched_barrier(0);
    DFH_MFMAS(afB, bfB);                                 // ... under the MFMAs of the second half of stage g
    __builtin_amdgcn_sched_barrier(0);
    buf = nbuf;
    if (++t < nk) continue;
    t = 0;

    // ---------------------------------------------------------------- epilogue of tile j (gemm.hip's staged branch on its own LDS)
    // Its DMA operands were issued at this tile's first k-step, BEFORE the stage of k-step 2: the counted waits of k-steps >= 3 retired
    // them (in-order completion; the launcher requires nk >= 4).  The barrier makes every wave's pieces visible to every wave.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* bias_s = (const float*)aux;
    const float* strip2_s = (const float*)(aux + 1024);
#pragma unroll
    for (int i = 0; i < PFM; ++i) {
      const int row = wm * PTM + i * 16 + fr;
      float mean = 0.f, rstd = 1.f;
      if (lnf) {                                          // gemm.h ln_row_stats on the records in LDS
        float sm = 0.f, sq = 0.f, s2 = 0.f;
        for (int p = 0; p < a.ln_parts; ++p) {
          const float2 r = *(const float2*)(aux + (2 + p) * 1024 + row * 8);
          sm += r.x; sq = fmaf(r.x, r.x, sq); s2 += r.y;
        }
        const float inv = 1.0f / (float)a.ln_parts;
        mean = sm * inv;
        const float m2 = s2 + (float)a.ln_cnt * fmaxf(sq - sm * mean, 0.f);
        rstd = rsqrtf(m2 * inv / (float)a.ln_cnt + a.ln_eps);
      }
#pragma unroll
      for (int jj = 0; jj < PFN; ++jj) {
        const int col = wn * PTN + jj * 16 + fg * 4;
        float4 bv = float4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) bv = *(const float4*)(bias_s + col);
        float v[4] = {acc[i][jj][0] + bv.x, acc[i][jj][1] + bv.y, acc[i][jj][2] + bv.z, acc[i][jj][3] + bv.w};
        if (lnf) {                                       // rstd * (acc - mean * s) + b'
          const float4 sv = *(const float4*)(strip2_s + col);
          const float ms = -mean * rstd;
          v[0] = fmaf(rstd, acc[i][jj][0], fmaf(ms, sv.x, bv.x)); v[1] = fmaf(rstd, acc[i][jj][1], fmaf(ms, sv.y, bv.y));
          v[2] = fmaf(rstd, acc[i][jj][2], fmaf(ms, sv.z, bv.z)); v[3] = fmaf(rstd, acc[i][jj][3], fmaf(ms, sv.w, bv.w));
        }
        if (a.rowvec) {
          const float4 rv = *(const float4*)(strip2_s + col);
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
        if (tr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) *(bf16_t*)(epi + (col + r) * P_RST + row * 2) = f2bf(v[r]);
        } else {
          if (a.resid) {
            const uint2 rr = *(const uint2*)(epi + row * P_RS + col * 2);
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
          }
          uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
          *(uint2*)(epi + row * P_RS + col * 2) = o;
        }
        acc[i][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    lds_barrier();
    if (tr) {
      bf16_t* const tbase = part2 ? (bf16_t*)a.out2 : (bf16_t*)a.out;
      const int tld = part2 ? a.ld_out2 : a.ld_out;
      const int tN = part2 ? a.N - a.n_split : a.N, tn0 = part2 ? a.n_split : 0;
      constexpr int CPT = PBM / 8;
      const int mm0 = m0 - tb_ * a.rows_per_b;
#pragma unroll
      for (int it = 0; it < P_STORES; ++it) {
        const int c = tid + it * PNWV * 64;
        const int col = c / CPT, cc = c - col * CPT;
        *(uint4*)(tbase + ((long)tb_ * tN + (n0 + col - tn0)) * tld + mm0 + cc * 8) = *(const uint4*)(epi + col * P_RST + cc * 16);
      }
      issued += P_STORES;
      continue;                                          // (the launcher sends no statistics request with a transposed output)
    }
    constexpr int CPR = PBN / 8;
    if (!(a.pad0 & 1))
#pragma unroll
    for (int it = 0; it < P_STORES; ++it) {
      const int c = tid + it * PNWV * 64;
      const int row = c / CPR, cc = c - row * CPR;
      *(uint4*)((bf16_t*)a.out + (long)(m0 + row) * a.ld_out + n0 + cc * 8) = *(const uint4*)(epi + row * P_RS + cc * 16);
    }
    issued += P_STORES;
    if (a.rowstat) {                                     // gemm.hip: row statistics of the bf16-rounded outputs for a LayerNorm folded into the consumer
      constexpr int TPR = PNWV * 64 / PBM;               // 4
      const int row = tid / TPR, part = tid % TPR;
      const unsigned char* src = epi + row * P_RS;
      float sum = 0.f;
      for (int c = part; c < CPR; c += TPR) {
        float f[8]; unpack8(*(const uint4*)(src + c * 16), f);
#pragma unroll
        for (int r = 0; r < 8; ++r) sum += f[r];
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      const float mean = sum * (1.0f / PBN);
      float m2 = 0.f;
      for (int c = part; c < CPR; c += TPR) {
        float f[8]; unpack8(*(const uint4*)(src + c * 16), f);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const float d = f[r] - mean; m2 = fmaf(d, d, m2); }
      }
      m2 += __shfl_xor(m2, 1, 64);
      m2 += __shfl_xor(m2, 2, 64);
      if (part == 0) *(float2*)(a.rowstat + ((long)nt_ * a.M + m0 + row) * 2) = float2{mean, m2};
    }
    if (a.gstat) {                                       // gemm.hip: GroupNorm statistics of the tile for its consumer (128-row chunks, fixed orders)
      float* qrt = (float*)aux;                          // [2 row halves][BN][2] + [BN][2] = 3,840 B over the aux slots (consumed above)
      float* cst = qrt + 2 * PBN * 2;
      lds_barrier();                                   // every wave is done with the aux slots and the row-statistics reads
      if (tid < 2 * PBN) {
        const int rq = tid / PBN, col = tid - rq * PBN;
        const unsigned char* src = epi + (rq * (PBM / 2)) * P_RS + col * 2;
        float ss = 0.f, qq = 0.f;
        for (int r = 0; r < PBM / 2; ++r) {
          const float v = bf2f(*(const bf16_t*)(src + r * P_RS));
          ss += v; qq = fmaf(v, v, qq);
        }
        qrt[(rq * PBN + col) * 2] = ss; qrt[(rq * PBN + col) * 2 + 1] = qq;
      }
      lds_barrier();
      if (tid < PBN) {
        cst[tid * 2] = qrt[tid * 2] + qrt[(PBN + tid) * 2];
        cst[tid * 2 + 1] = qrt[tid * 2 + 1] + qrt[(PBN + tid) * 2 + 1];
      }
      lds_barrier();
      const int cpg = a.gstat_cpg;
      if (tid < PBN / cpg) {
        float ss = 0.f, qq = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { ss += cst[c * 2]; qq += cst[c * 2 + 1]; }
        const int b = m0 / a.gstat_hw, chunk = (m0 - b * a.gstat_hw) / PBM, chunks = a.gstat_hw / PBM;
        const int gi = (n0 + tid * cpg) / cpg, Gn = a.N / cpg;
        float* dst = a.gstat + (((long)b * Gn + gi) * chunks + chunk) * 2;
        dst[0] = ss; dst[1] = qq;
      }
    }
  }
}

}  // namespace

namespace dfh {

// Can this launch run on the persistent kernel?  (a is the launcher's view: ksteps / rows_per_b / tile order filled in.)
bool gemm_persist_ok(const GemmArgs& a) {
  if (a.ntaps != 0 || a.nplain < 1 || a.nplain > 2 || a.nbatch > 1 || a.ksplit > 1 || a.w_blocked || a.phase2x || a.pre_out) return false;
  for (int i = 0; i < a.nplain; ++i) if (a.p_c[i] <= 0 || a.p_c[i] % BK != 0) return false;
  if (a.M % PBM != 0 || a.N % PBN != 0 || a.ksteps < 4) return false;
  if (a.act == ACT_GEGLU) return false;
  const bool plain = a.out_mode == OUT_BF16 && (a.ld_out & 7) == 0;
  const bool trans = a.out_mode == OUT_BF16_T;
  if (!plain && !trans) return false;
  const bool any_tr = trans || a.out2 != nullptr;
  if (any_tr) {      // transposed tiles: whole tiles inside one batch element, 16-byte rows, no residual (gemm.hip's `tr` conditions)
    const int tld = a.out2 ? a.ld_out2 : a.ld_out;
    if (a.resid || (tld & 7) || a.rows_per_b % PBM != 0 || a.gstat || a.rowstat) return false;
    if (a.out2 && (a.n_split <= 0 || a.n_split % PBN != 0)) return false;
  }
  if (a.resid && ((a.ld_res & 7) || (double)a.M * a.ld_res * 2.0 >= 4.0e9)) return false;
  double amax = 0.0;
  for (int i = 0; i < a.nplain; ++i) amax = std::max(amax, (double)a.M * a.p_c[i] * 2.0);
  if (amax >= 4.0e9 || (double)a.N * a.ldw * 2.0 >= 4.0e9) return false;
  if (a.rowvec && !(a.rv_ld == 0 || a.rows_per_b % PBM == 0)) return false;       // the tile's rows share one row vector
  if (a.rowvec && ((a.rv_ld & 3) || (a.rv_off & 3))) return false;
  if (a.ln_stat && (a.rowvec || a.resid || a.ln_parts < 1 || a.ln_parts > P_MAX_LN_PARTS || !a.ln_s)) return false;
  if (a.w_img_bs && (a.rows_per_b % PBM != 0)) return false;
  if (a.gstat && (a.gstat_cpg <= 0 || PBN % a.gstat_cpg != 0 || a.gstat_hw % PBM != 0)) return false;
  return true;
}

int gemm_persist_launch(const GemmArgs& a, hipStream_t stream, int max_wgs) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    attr_set = true;
  }
  const int tiles = (a.M / PBM) * (a.N / PBN);
  int wgs = std::min(tiles, max_wgs > 0 ? max_wgs : 256);
  if (wgs >= 8) wgs &= ~7;                               // a multiple of the 8 XCDs: tile v and v + grid then run on the same XCD, as in the tile launch
  GemmArgs b = a;
  static const int dbg = [] { const char* e = getenv("DFH_PERSIST_DBG"); return e ? atoi(e) : 0; }();      // timing probe only (results are wrong): 1 = no output stores (the first form also had 2 = no waits, 4 = no refills: profiles/r06/persist_timing_probes.txt)
  b.pad0 = dbg;
  hipLaunchKernelGGL(gemm_persist_kernel, dim3(wgs), dim3(PNWV * 64), P_LDS, stream, b);
  return check_launch("gemm_persist_kernel");
}

}  // namespace dfh
