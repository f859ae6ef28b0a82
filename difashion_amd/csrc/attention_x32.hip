// Fused softmax(Q K^T / sqrt(d)) V on v_mfma_f32_32x32x16_bf16 for the U-Net's long self-attention launches
// (SD-1.5: (N, d) = (4096, 40) and (1024, 80): 90 % of the attention time of a sampling step).
// Reference call sites: diffusers Attention (attn1 of BasicTransformerBlock) reached via
// DiFashion/models/difashion.py:249-253,518-523 (xformers memory_efficient_attention there, difashion.py:118).
//
// Why a second kernel (attention.hip stays for head dims > 80, short / ragged key ranges and the training LSE path it
// was tuned for): at d = 40 the 16x16x32 kernel pads the contraction 40 -> 64 and spends 165 VALU instructions per 32 x 64
// score tile -- the softmax VALU work, not the MFMA pipe, paces it (round-1 PMC: VALU active 70 %, MFMA busy 34 %).
// This kernel removes both costs:
//   * 32x32x16 MFMA: the contraction is padded to a multiple of 16 (40 -> 48), a 32x32x16 issues at 32 cycles
//     (1024 flop/cycle/SIMD) where the 16x16x32 measured 19.5-20 (820 flop/cycle) -- profiles/r02/mfma_rate2.txt.
//   * scores are computed transposed, S^T = K . Q^T (keys = rows): a lane owns ONE query (column lane & 31) and
//     16 keys per 32-key block, so the softmax needs no cross-lane traffic, and the C layout of S^T IS the B layout of
//     the P^T operand of O^T = V^T . P^T (lane half hi holds k-slots 8 hi .. 8 hi + 7 of every 16-key MFMA) once the K
//     rows of a block are read in the order key = swap_bits_2_3(slot): probabilities never leave registers and need
//     no permutes at all.
//   * the two free contraction slots of the padding carry the softmax bookkeeping: K column d = D is 1.0 and
//     Q column D is -m (the running max, kept bf16-exact), Q is pre-scaled by scale * log2(e) -> the MFMA delivers
//     s * c - m directly: no multiply / subtract per score.  K column D+1 is 1.0 for keys beyond Nk and Q column D+1 is
//     -30000 -> ragged key ranges are masked by the MFMA as well.  V^T row D is all ones, so O^T row D accumulates the
//     softmax denominator: no per-score add.
//   * what is left per score: one v_exp_f32, half a v_cvt_pk_bf16_f32, half a v_max3_f32 (the deferred-max check:
//     the running max only moves when some score exceeds it by 2^8, a wave-uniform rare branch).
//   * 4 waves x 64 queries (two 32-query blocks: every K / V^T fragment read feeds two MFMAs) = 256 queries per
//     workgroup, ~240 VGPRs, two workgroups per CU: while one wave of a SIMD is in its exp / pack phase its neighbour
//     issues MFMAs.  K / V^T tiles of 64 keys are double-buffered in LDS (issue-early / write-late register staging through
//     buffer loads, one barrier per tile), 16-byte slots XOR-swizzled so every ds_read_b128 fragment read is conflict-free.
//
// What bounds it (profiles/r02/coissue.txt, valu_rate.txt -- scripts/probes/coissue.hip): per SIMD a v_exp_f32 occupies the
// VALU port for ~8.7 cycles and any other VALU instruction for ~4.6, whichever wave issues it, and one wave mixing MFMAs with
// exps overlaps them only partially (7 MFMA + 16 exp + 8 cvt: 328 cycles alone, 279 each for two co-resident waves, against
// 224 of matrix-pipe time).  A 64-query x 64-key tile costs 28 MFMAs = 896 matrix-pipe cycles and 64 exp + 32 cvt + ~30 other
// = ~840 VALU-port cycles: the two are balanced, so d = 40 attention cannot approach the MFMA roof the way d = 128 does
// (one exp per 4 x 56 padded flop instead of per 4 x 128).  Measured: 2130 cycles per tile with one wave per SIMD, 1830 per SIMD
// with two; a software-pipelined variant (S of block i+1 and P.V of block i-1 issued around the exps of block i, LDS-DMA
// staging) and s_setprio around the MFMA clusters were both measured and were not faster (555 / 475 vs 470 us).
#include "dfh_common.h"
#include "attention.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int KVT = 64;               // keys per LDS tile (two 32-key MFMA blocks)
constexpr float THR = 8.0f;           // deferred max: rescale when a score exceeds the running max by 2^8 (log2 domain)
constexpr float MASK_Q = -30000.0f;   // Q-side value of the mask slot (bf16-representable to 3 digits; exp2 -> 0)

template <int D> struct X32Geom {
  static_assert(D % 8 == 0, "head dim must be a multiple of 8");
  static constexpr int DCH = D / 8;                        // 16-byte data chunks per K row
  static constexpr int KS = (D + 2 + 15) / 16;             // 16-deep contraction steps incl. the two bookkeeping slots
  static constexpr int NCH = 2 * KS;                       // chunks per K row in LDS (data + pad chunk + zero chunks)
  static constexpr int DB = (D + 1 + 31) / 32;             // 32-row blocks of O^T incl. the ones row
  static constexpr int KROW = NCH <= 8 ? 128 : 256;        // K row stride (bytes)
  static constexpr int K_BYTES = KVT * KROW;
  static constexpr int VROWS = D + 2;                      // data rows, the ones row (D), the zero row (D + 1)
  static constexpr int V_BYTES = VROWS * 128;
  static constexpr int BUF = K_BYTES + V_BYTES;
  static constexpr int PAD_KS = DCH / 2, PAD_HI = DCH & 1; // fragment (k-step, lane half) holding slots D, D + 1
  static constexpr int NKI = (KVT * DCH + 255) / 256;      // K staging chunks per thread
  static constexpr int NVI = (D * 8 + 255) / 256;          // V^T staging chunks per thread
  static constexpr int LR = D % 32;                        // row of the softmax denominator inside O^T block D / 32
  static constexpr int L_HI = (LR >> 2) & 1, L_REG = (LR & 3) | ((LR >> 3) << 2);
  static_assert((D % 8) == 0 && (D + 1) / 8 == DCH, "slots D, D+1 must share one chunk");
};

template <int KROW> DFH_DEVICE int k_swz(int key) { return KROW == 128 ? ((key >> 1) & 7) : (key & 15); }
DFH_DEVICE int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// QB = 32-query blocks per wave (2: 256 queries per workgroup; 1: 128, for head dims whose accumulators would not fit)
template <int D, int QB, int MINW, bool PROF = false>
__global__ __launch_bounds__(256, MINW) void attention_x32_kernel(const AttnArgs a) {
  constexpr int NBUF = 2;                                  // K / V^T tile buffers in LDS
  using G = X32Geom<D>;
  constexpr int KS = G::KS, DCH = G::DCH, NCH = G::NCH, KROW = G::KROW, NKI = G::NKI, NVI = G::NVI;
  // LSUM (head dims that are whole 32-row blocks of O^T, d = 64: SD-2-base): the ones row of V^T would open a block of its own -- a third
  // of the P.V MFMAs and 16 accumulator registers per query block for ONE useful row (the instantiation spilled 13 VGPRs).  The softmax
  // denominator is summed on the VALU instead: one v_dot2_f32_bf16 per packed pair of probabilities against (1, 1), i.e. the sum of the
  // ROUNDED probabilities the MFMA multiplies, exactly what the ones row delivers; each lane half sums the keys it holds.
  constexpr bool LSUM = D % 32 == 0;
  constexpr int DB = LSUM ? D / 32 : G::DB;
  constexpr int WQ = QB * 32;                              // queries per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;

  // one (batch, head) per group of consecutive logical blocks, and consecutive logical blocks on ONE XCD: the 16 workgroups
  // of a (batch, head) share its K / V^T through one L2 instead of fetching them on all eight
  const int nqb = (a.Nq + 4 * WQ - 1) / (4 * WQ);
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lb / nqb, qblk = lb - bh * nqb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = qblk * 4 * WQ + wave * WQ;

  const bf16_t* Qb = a.Q + (long)b * a.Nq * a.ldq + h * D;
  const bf16_t* Kb = a.K + (long)b * a.Nk * a.ldk + h * D;
  const bf16_t* Vb = a.Vt + (long)b * (a.vt_bstride ? a.vt_bstride : (long)a.H * D * a.ldvt) + (long)h * D * a.ldvt;

  // ---- Q fragments (B operand of S^T = K . Q^T): lane (query ql, half hi) holds d = 16 ks + 8 hi .. + 8,
  //      pre-scaled by scale * log2(e); the pad chunk holds {-m, MASK_Q, 0 ...}
  const float c = a.scale * 1.44269504088896340736f;
  uint4 qf[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + qb * 32 + ql;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 v = uint4{0u, 0u, 0u, 0u};
      const int ch = 2 * ks + hi;
      if (ch < DCH && q < a.Nq) {
        const uint4 raw = *(const uint4*)(Qb + (long)q * a.ldq + ch * 8);
        float f[8];
        unpack8(raw, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= c;
        v = pack8(f);
      } else if (ch == DCH) {
        v.x = pack2bf(0.0f, MASK_Q);
      }
      qf[qb][ks] = v;
    }
  }

  // ---- staging bookkeeping (fixed per thread).  Chunk ids wrap around the tile: the threads left over in the last round
  //      re-stage a chunk another thread stages too (same bytes to the same address) -- no divergent branch in the loop
  int k_key[NKI], k_goff[NKI], k_lds[NKI];
#pragma unroll
  for (int i = 0; i < NKI; ++i) {
    const int idx = (tid + i * 256) % (KVT * DCH);
    k_key[i] = idx / DCH;
    const int ch = idx - k_key[i] * DCH;
    k_goff[i] = k_key[i] * a.ldk + ch * 8;
    k_lds[i] = k_key[i] * KROW + ((ch ^ k_swz<KROW>(k_key[i])) << 4);
  }
  int v_goff[NVI], v_lds[NVI], v_k0[NVI];
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int idx = (tid + i * 256) % (D * 8);
    const int row = idx >> 3, ch = idx & 7;
    v_goff[i] = row * a.ldvt + ch * 8;
    v_k0[i] = ch * 8;
    v_lds[i] = G::K_BYTES + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
  }
  uint4 kreg[NKI], vreg[NVI];

  // K / V^T tiles are fetched with buffer loads: one descriptor per operand in SGPRs (built from wave-uniform values), a
  // 32-bit per-lane byte offset fixed for the kernel and the tile's byte offset in an SGPR -- no 64-bit address VGPRs in the
  // loop (as pointers they cost 8 VGPRs that hipcc spilled to scratch and reloaded, ~500 cycles each, at the top of every
  // tile).  The K descriptor ends with the last valid key row, so rows beyond Nk read as zeros without a predicate.
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const unsigned k_bytes = a.Nk > 0 ? (unsigned)(((long)(a.Nk - 1) * a.ldk + D) * 2) : 0u;
  const unsigned v_bytes = (unsigned)(((long)(D - 1) * a.ldvt + a.ldvt) * 2);
  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, k_bytes, 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, v_bytes, 0x00020000);
  auto load_tile = [&](int kv0, auto full_c) {   // global -> registers (zeros beyond Nk)
    constexpr bool FULL = decltype(full_c)::value;
    const int k_soff = kv0 * a.ldk * 2, v_soff = kv0 * 2;                    // wave-uniform byte offsets of the tile
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, k_goff[i] * 2, k_soff, 0);
      kreg[i] = uint4{r[0], r[1], r[2], r[3]};
    }
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
      const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, v_goff[i] * 2, v_soff, 0);
      uint4 v = uint4{r[0], r[1], r[2], r[3]};
      if (!FULL) {
        const int k0 = kv0 + v_k0[i];
        if (k0 >= a.Nk) v = uint4{0u, 0u, 0u, 0u};
        else if (k0 + 8 > a.Nk) {   // ragged tail: zero the padding keys (they may hold anything, NaN included)
          const int valid = a.Nk - k0;
          uint32_t* w = (uint32_t*)&v;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e >= valid) w[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
        }
      }
      vreg[i] = v;
    }
  };
  auto store_tile = [&](int buf) {               // registers -> swizzled LDS image
    unsigned char* Bs = smem + buf * G::BUF;
#pragma unroll
    for (int i = 0; i < NKI; ++i) *(uint4*)(Bs + k_lds[i]) = kreg[i];
#pragma unroll
    for (int i = 0; i < NVI; ++i) *(uint4*)(Bs + v_lds[i]) = vreg[i];
  };
  // constant parts of a buffer: K chunks DCH .. NCH-1 ({1, mask, 0 ..} then zeros), V^T ones row and zero row.
  // first_masked = first key of the tile that lies beyond Nk (KVT: none)
  auto store_const = [&](int buf, int first_masked) {
    unsigned char* Ks = smem + buf * G::BUF;
    unsigned char* Vs = Ks + G::K_BYTES;
    for (int idx = tid; idx < KVT * (NCH - DCH); idx += 256) {
      const int key = idx / (NCH - DCH), ch = DCH + (idx - key * (NCH - DCH));
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (ch == DCH) v.x = key >= first_masked ? 0x3f803f80u : 0x00003f80u;      // {1.0, mask}
      *(uint4*)(Ks + key * KROW + ((ch ^ k_swz<KROW>(key)) << 4)) = v;
    }
    if (tid < 16) {
      const int row = D + (tid >> 3), ch = tid & 7;
      const uint32_t w = row == D ? 0x3f803f80u : 0u;
      *(uint4*)(Vs + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = uint4{w, w, w, w};
    }
  };

  // ---- fragment read offsets (fixed per lane)
  const int kkey = swap23(ql);                    // K row of S^T row slot ql inside a 32-key block
  int k_off[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_off[ks] = kkey * KROW + (((2 * ks + hi) ^ k_swz<KROW>(kkey)) << 4);
  int v_row[DB], v_sw[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    const int pr = min(db * 32 + ql, D + 1);      // rows beyond the ones row read the zero row
    v_row[db] = pr * 128; v_sw[db] = (pr >> 1) & 7;
  }

  f32x16_t o[DB][QB];
  float m_run[QB];
  float l_acc[QB];                                           // LSUM: this lane half's part of the softmax denominator
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
  auto den = [&](int qb) -> float {                          // the denominator as this lane sees it (the caller combines the halves)
    if constexpr (LSUM) return l_acc[qb];
    else return o[D / 32][qb][G::L_REG];
  };
  auto den_total = [&](int qb) -> float {
    const float lv = den(qb), lo = lane_xor32(lv);
    if constexpr (LSUM) return lv + lo;                      // the two halves hold different keys of the query
    else return hi == G::L_HI ? lv : lo;                     // lanes of half L_HI hold it; the other half holds a zero row of O^T
  };
  const int ntiles = (a.Nk + KVT - 1) / KVT;
  const int tail_valid = a.Nk - (ntiles - 1) * KVT;          // valid keys of the last tile (KVT when Nk % 64 == 0)
  const int nfast = a.Nk / KVT - 1;                          // tiles whose successor is a full tile
  unsigned* poison_flag = (unsigned*)(smem + NBUF * G::BUF); // one word behind the tile buffers
  if (tid == 0) *poison_flag = 0u;

  // One pass over the keys.  SAFE = false (the fast pass): after the first tile there is NO per-score max -- the running
  // offset m only has to keep 2^(s - m) inside the fp32 / bf16 exponent range (both carry 8 exponent bits, so the relative
  // precision of P does not depend on its magnitude), which one compare per tile on the softmax denominator (row D of O^T)
  // guarantees: when it passes 2^24 the wave divides O by an exact power of two and raises m by that integer (m stays an
  // integer, hence bf16-exact in its contraction slot).  A score more than ~100 log2 units above m inside ONE tile would
  // overflow the exp before the compare sees it: the denominator then reads >= 2^100 / inf / NaN, the wave raises the
  // workgroup's poison flag and the whole workgroup repeats the pass with SAFE = true -- the exact deferred-max scheme
  // (lane-local v_max3 tree before the exp, rescale when a score exceeds m by 2^8) that is also used for tile 0.
  auto pass = [&](auto safe_c) {
    constexpr bool SAFE = decltype(safe_c)::value;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][qb][r] = 0.f;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      m_run[qb] = 0.f; l_acc[qb] = 0.f;
      if (hi == G::PAD_HI) qf[qb][G::PAD_KS].x = pack2bf(0.0f, MASK_Q);
    }
    bool poison = false;
    store_const(0, ntiles == 1 ? tail_valid : KVT);
    store_const(1, ntiles == 2 ? tail_valid : KVT);
    load_tile(0, std::false_type{});
    store_tile(0);
    __syncthreads();

    // precheck_c: lane-local max tree + deferred-max rescale BEFORE the exp (tile 0 and the SAFE pass)
    auto tile = [&](int t, auto fast_c, auto precheck_c) {
      constexpr bool FAST = decltype(fast_c)::value;            // this tile's successor exists and is a full tile
      constexpr bool PRE = decltype(precheck_c)::value;
      const int kv0 = t * KVT;
      const bool more = FAST || t + 1 < ntiles;
      if (FAST) load_tile(kv0 + KVT, std::true_type{});
      else if (more) load_tile(kv0 + KVT, std::false_type{});
      const unsigned char* Ks = smem + (t & 1) * G::BUF;
      const unsigned char* Vs = Ks + G::K_BYTES;
      const bool stamp = PROF && a.prof && blockIdx.x == 0 && wave == 0 && lane == 0 && t < 64;
      auto mark = [&](int i) {          // diagnosis build only: wave-issue timeline (s_memtime = shader cycles)
        if (PROF) { __builtin_amdgcn_sched_barrier(0); if (stamp) a.prof[t * 8 + i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
      };
      mark(0);

      // ---- four query blocks per wave, ONE wave per SIMD (QB = 4, steady-state tiles): nothing else on the SIMD can fill the matrix pipe
      //      while this wave runs its exponentials, so the tile is software-pipelined INSIDE the wave, in source order and pinned there
      //      by scheduling barriers: every MFMA of a product is followed by a slice of the exponentials of the PREVIOUS score block --
      //      S(kb 0) | S(kb 1) + exp of the first key half of block 0 | P.V(block 0, keys 0-15) + the second half | P.V(block 0, keys
      //      16-31) + exp(block 1, first half) | P.V(block 1, 0-15) + exp(block 1, second half) | P.V(block 1, 16-31).  An MFMA occupies
      //      the pipe for 32 cycles after a 4-cycle issue; the VALU instructions behind it issue in its shadow.  Each K / V^T fragment
      //      read feeds FOUR MFMAs (14 reads per 56 MFMAs; the two-block kernel: 14 per 28).
      if constexpr (QB == 4 && !PRE) {
        f32x16_t s[2][QB];
        uint32_t pw[2][QB][8];
        // unit u of score block kb: query block u & 3, register pair u >> 2 -- units 0..15 are the keys of the first 16-key MFMA (m2 = 0)
        auto exp_unit = [&](int kb, int u) {
          const int qb = u & 3, pi = u >> 2;
          pw[kb][qb][pi] = pack2bf(__builtin_amdgcn_exp2f(s[kb][qb][2 * pi]), __builtin_amdgcn_exp2f(s[kb][qb][2 * pi + 1]));
          if constexpr (LSUM)
            l_acc[qb] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, pw[kb][qb][pi]), __builtin_bit_cast(bf16x2v_t, 0x3f803f80u), l_acc[qb], false);
        };
        auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
        // S of score block kb, with `units` exponential units of block ekb (from u0 on) spread behind its MFMAs
        auto s_block = [&](int kb, int ekb, int u0, int units) {
          bf16x8_t kf[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const bf16x8_t*)(Ks + kb * 32 * KROW + k_off[ks]);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
              // scores in VGPRs (the exponentials read them), the Q fragments -- read-only B operands, 48 registers -- from the AGPR half.
              // No VALU instruction reads a score block before at least four further MFMAs have issued behind the one that completed
              // it (the pipe is serial: 32 cycles each), so the XDL-write -> VALU-read wait states the compiler cannot see are covered.
              if (ks == 0)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s[kb][qb]) : "v"(kf[ks]), "a"(__builtin_bit_cast(bf16x8_t, qf[qb][ks])));
              else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[kb][qb]) : "v"(kf[ks]), "a"(__builtin_bit_cast(bf16x8_t, qf[qb][ks])));
              if (units > 0) {
                fence();
                const int from = ((ks * QB + qb) * units) / (KS * QB), upto = ((ks * QB + qb + 1) * units) / (KS * QB);
#pragma unroll
                for (int u = from; u < upto; ++u) exp_unit(ekb, u0 + u);
                fence();
              }
            }
        };
        // P.V of (score block kb, 16-key half m2), with `units` exponential units of block ekb (from u0 on) behind its MFMAs
        auto pv_half = [&](int kb, int m2, int ekb, int u0, int units) {
          bf16x8_t vf[DB];
#pragma unroll
          for (int db = 0; db < DB; ++db) vf[db] = *(const bf16x8_t*)(Vs + v_row[db] + (((kb * 4 + m2 * 2 + hi) ^ v_sw[db]) << 4));
#pragma unroll
          for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
              const uint4 pv = uint4{pw[kb][qb][4 * m2], pw[kb][qb][4 * m2 + 1], pw[kb][qb][4 * m2 + 2], pw[kb][qb][4 * m2 + 3]};
              // the O^T accumulators (128 registers) are pinned in the AGPR half of the register file: the VALU never touches them inside
              // the loop (one denominator register per query block aside), while S^T -- which the exponentials read -- stays in VGPRs.
              // The builtin leaves that choice to one per-function switch; with both accumulator sets in VGPRs the allocator shuffled
              // ~300 v_accvgpr_read / write / mov per tile through the VALU this pipeline is built to keep free.
              asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[db][qb]) : "v"(vf[db]), "v"(__builtin_bit_cast(bf16x8_t, pv)));
              if (units > 0) {
                fence();
                const int from = ((db * QB + qb) * units) / (DB * QB), upto = ((db * QB + qb + 1) * units) / (DB * QB);
#pragma unroll
                for (int u = from; u < upto; ++u) exp_unit(ekb, u0 + u);
                fence();
              }
            }
        };
        fence();
        s_block(0, 0, 0, 0);
        fence();
        s_block(1, 0, 0, 16);
        mark(1);
        pv_half(0, 0, 0, 16, 16);
        pv_half(0, 1, 1, 0, 16);
        mark(2);
        pv_half(1, 0, 1, 16, 16);
        pv_half(1, 1, 0, 0, 0);
        fence();
        // the compiler's hazard recogniser does not look inside inline asm: a VALU read of an accumulator (the denominator check below)
        // needs 18 wait states behind the 16-pass MFMA that wrote it
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        mark(3);
      } else {
      // ---- S^T = K . Q'^T - m : [kb] 32 keys x [qb] 32 queries
      f32x16_t s[2][QB];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t kf = *(const bf16x8_t*)(Ks + kb * 32 * KROW + k_off[ks]);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            if (ks == 0) {
              f32x16_t z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.f;
              s[kb][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, __builtin_bit_cast(bf16x8_t, qf[qb][ks]), z, 0, 0, 0);
            } else {
              s[kb][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, __builtin_bit_cast(bf16x8_t, qf[qb][ks]), s[kb][qb], 0, 0, 0);
            }
          }
        }
      }
      mark(1);
      if (PRE) {
        // ---- deferred max: lane-local maxima (a tree: four independent chains per query block), one wave-uniform test
        float mx[QB];
        bool over = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          float c4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            c4[j] = fmaxf(s[0][qb][4 * j], s[1][qb][4 * j]);
#pragma unroll
            for (int r = 1; r < 4; ++r) c4[j] = fmaxf(fmaxf(c4[j], s[0][qb][4 * j + r]), s[1][qb][4 * j + r]);
          }
          mx[qb] = fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3]));
          over |= mx[qb] > THR;
        }
        if (t == 0 || __any(over)) {
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const float ml = fmaxf(mx[qb], lane_xor32(mx[qb]));       // both halves of the query's column
            float m_new = m_run[qb] + ml;
            if (t > 0) m_new = fmaxf(m_new, m_run[qb]);
            // the running offset rides in a bf16 contraction slot of Q: keep it bf16-exact; the fast pass needs an integer
            m_new = bf2f(f2bf(SAFE ? m_new : ceilf(m_new)));
            const float delta = m_new - m_run[qb];
            m_run[qb] = m_new;
            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
              for (int r = 0; r < 16; ++r) s[kb][qb][r] -= delta;
            if (t > 0) {                                               // tile 0: O is still zero (and alpha may overflow)
#pragma unroll
              for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][qb][r] *= alpha;
              l_acc[qb] *= alpha;
            }
            if (hi == G::PAD_HI) qf[qb][G::PAD_KS].x = pack2bf(-m_new, MASK_Q);
          }
        }
      }
      // ---- P = 2^S, packed in place into the B fragments of O^T += V^T . P^T
      uint32_t pw[2][QB][8];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            pw[kb][qb][r >> 1] = pack2bf(__builtin_amdgcn_exp2f(s[kb][qb][r]), __builtin_amdgcn_exp2f(s[kb][qb][r + 1]));
            if constexpr (LSUM)
              l_acc[qb] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, pw[kb][qb][r >> 1]), __builtin_bit_cast(bf16x2v_t, 0x3f803f80u),
                                                          l_acc[qb], false);
          }
      mark(2);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int db = 0; db < DB; ++db) {
            const bf16x8_t vf = *(const bf16x8_t*)(Vs + v_row[db] + (((kb * 4 + m2 * 2 + hi) ^ v_sw[db]) << 4));
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
              const uint4 pv = uint4{pw[kb][qb][4 * m2], pw[kb][qb][4 * m2 + 1], pw[kb][qb][4 * m2 + 2], pw[kb][qb][4 * m2 + 3]};
              o[db][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8_t, pv), o[db][qb], 0, 0, 0);
            }
          }
      mark(3);
      }
      if (more) {
        store_tile((t + 1) & 1);                     // the other buffer: last read one barrier ago
        // a ragged last tile changes the mask column of the buffer it lands in (buffers 0 / 1 were initialised for tiles 0 / 1)
        if (!FAST && t + 2 == ntiles && t + 1 >= 2 && tail_valid != KVT) store_const((t + 1) & 1, tail_valid);
      }
      if (!PRE) {
        // ---- fast pass: watch the denominator (lanes of half L_HI hold it; the other half holds a zero row of O^T)
        bool big = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float lv = den(qb);
          big |= !(lv <= 16777216.0f);               // 2^24; also true for NaN
        }
        if (__any(big)) {
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const float l = den_total(qb);
            poison |= !(l < 1.2676506e30f);          // 2^100: an exp may already have overflowed
            // exponent of l as an integer-valued float, kept a multiple of the bf16 spacing of the new m
            float m_new = m_run[qb] + (l > 2.0f ? floorf(__builtin_amdgcn_logf(l)) : 0.0f);      // v_log_f32 = log2
            m_new = bf2f(f2bf(m_new));
            const float delta = m_new - m_run[qb];   // an integer >= 0 (both are bf16-exact integers)
            m_run[qb] = m_new;
            const float alpha = __builtin_amdgcn_exp2f(-delta);       // exact power of two
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[db][qb][r] *= alpha;
            l_acc[qb] *= alpha;
            if (hi == G::PAD_HI) qf[qb][G::PAD_KS].x = pack2bf(-m_new, MASK_Q);
          }
        }
      }
      mark(4);
      __syncthreads();
      mark(5);
    };
    if (SAFE) {
      int t = 0;
      for (; t < nfast; ++t) tile(t, std::true_type{}, std::true_type{});
      for (; t < ntiles; ++t) tile(t, std::false_type{}, std::true_type{});
    } else {
      if (nfast > 0) tile(0, std::true_type{}, std::true_type{});
      else tile(0, std::false_type{}, std::true_type{});
      int t = 1;
      for (; t < nfast; ++t) tile(t, std::true_type{}, std::false_type{});
      for (; t < ntiles; ++t) tile(t, std::false_type{}, std::false_type{});
    }
    return poison;
  };


  const bool poisoned = pass(std::false_type{});
  if (__any(poisoned) && lane == 0) *poison_flag = 1u;
  __syncthreads();
  if (*poison_flag) {            // workgroup-uniform and essentially never: an in-tile score jump of > 2^100
    __syncthreads();
    (void)pass(std::true_type{});
  }

  // ---- normalise and store: lane (query ql, half hi) holds O[q][d = 32 db + 8 (r >> 2) + 4 hi + (r & 3)]
  const float qmul = attn_qmul(a, b);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l = den_total(qb);
    const float inv = qmul / l;
    const int q = q0 + qb * 32 + ql;
    if (q >= a.Nq) continue;
    if (a.lse && hi == 0) a.lse[((long)b * a.H + h) * a.Nq + q] = m_run[qb] + __builtin_amdgcn_logf(l);   // v_log_f32 = log2
    const long orow = ((long)b * a.Nq + q) * a.ldo + h * D;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + g * 8 + hi * 4;
        if (d0 < D) attn_store4(a, orow, d0, o[db][qb][4 * g] * inv, o[db][qb][4 * g + 1] * inv, o[db][qb][4 * g + 2] * inv, o[db][qb][4 * g + 3] * inv);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Short key ranges: the cross-attention over the 77 text tokens (attn2 of BasicTransformerBlock, reference difashion.py:518-523).
// With at most 2 * KVT keys the whole K / V^T of a (batch, head) fits the two LDS buffers: they are staged ONCE, and then every wave
// walks `qrep` blocks of 32 * QB queries with no further barrier -- Q in, two score tiles, softmax, P.V, O out.  In the streaming
// kernel above such a launch is all prologue and epilogue (one dependent chain of Q load -> tile staging -> barrier -> ... per 256
// queries: 44 us for 84 MB at the 64x64 level); here the chain is paid once per workgroup and the per-block work of the four
// waves of a workgroup (and of the two workgroups of a CU) overlaps freely.  The softmax is the exact deferred-max form of the
// streaming kernel's SAFE pass on both tiles (two tiles: nothing to win from the fast pass), same fragment layouts, same numerics.
#ifndef XS_PREFETCH_Q
#define XS_PREFETCH_Q(QB) ((QB) == 1)
#define XS_STAGE_O(QB) true
#endif
template <int D, int QB>
__global__ __launch_bounds__(256, 2) void attention_xs_kernel(const AttnArgs a, const int qrep) {
  using G = X32Geom<D>;
  constexpr int KS = G::KS, DB = G::DB, DCH = G::DCH, NCH = G::NCH, KROW = G::KROW, NKI = G::NKI, NVI = G::NVI;
  constexpr int WQ = QB * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int per_wg = 4 * WQ * qrep;
  const int nqb = (a.Nq + per_wg - 1) / per_wg;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lb / nqb, qblk = lb - bh * nqb;
  const int b = bh / a.H, h = bh - b * a.H;
  const bf16_t* Qb = a.Q + (long)b * a.Nq * a.ldq + h * D;
  const bf16_t* Kb = a.K + (long)b * a.Nk * a.ldk + h * D;
  const bf16_t* Vb = a.Vt + (long)b * (a.vt_bstride ? a.vt_bstride : (long)a.H * D * a.ldvt) + (long)h * D * a.ldvt;
  const int ntiles = (a.Nk + KVT - 1) / KVT;                 // 1 or 2 (the launcher guarantees Nk <= 2 * KVT)
  const int tail_valid = a.Nk - (ntiles - 1) * KVT;

  // ---- stage every key once: tile t -> buffer t (ragged tail zero-filled, mask column set), constant chunks as in the streaming kernel
  for (int t = 0; t < ntiles; ++t) {
    unsigned char* Ks = smem + t * G::BUF;
    unsigned char* Vs = Ks + G::K_BYTES;
    const int kv0 = t * KVT, first_masked = (t == ntiles - 1) ? tail_valid : KVT;
    for (int idx = tid; idx < KVT * DCH; idx += 256) {
      const int key = idx / DCH, ch = idx - key * DCH;
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (kv0 + key < a.Nk) v = *(const uint4*)(Kb + (long)(kv0 + key) * a.ldk + ch * 8);
      *(uint4*)(Ks + key * KROW + ((ch ^ k_swz<KROW>(key)) << 4)) = v;
    }
    for (int idx = tid; idx < KVT * (NCH - DCH); idx += 256) {
      const int key = idx / (NCH - DCH), ch = DCH + (idx - key * (NCH - DCH));
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (ch == DCH) v.x = key >= first_masked ? 0x3f803f80u : 0x00003f80u;      // {1.0, mask}
      *(uint4*)(Ks + key * KROW + ((ch ^ k_swz<KROW>(key)) << 4)) = v;
    }
    for (int idx = tid; idx < D * 8; idx += 256) {
      const int row = idx >> 3, ch = idx & 7;
      const int k0 = kv0 + ch * 8;
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (k0 < a.Nk) {
        v = *(const uint4*)(Vb + (long)row * a.ldvt + k0);       // ldvt >= roundup8(Nk): the chunk exists
        if (k0 + 8 > a.Nk) {                                      // ragged tail: zero the padding keys (NaN-proof)
          const int valid = a.Nk - k0;
          uint32_t* w = (uint32_t*)&v;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e >= valid) w[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
        }
      }
      *(uint4*)(Vs + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v;
    }
    if (tid < 16) {
      const int row = D + (tid >> 3), ch = tid & 7;
      const uint32_t w = row == D ? 0x3f803f80u : 0u;
      *(uint4*)(Vs + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = uint4{w, w, w, w};
    }
  }
  __syncthreads();

  // ---- fragment read offsets (fixed per lane), as in the streaming kernel
  const int kkey = swap23(ql);
  int k_off[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_off[ks] = kkey * KROW + (((2 * ks + hi) ^ k_swz<KROW>(kkey)) << 4);
  int v_row[DB], v_sw[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    const int pr = min(db * 32 + ql, D + 1);
    v_row[db] = pr * 128; v_sw[db] = (pr >> 1) & 7;
  }
  const float c = a.scale * 1.44269504088896340736f;

  // the Q rows of the NEXT block are requested before the current block is computed: load -> scores -> softmax -> P.V -> store was one
  // dependent chain per block and wave, with nothing in flight while it computed
  constexpr bool PREFETCH_Q = XS_PREFETCH_Q(QB), STAGE_O = XS_STAGE_O(QB);
  uint4 qraw[QB][KS];
  auto fetch_q = [&](int q0) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int q = q0 + qb * 32 + ql;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int ch = 2 * ks + hi;
        qraw[qb][ks] = uint4{0u, 0u, 0u, 0u};
        if (ch < DCH && q < a.Nq) qraw[qb][ks] = *(const uint4*)(Qb + (long)q * a.ldq + ch * 8);
      }
    }
  };
  if (PREFETCH_Q) fetch_q(qblk * per_wg + wave * WQ);
  for (int rep = 0; rep < qrep; ++rep) {
    const int q0 = qblk * per_wg + rep * 4 * WQ + wave * WQ;
    if (q0 >= a.Nq) break;                                    // wave-uniform
    if (!PREFETCH_Q) fetch_q(q0);
    uint4 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int q = q0 + qb * 32 + ql;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 v = uint4{0u, 0u, 0u, 0u};
        const int ch = 2 * ks + hi;
        if (ch < DCH && q < a.Nq) {
          float f[8];
          unpack8(qraw[qb][ks], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= c;
          v = pack8(f);
        } else if (ch == DCH) {
          v.x = pack2bf(0.0f, MASK_Q);
        }
        qf[qb][ks] = v;
      }
    }
    if (PREFETCH_Q && rep + 1 < qrep) fetch_q(q0 + 4 * WQ);
    f32x16_t o[DB][QB];
    float m_run[QB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][qb][r] = 0.f;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) m_run[qb] = 0.f;

    for (int t = 0; t < ntiles; ++t) {
      const unsigned char* Ks = smem + t * G::BUF;
      const unsigned char* Vs = Ks + G::K_BYTES;
      // second 32-key block of the tile entirely beyond Nk (77 text tokens: keys 96..127): its scores are the mask value, its
      // probabilities zero -- no MFMA, no exponentials, no P.V for it (a quarter of the launch's matrix and VALU work)
      const bool kb1 = t * KVT + 32 < a.Nk;                  // workgroup-uniform
      f32x16_t s[2][QB];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if (kb == 1 && !kb1) {
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[1][qb][r] = MASK_Q;
          continue;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t kf = *(const bf16x8_t*)(Ks + kb * 32 * KROW + k_off[ks]);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            if (ks == 0) {
              f32x16_t z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.f;
              s[kb][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, __builtin_bit_cast(bf16x8_t, qf[qb][ks]), z, 0, 0, 0);
            } else {
              s[kb][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, __builtin_bit_cast(bf16x8_t, qf[qb][ks]), s[kb][qb], 0, 0, 0);
            }
          }
        }
      }
      // deferred max (exact form): lane-local maxima, rescale when some score exceeds the running offset by 2^8 (always on tile 0)
      float mx[QB];
      bool over = false;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float c4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c4[j] = fmaxf(s[0][qb][4 * j], s[1][qb][4 * j]);
#pragma unroll
          for (int r = 1; r < 4; ++r) c4[j] = fmaxf(fmaxf(c4[j], s[0][qb][4 * j + r]), s[1][qb][4 * j + r]);
        }
        mx[qb] = fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3]));
        over |= mx[qb] > THR;
      }
      if (t == 0 || __any(over)) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float ml = fmaxf(mx[qb], lane_xor32(mx[qb]));
          float m_new = m_run[qb] + ml;
          if (t > 0) m_new = fmaxf(m_new, m_run[qb]);
          m_new = bf2f(f2bf(m_new));                        // rides in a bf16 contraction slot of Q
          const float delta = m_new - m_run[qb];
          m_run[qb] = m_new;
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][qb][r] -= delta;
          if (t > 0) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[db][qb][r] *= alpha;
          }
          if (hi == G::PAD_HI) qf[qb][G::PAD_KS].x = pack2bf(-m_new, MASK_Q);
        }
      }
      uint32_t pw[2][QB][8];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if (kb == 1 && !kb1) continue;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; r += 2)
            pw[kb][qb][r >> 1] = pack2bf(__builtin_amdgcn_exp2f(s[kb][qb][r]), __builtin_amdgcn_exp2f(s[kb][qb][r + 1]));
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if (kb == 1 && !kb1) continue;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int db = 0; db < DB; ++db) {
            const bf16x8_t vf = *(const bf16x8_t*)(Vs + v_row[db] + (((kb * 4 + m2 * 2 + hi) ^ v_sw[db]) << 4));
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
              const uint4 pv = uint4{pw[kb][qb][4 * m2], pw[kb][qb][4 * m2 + 1], pw[kb][qb][4 * m2 + 2], pw[kb][qb][4 * m2 + 3]};
              o[db][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8_t, pv), o[db][qb], 0, 0, 0);
            }
          }
      }
    }
    // ---- normalise and store.  bf16 output: the wave's WQ x D block is turned through a wave-private LDS region so that a row leaves as
    // D / 8 consecutive 16-byte stores (one 2 * D-byte run per row) -- straight from the accumulator layout a row left in 16-byte pieces
    // from five different instructions, and this launch is nothing but Q in / O out (84 MB at the 64x64 level)
    if (STAGE_O && !a.O8) {
      unsigned char* st = smem + 2 * G::BUF + 16 + wave * (WQ * D * 2);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float lv = o[D / 32][qb][G::L_REG];
        const float lo = lane_xor32(lv);
        const float l = hi == G::L_HI ? lv : lo;
        const float inv = 1.0f / l;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d0 = db * 32 + g * 8 + hi * 4;
            if (d0 < D) {
              uint2 w;
              w.x = pack2bf(o[db][qb][4 * g] * inv, o[db][qb][4 * g + 1] * inv);
              w.y = pack2bf(o[db][qb][4 * g + 2] * inv, o[db][qb][4 * g + 3] * inv);
              *(uint2*)(st + (qb * 32 + ql) * (D * 2) + d0 * 2) = w;
            }
          }
      }
      constexpr int CH = D / 8;                                // 16-byte chunks per row
#pragma unroll
      for (int i0 = 0; i0 < WQ * CH; i0 += 64) {
        const int i = i0 + lane;
        const int row = i / CH, ch = i - row * CH;
        const int q = q0 + row;
        if (i < WQ * CH && q < a.Nq)
          *(uint4*)(a.O + ((long)b * a.Nq + q) * a.ldo + h * D + ch * 8) = *(const uint4*)(st + row * (D * 2) + ch * 16);
      }
      continue;
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float lv = o[D / 32][qb][G::L_REG];
      const float lo = lane_xor32(lv);
      const float l = hi == G::L_HI ? lv : lo;
      const float inv = attn_qmul(a, b) / l;
      const int q = q0 + qb * 32 + ql;
      if (q >= a.Nq) continue;
      const long orow = ((long)b * a.Nq + q) * a.ldo + h * D;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = db * 32 + g * 8 + hi * 4;
          if (d0 < D) attn_store4(a, orow, d0, o[db][qb][4 * g] * inv, o[db][qb][4 * g + 1] * inv, o[db][qb][4 * g + 2] * inv, o[db][qb][4 * g + 3] * inv);
        }
    }
  }
}

template <int D, int QB>
int launch_xs(const AttnArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * X32Geom<D>::BUF + 16 + 4 * QB * 32 * D * 2;      // + the four waves' output staging blocks
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attention_xs_kernel<D, QB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  // query blocks per wave: enough workgroups to give every CU its two, few enough to amortise the one-time K / V^T staging
  const long units = (long)a.B * a.H * ((a.Nq + 128 * QB - 1) / (128 * QB));
  int qrep = (int)std::max(1L, std::min(8L, units / 512));
  const int nqb = (a.Nq + 128 * QB * qrep - 1) / (128 * QB * qrep);
  dfh::ProfScope ps(dfh::PC_ATTN, 4.0 * a.B * a.H * (double)a.Nq * a.Nk * D,
                    2.0 * a.B * a.H * D * (2.0 * a.Nq + 2.0 * a.Nk), stream);
  hipLaunchKernelGGL((attention_xs_kernel<D, QB>), dim3(nqb * a.H * a.B), dim3(256), lds, stream, a, qrep);
  return dfh::check_launch("attention_xs_kernel");
}

template <int D, int QB, int MINW, bool PROF = false>
int launch_x32(const AttnArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * X32Geom<D>::BUF + 16;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attention_x32_kernel<D, QB, MINW, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int nqb = (a.Nq + 128 * QB - 1) / (128 * QB);
  dfh::ProfScope ps(dfh::PC_ATTN, 4.0 * a.B * a.H * (double)a.Nq * a.Nk * D,
                    2.0 * a.B * a.H * D * (2.0 * a.Nq + 2.0 * a.Nk), stream);
  hipLaunchKernelGGL((attention_x32_kernel<D, QB, MINW, PROF>), dim3(nqb * a.H * a.B), dim3(256), lds, stream, a);
  return dfh::check_launch("attention_x32_kernel");
}

}  // namespace

namespace dfh {

// 0 = not handled here (the caller falls back to attention_kernel)
bool attention_x32_eligible(const AttnArgs& a) {
  // 40 / 80: SD-1.5 (8 heads); 64: SD-2-base, the reference's own default (stabilityai/stable-diffusion-2-base, train.py:44,
  // inf4eval.py:65: head dim 64 at every level)
  if (a.D != 40 && a.D != 64 && a.D != 80) return false;
  // d = 64: only the long self-attention launches (64x64 level 478 -> 412 us, 32x32 level equal); below that and for the 77 text keys
  // the 16x16x32 kernel measures the same or a little better (profiles/r04/attn_sd2_microbench.txt)
  if (a.D == 64) return a.Nq >= 1024 && a.Nk >= 1024;
  return a.Nq >= 256 && a.Nk >= 64;
}

int attention_x32_launch(const AttnArgs& a, hipStream_t stream) {
  census(CK_ATTN_X32);
  // short key ranges (cross-attention): keys staged once, waves stream query blocks.  DFH_ATTN_XS=0 turns it off (A/B).
  static const bool xs_off = [] { const char* e = getenv("DFH_ATTN_XS"); return e && e[0] == '0'; }();
  if (!xs_off && a.Nk <= 2 * KVT && a.lse == nullptr) {
    if (a.D == 40) return launch_xs<40, 2>(a, stream);
    if (a.D == 64) return launch_xs<64, 1>(a, stream);
    if (a.D == 80) return launch_xs<80, 1>(a, stream);
  }
#ifdef DFH_PROBES   // experiment instantiations (one / four query blocks per wave, phase stamps): probe builds only (scripts/probes/Makefile)
  static const int variant = [] { const char* e = getenv("DFH_ATTN_VARIANT"); return e ? atoi(e) : 0; }();   // experiments
  switch (a.D) {
    case 40:
      if (variant == 9) {      // diagnosis: one launch with the phase stamps, printed as per-phase cycle averages
        static unsigned long long* buf = nullptr;
        if (!buf && hipMalloc((void**)&buf, 64 * 8 * 8) != hipSuccess) return -1;
        (void)hipMemsetAsync(buf, 0, 64 * 8 * 8, stream);
        AttnArgs b = a; b.prof = buf;
        const int rc = launch_x32<40, 2, 2, true>(b, stream);
        unsigned long long h[64 * 8];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
        double ph[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
        for (int t = 4; t < 60; ++t) {
          if (!h[t * 8] || !h[(t + 1) * 8]) continue;
          for (int i = 0; i < 5; ++i) ph[i] += (double)(h[t * 8 + i + 1] - h[t * 8 + i]);
          ph[5] += (double)(h[(t + 1) * 8] - h[t * 8 + 5]); ++n;
        }
        if (n) fprintf(stderr, "[attn prof] cycles per tile (wave 0 of workgroup 0, %d tiles): S-issue %.0f | exp+pack %.0f | PV-issue %.0f | "
                               "stage-store+check %.0f | barrier %.0f | loop-back %.0f\n", n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n);
        return rc;
      }
      if (variant == 1) return launch_x32<40, 1, 3>(a, stream);      // experiments: one 32-query block per wave at 3 / 4 waves per SIMD
      if (variant == 2) return launch_x32<40, 1, 4>(a, stream);
      // four blocks per wave, ONE wave per SIMD: 14 fragment reads per 56 MFMAs.  Round 2, plain order: 524 vs 458 us; round 5, with the in-wave
      // software pipeline of the steady-state tiles (see the kernel): 506 vs 473 us -- a wave does not overlap its own MFMAs with its own VALU issue
      if (variant == 3 && a.Nq >= 512 && a.Nk >= 128) return launch_x32<40, 4, 1>(a, stream);
      break;
    case 80:
      if (variant == 1) return launch_x32<80, 1, 3>(a, stream);
      break;
    default: break;
  }
#endif
  switch (a.D) {
    case 40: return launch_x32<40, 2, 2>(a, stream);
    case 64: {
      static const int qb64 = [] { const char* e = getenv("DFH_ATTN_QB64"); return e ? atoi(e) : 2; }();     // probe knob
      return qb64 == 1 ? launch_x32<64, 1, 2>(a, stream) : launch_x32<64, 2, 2>(a, stream);
    }
    case 80: return launch_x32<80, 1, 2>(a, stream);
    default: break;
  }
  set_error("attention_x32_launch: unsupported head dim");
  return -1;
}

}  // namespace dfh
