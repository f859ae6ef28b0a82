// Attention backward for the training step (torch autograd of diffusers Attention / xformers
// memory_efficient_attention reached through DiFashion/train.py:699).  Flash-style: the N x N matrices
// P and dS are recomputed tile by tile from Q, K, V, dO and the forward's per-row log-sum-exp; nothing
// quadratic is stored.  With S = scale * Q K^T, P = softmax(S), delta = rowsum(dO * O):
//     dV = P^T dO          dP = dO V^T          dS = P o (dP - delta)
//     dQ = scale * dS K    dK = scale * dS^T Q
//
// One kernel template, run twice (no atomics, deterministic):
//   pass dQ   : a wave owns 32 QUERY columns (registers: Q, dO fragments); K / V tiles stream through LDS
//   pass dK,dV: a wave owns 16/32 KEY columns (registers: K, V fragments); Q / dO tiles stream through LDS
// In both, the "row side" tiles X1 (K | Q) and X2 (V | dO) give  T = X1 . Y1^T  (scores) and
// U = X2 . Y2^T (dP) as 16x16x32 MFMAs with the rows read by ds_read_b128 in the same permuted order
// as the forward kernel, so a lane ends up with 8 CONSECUTIVE rows per 32-row chunk for its own column:
// P and dS go straight back in as MFMA B operands, while the A operands of the accumulations
// (X^T fragments: head-dim rows, tile rows as contraction) are read transposed from the same row-major
// LDS tiles with ds_read_b64_tr_b16.  Accumulators are transposed (out^T[d][col]) so a lane finishes with
// 4 consecutive head-dim values of its own query / key: 8-byte stores, no cross-lane traffic.
#include <cstdlib>
#include <type_traits>
#include "dfh_common.h"
#include "attention.h"

namespace {

constexpr int XT = 64;   // rows per streamed tile

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

DFH_DEVICE int row_rho(int r) { return (((r >> 3) & 3) << 2) | (r & 3); }

template <int D> struct BwdGeom {
  static constexpr int KS = (D + 31) / 32;
  static constexpr int DF = (D + 15) / 16;
  static constexpr int STR = D <= 64 ? 128 : (D <= 128 ? 256 : 512);   // tile row stride (bytes)
  static constexpr int TILE = XT * STR;
  static constexpr int BUF = 2 * TILE + 2 * XT * 4;                    // X1, X2, row stats (lse, delta)
};

template <int STR> DFH_DEVICE int swz(int row) {
  const int rho = row_rho(row);
  return STR == 128 ? ((rho >> 1) & 7) : rho;
}

// transposed fragment: 8 consecutive tile rows m0..m0+7 of head-dim column d0 + L (A operand rows = head dim)
template <int STR>
DFH_DEVICE bf16x8_t tr_frag(const unsigned char* tile, int m0, int d0, int L) {
  const int col = d0 + (L & 3) * 4;
  const int r0 = m0 + (L >> 2), r1 = r0 + 4;
  const unsigned char* p0 = tile + r0 * STR + ((((col >> 3)) ^ swz<STR>(r0)) << 4) + (col & 7) * 2;
  const unsigned char* p1 = tile + r1 * STR + ((((col >> 3)) ^ swz<STR>(r1)) << 4) + (col & 7) * 2;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// X32 variant: rows m0 + {0..3} and m0 + 8 + {0..3} of head-dim column d0 + L (the k-slot order the swapped 32x32 C layout leaves in a lane
// group), tile swizzled by (row >> 1) & 7
template <int STR>
DFH_DEVICE bf16x8_t tr_frag_x32(const unsigned char* tile, int m0, int d0, int L) {
  const int col = d0 + (L & 3) * 4;
  const int r0 = m0 + (L >> 2), r1 = r0 + 8;
  const unsigned char* p0 = tile + r0 * STR + ((((col >> 3)) ^ ((r0 >> 1) & 7)) << 4) + (col & 7) * 2;
  const unsigned char* p1 = tile + r1 * STR + ((((col >> 3)) ^ ((r1 >> 1) & 7)) << 4) + (col & 7) * 2;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// KV_SIDE = false: columns are queries (dQ pass); true: columns are keys (dK / dV pass)
template <int D, bool KV_SIDE, int NT, bool X32W = false>
__global__ __launch_bounds__(256, ((D <= 40 && !KV_SIDE) ? 3 : 1)) void attention_bwd_kernel(const AttnBwdArgs a) {
  using G = BwdGeom<D>;
  constexpr int KS = G::KS, DF = G::DF, STR = G::STR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ncols = KV_SIDE ? a.Nk : a.Nq;     // column entities (owned by lanes)
  const int nrows = KV_SIDE ? a.Nq : a.Nk;     // row entities (streamed)
  const int c0 = blockIdx.x * (4 * NT * 16) + wave * (NT * 16);

  // column-side operands (registers) and row-side operands (streamed)
  const bf16_t* Y1 = KV_SIDE ? a.K + (long)b * a.Nk * a.ldk + h * D : a.Q + (long)b * a.Nq * a.ldq + h * D;
  const bf16_t* Y2 = KV_SIDE ? a.V + (long)b * a.Nk * a.ldv + h * D : a.dO + (long)b * a.Nq * a.ldo + h * D;
  const int ldy1 = KV_SIDE ? a.ldk : a.ldq, ldy2 = KV_SIDE ? a.ldv : a.ldo;
  const bf16_t* X1 = KV_SIDE ? a.Q + (long)b * a.Nq * a.ldq + h * D : a.K + (long)b * a.Nk * a.ldk + h * D;
  const bf16_t* X2 = KV_SIDE ? a.dO + (long)b * a.Nq * a.ldo + h * D : a.V + (long)b * a.Nk * a.ldv + h * D;
  const int ldx1 = KV_SIDE ? a.ldq : a.ldk, ldx2 = KV_SIDE ? a.ldo : a.ldv;
  const float* lse = a.lse + ((long)b * a.H + h) * a.Nq;
  const float* dlt = a.delta + ((long)b * a.H + h) * a.Nq;

  // PADS (head dims whose 32-deep contraction steps leave a free 8-slot chunk: d = 40, 80): the softmax bookkeeping rides in the padding
  // of the contraction, as in the forward kernel (attention_x32.hip).  The QUERY side carries -lse as a bf16 pair (hi + lo: 16 mantissa
  // bits) in slots D, D + 1 of Q and -delta in the same slots of dO; the KEY side carries 1.0 there (K and V), and K is pre-scaled by
  // scale * log2(e).  The two MFMA chains then deliver  s * c - lse  and  dP - delta  directly: per score one v_exp_f32, one multiply and
  // the packs are left on the VALU (it was fma, exp, subtract, multiply, packs -- and VALU issue, not the matrix pipe, paces d = 40).
  // Rows beyond the streamed range: slot D + 2 is 1.0 on such a key row and -30000 on every query column (P = exp2(-30000) = 0); a query
  // row beyond Nq carries -30000 in place of -lse.
  constexpr bool PADS = KS * 32 - D >= 8 && D % 8 == 0;
  // X32 (d = 40): S and dP on v_mfma_f32_32x32x16_bf16 -- the contraction padded to 48 instead of 64 (three 16-deep steps: a quarter of
  // the S / dP matrix-pipe cycles), a lane owns ONE column (lane & 31) and 16 of the 32 rows of a chunk.  P and dS then leave the C
  // layout of the 32x32 tile for the B layout of the 16x16x32 accumulation MFMAs through four v_permlane16_swap per matrix: the pairs
  // of rows (8 i + 4 hi + j, i = 0..1) and (i = 2..3) of the two column halves trade 16-lane rows, after which lane group g holds
  // rows {0-3, 8-11} + {0, 16, 4, 20}[g] of its column -- the order the transposed fragment reads of X^T follow.
  constexpr bool X32 = X32W && PADS && D == 40 && NT == 2;
  constexpr int KS32 = (D + 8 + 15) / 16;
  constexpr float MASKV = -30000.0f;
  const float c = a.scale * 1.44269504088896340736f;
  auto split2 = [](float v) {                       // v ~ hi + lo, both bf16
    const float hi = bf2f((bf16_t)(pack2bf(v, 0.f) & 0xffffu));
    return pack2bf(hi, v - hi);
  };
  auto sw_of = [](int row) { return X32 ? ((row >> 1) & 7) : swz<STR>(row); };      // X32: 32 consecutive rows per fragment read
  // X32: column operands of the 32x32x16 MFMAs -- lane (column lane & 31, half hi = lane >> 5) holds d = 16 ks + 8 hi .. + 7
  bf16x8_t yx1[X32 ? KS32 : 1], yx2[X32 ? KS32 : 1];
  if (X32) {
    const int col = c0 + (lane & 31), hi = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < KS32; ++ks) {
      const int d0 = ks * 16 + hi * 8;
      uint4 v1 = uint4{0, 0, 0, 0}, v2 = uint4{0, 0, 0, 0};
      if (col < ncols && d0 < D) {
        v1 = *(const uint4*)(Y1 + (long)col * ldy1 + d0);
        v2 = *(const uint4*)(Y2 + (long)col * ldy2 + d0);
        float f[8];
        unpack8(v1, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= c;
        v1 = pack8(f);
      } else if (col < ncols && d0 == D) {
        if (KV_SIDE) { v1.x = pack2bf(1.0f, 1.0f); v2.x = v1.x; }
        else { v1.x = split2(-lse[col]); v1.y = pack2bf(MASKV, 0.f); v2.x = split2(-dlt[col]); }
      }
      yx1[ks] = __builtin_bit_cast(bf16x8_t, v1);
      yx2[ks] = __builtin_bit_cast(bf16x8_t, v2);
    }
  }
  bf16x8_t y1[NT][KS], y2[NT][KS];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (X32) break;
    const int col = c0 + nt * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 32 + fg * 8;
      uint4 v1 = uint4{0, 0, 0, 0}, v2 = uint4{0, 0, 0, 0};
      if (col < ncols && d0 < D) {
        v1 = *(const uint4*)(Y1 + (long)col * ldy1 + d0);
        v2 = *(const uint4*)(Y2 + (long)col * ldy2 + d0);
        if (PADS) {                                  // the score scale on the column operand of S (K in the dK / dV pass, Q in the dQ pass)
          float f[8];
          unpack8(v1, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= c;
          v1 = pack8(f);
        }
      } else if (PADS && col < ncols && d0 == D) {
        if (KV_SIDE) { v1.x = pack2bf(1.0f, 1.0f); v2.x = v1.x; }
        else { v1.x = split2(-lse[col]); v1.y = pack2bf(MASKV, 0.f); v2.x = split2(-dlt[col]); }
      }
      y1[nt][ks] = __builtin_bit_cast(bf16x8_t, v1);
      y2[nt][ks] = __builtin_bit_cast(bf16x8_t, v2);
    }
  }
  // per-column softmax statistics (dQ pass: the lane's own query)
  float col_l[NT], col_d[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = c0 + nt * 16 + fr;
    col_l[nt] = (!PADS && !KV_SIDE && col < ncols) ? lse[col] : INFINITY;
    col_d[nt] = (!PADS && !KV_SIDE && col < ncols) ? dlt[col] : 0.f;
  }

  // staging bookkeeping: chunk i of a tile -> (row, 16-byte slot)
  int s_row[KS], s_slot[KS], s_lds[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int idx = tid + i * 256;
    s_row[i] = idx / (KS * 4);
    s_slot[i] = idx - s_row[i] * (KS * 4);
    s_lds[i] = s_row[i] * STR + ((s_slot[i] ^ sw_of(s_row[i])) << 4);
  }
  uint4 r1[KS], r2[KS];
  float rl = 0.f, rd = 0.f;
  float pl[KS], pd[KS]; bool pin[KS];              // PADS: raw lse / delta of the row of chunk i, row inside the streamed range
  // full_c = std::true_type: the tile lies entirely inside nrows (no bounds predicates)
  auto load_tile = [&](int row0, auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    if (PADS) {
      // Branch-free: every lane loads a chunk (rows beyond the range and the chunks behind the head dim from a clamped address) and
      // store_tile selects between the data, zeros and the bookkeeping chunk (see PADS above).  With the loads under a lane-varying
      // branch and the chunk patched in afterwards, the compiler copied the loaded registers INSIDE the branch: an s_waitcnt right behind
      // the loads, i.e. a synchronous prefetch (1739 -> 2068 us).  The raw statistics are requested by every lane (one address per row)
      // BEFORE the tile chunks: vector loads return in order, so a wait for these never waits for the tile behind them.
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const int r = row0 + s_row[i];
        const bool in = FULL || r < nrows;
        pin[i] = in;
        if (KV_SIDE) { const int rc_ = in ? r : 0; pl[i] = lse[rc_]; pd[i] = dlt[rc_]; }
      }
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const int r = pin[i] ? row0 + s_row[i] : 0;
        const int sl = min(s_slot[i], D / 8 - 1);
        r1[i] = *(const uint4*)(X1 + (long)r * ldx1 + sl * 8);
        r2[i] = *(const uint4*)(X2 + (long)r * ldx2 + sl * 8);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      r1[i] = uint4{0, 0, 0, 0}; r2[i] = uint4{0, 0, 0, 0};
      const int r = row0 + s_row[i];
      if ((FULL || r < nrows) && s_slot[i] * 8 < D) {
        r1[i] = *(const uint4*)(X1 + (long)r * ldx1 + s_slot[i] * 8);
        r2[i] = *(const uint4*)(X2 + (long)r * ldx2 + s_slot[i] * 8);
      }
    }
    if (KV_SIDE && tid < XT) {     // row statistics: +inf lse for rows beyond Nq makes their P exactly 0
      const int r = row0 + tid;
      rl = (FULL || r < nrows) ? lse[r] : INFINITY;
      rd = (FULL || r < nrows) ? dlt[r] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* T1 = smem + buf * G::BUF;
    unsigned char* T2 = T1 + G::TILE;
    if (PADS) {
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const bool data = pin[i] && s_slot[i] * 8 < D, pad = s_slot[i] * 8 == D;
        uint4 w1 = r1[i], w2 = r2[i];
        if (!data) { w1 = uint4{0, 0, 0, 0}; w2 = uint4{0, 0, 0, 0}; }
        unsigned a1, b1 = 0u, a2;
        if (KV_SIDE) {                               // query rows: -lse, -delta
          a1 = pin[i] ? split2(-pl[i]) : pack2bf(MASKV, 0.f);
          a2 = pin[i] ? split2(-pd[i]) : 0u;
        } else {                                     // key rows: ones, and the mask slot on rows beyond Nk
          a1 = pin[i] ? pack2bf(1.0f, 1.0f) : 0u;
          b1 = pin[i] ? 0u : pack2bf(1.0f, 0.f);
          a2 = a1;
        }
        if (pad) { w1.x = a1; w1.y = b1; w2.x = a2; }
        *(uint4*)(T1 + s_lds[i]) = w1; *(uint4*)(T2 + s_lds[i]) = w2;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) { *(uint4*)(T1 + s_lds[i]) = r1[i]; *(uint4*)(T2 + s_lds[i]) = r2[i]; }
    if (KV_SIDE && tid < XT) {
      float* st = (float*)(T2 + G::TILE);
      st[tid] = rl; st[XT + tid] = rd;
    }
  };

  f32x4_t acc1[NT][DF], acc2[KV_SIDE ? NT : 1][KV_SIDE ? DF : 1];   // acc1: dQ^T | dK^T ; acc2: dV^T
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int f = 0; f < DF; ++f) {
      acc1[nt][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (KV_SIDE) acc2[nt][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

  const int ntiles = (nrows + XT - 1) / XT;
  load_tile(0, std::false_type{});
  store_tile(0);
  __syncthreads();
  // fast_c = std::true_type: this row tile and the next lie entirely inside nrows -- an instantiation without bounds
  // predicates or tail masking (see attention.hip: left in one body they are hoisted into every iteration)
  auto tile = [&](int t, auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    const int row0 = t * XT;
    const bool more = FAST || t + 1 < ntiles;
    if (FAST) load_tile(row0 + XT, std::true_type{});
    else if (more) load_tile(row0 + XT, std::false_type{});
    if (PADS) __builtin_amdgcn_sched_barrier(0);     // straight-line loads: without the fence the scheduler sinks them behind the MFMAs they are meant to hide under
    const unsigned char* T1 = smem + (t & 1) * G::BUF;
    const unsigned char* T2 = T1 + G::TILE;
    const float* st = (const float*)(T2 + G::TILE);

    bf16x8_t pf[NT][2], dsf[NT][2];
    if (X32) {
      typedef __attribute__((ext_vector_type(16))) float f32x16_t;
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
      const int hi = lane >> 5;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const int row = ch * 32 + (lane & 31);
        const int sw = (row >> 1) & 7;
        f32x16_t tS, tU;
#pragma unroll
        for (int e = 0; e < 16; ++e) { tS[e] = 0.f; tU[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
          const bf16x8_t x1 = *(const bf16x8_t*)(T1 + row * STR + (((ks * 2 + hi) ^ sw) << 4));
          const bf16x8_t x2 = *(const bf16x8_t*)(T2 + row * STR + (((ks * 2 + hi) ^ sw) << 4));
          tS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, yx1[ks], tS, 0, 0, 0);
          tU = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, yx2[ks], tU, 0, 0, 0);
        }
        // lane (column lane & 31, half hi) holds rows 8 i + 4 hi + j of the chunk in register 4 i + j
        unsigned pp[8], dd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p0 = __builtin_amdgcn_exp2f(tS[2 * e]), p1 = __builtin_amdgcn_exp2f(tS[2 * e + 1]);
          pp[e] = pack2bf(p0, p1);
          dd[e] = pack2bf(p0 * tU[2 * e], p1 * tU[2 * e + 1]);
        }
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        u32x4_t p0v, p1v, d0v, d1v;                      // column half 0 / 1 after the swaps
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const u32x2_t sp = __builtin_amdgcn_permlane16_swap(pp[e], pp[4 + e], false, false);
          const u32x2_t sd = __builtin_amdgcn_permlane16_swap(dd[e], dd[4 + e], false, false);
          p0v[e] = sp[0]; p1v[e] = sp[1]; d0v[e] = sd[0]; d1v[e] = sd[1];
        }
        pf[0][ch] = __builtin_bit_cast(bf16x8_t, p0v); pf[1][ch] = __builtin_bit_cast(bf16x8_t, p1v);
        dsf[0][ch] = __builtin_bit_cast(bf16x8_t, d0v); dsf[1][ch] = __builtin_bit_cast(bf16x8_t, d1v);
      }
    }
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      if (X32) break;
      // two 16-row MFMA tiles cover rows ch*32 + fg*8 + {0..3} and {4..7} for this lane
      f32x4_t tS[NT][2], tU[NT][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = ch * 32 + (fr >> 2) * 8 + u * 4 + (fr & 3);
        const int sw = swz<STR>(row);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { tS[nt][u] = f32x4_t{0.f, 0.f, 0.f, 0.f}; tU[nt][u] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t x1 = *(const bf16x8_t*)(T1 + row * STR + (((ks * 4 + fg) ^ sw) << 4));
          const bf16x8_t x2 = *(const bf16x8_t*)(T2 + row * STR + (((ks * 4 + fg) ^ sw) << 4));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            tS[nt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, y1[nt][ks], tS[nt][u], 0, 0, 0);
            tU[nt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, y2[nt][ks], tU[nt][u], 0, 0, 0);
          }
        }
      }
      // lane (column fr, group fg) holds rows row0 + ch*32 + fg*8 + u*4 + r
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float p[8], ds[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (PADS) {                                         // the MFMAs delivered s * c - lse and dP - delta
              const float pv = __builtin_amdgcn_exp2f(tS[nt][u][r]);
              p[u * 4 + r] = pv;
              ds[u * 4 + r] = pv * tU[nt][u][r];
              continue;
            }
            const int lr = ch * 32 + fg * 8 + u * 4 + r;      // row inside the tile
            float L, Dl;
            if (KV_SIDE) { L = st[lr]; Dl = st[XT + lr]; }
            else { L = (FAST || row0 + lr < nrows) ? col_l[nt] : INFINITY; Dl = col_d[nt]; }
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(tS[nt][u][r], c, -L));
            p[u * 4 + r] = pv;
            ds[u * 4 + r] = pv * (tU[nt][u][r] - Dl);
          }
        uint4 wp, wd;
        wp.x = pack2bf(p[0], p[1]); wp.y = pack2bf(p[2], p[3]); wp.z = pack2bf(p[4], p[5]); wp.w = pack2bf(p[6], p[7]);
        wd.x = pack2bf(ds[0], ds[1]); wd.y = pack2bf(ds[2], ds[3]); wd.z = pack2bf(ds[4], ds[5]); wd.w = pack2bf(ds[6], ds[7]);
        pf[nt][ch] = __builtin_bit_cast(bf16x8_t, wp);
        dsf[nt][ch] = __builtin_bit_cast(bf16x8_t, wd);
      }
    }
    // ---- accumulate  out^T[d][col] += X^T[d][rows] . Z[rows][col]
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const bf16x8_t a1 = X32 ? tr_frag_x32<STR>(T1, ch * 32 + (((fg & 1) << 4) | ((fg >> 1) << 2)), f * 16, fr) : tr_frag<STR>(T1, ch * 32 + fg * 8, f * 16, fr);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc1[nt][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, dsf[nt][ch], acc1[nt][f], 0, 0, 0);
        if (KV_SIDE) {
          const bf16x8_t a2 = X32 ? tr_frag_x32<STR>(T2, ch * 32 + (((fg & 1) << 4) | ((fg >> 1) << 2)), f * 16, fr) : tr_frag<STR>(T2, ch * 32 + fg * 8, f * 16, fr);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc2[nt][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, pf[nt][ch], acc2[nt][f], 0, 0, 0);
        }
      }
    if (more) store_tile((t + 1) & 1);
    __syncthreads();
  };
  const int nfast = nrows / XT - 1;
  int t = 0;
  for (; t < nfast; ++t) tile(t, std::true_type{});
  for (; t < ntiles; ++t) tile(t, std::false_type{});

  // ---- store: lane holds out[col = fr][d = f*16 + fg*4 + r]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = c0 + nt * 16 + fr;
    if (col >= ncols) continue;
    bf16_t* o1 = KV_SIDE ? a.dK + ((long)b * a.Nk + col) * a.lddk + h * D : a.dQ + ((long)b * a.Nq + col) * a.lddq + h * D;
#pragma unroll
    for (int f = 0; f < DF; ++f) {
      const int d0 = f * 16 + fg * 4;
      if (d0 >= D) continue;
      uint2 o;
      o.x = pack2bf(acc1[nt][f][0] * a.scale, acc1[nt][f][1] * a.scale);
      o.y = pack2bf(acc1[nt][f][2] * a.scale, acc1[nt][f][3] * a.scale);
      *(uint2*)(o1 + d0) = o;
      if (KV_SIDE) {
        bf16_t* o2 = a.dV + ((long)b * a.Nk + col) * a.lddv + h * D;
        uint2 v;
        v.x = pack2bf(acc2[nt][f][0], acc2[nt][f][1]);
        v.y = pack2bf(acc2[nt][f][2], acc2[nt][f][3]);
        *(uint2*)(o2 + d0) = v;
      }
    }
  }
}

// delta[b][h][q] = sum_d dO * O.  A block takes DELTA_ROWS token rows: every thread multiplies 8 channels (one 16-byte load per tensor)
// into a partial in LDS, then one thread per (row, head) adds the D / 8 partials of its head.
constexpr int DELTA_ROWS = 32;
__global__ __launch_bounds__(256) void attention_delta_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO, int ld,
                                                              float* __restrict__ delta, int B, int H, int D, int Nq) {
  extern __shared__ float part[];                         // [DELTA_ROWS][H * D / 8]
  const long rows = (long)B * Nq, row0 = (long)blockIdx.x * DELTA_ROWS;
  const int c8 = H * D / 8, d8 = D / 8;
  const int nrow = (int)min((long)DELTA_ROWS, rows - row0);
  for (int i = threadIdx.x; i < nrow * c8; i += 256) {
    const int r = i / c8, c = i - r * c8;
    const long at = (row0 + r) * ld + c * 8;
    float a[8], g[8];
    unpack8(*(const uint4*)(O + at), a);
    unpack8(*(const uint4*)(dO + at), g);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] * g[k];
    part[i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nrow * H; i += 256) {
    const int h = i / nrow, r = i - h * nrow;           // consecutive threads -> consecutive q of one head: coalesced stores
    const float* p = part + r * c8 + h * d8;
    float s = 0.f;
    for (int k = 0; k < d8; ++k) s += p[k];
    const long row = row0 + r;
    const int b = (int)(row / Nq), q = (int)(row - (long)b * Nq);
    delta[((long)b * H + h) * Nq + q] = s;
  }
}

template <int D, bool KV, int NT, bool X32W = false>
int launch_pass(const AttnBwdArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * BwdGeom<D>::BUF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attention_bwd_kernel<D, KV, NT, X32W>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int ncols = KV ? a.Nk : a.Nq;
  dim3 grid((ncols + 4 * NT * 16 - 1) / (4 * NT * 16), a.H, a.B);
  hipLaunchKernelGGL((attention_bwd_kernel<D, KV, NT, X32W>), grid, dim3(256), lds, stream, a);
  return dfh::check_launch("attention_bwd_kernel");
}

template <int D>
int launch_bwd(const AttnBwdArgs& a, hipStream_t stream) {
  dfh::ProfScope ps(dfh::PC_ATTN_BWD, 14.0 * a.B * a.H * (double)a.Nq * a.Nk * D, 2.0 * a.B * a.H * D * (4.0 * a.Nq + 4.0 * a.Nk), stream);
#ifdef DFH_PROBES
  if constexpr (D == 40) {
    // PROBE builds only -- DFH_ATTN_BWD_X32: 1 = both passes with S / dP on 32x32x16 (48-deep instead of 64-deep: a quarter of their
    // matrix-pipe cycles, P / dS re-laid-out with v_permlane16_swap), 2 / 3 = only the dQ / the dK-dV pass.  Parity-green and SLOWER:
    // 1616 against 1575 us on the 64x64-level launch pair (profiles/r05/attn_bwd_x32_ab.txt) -- the passes are not matrix-pipe bound.
    static const int x32 = [] { const char* e = getenv("DFH_ATTN_BWD_X32"); return e ? atoi(e) : 0; }();
    if (int rc = (x32 == 1 || x32 == 2) ? launch_pass<D, false, 2, true>(a, stream) : launch_pass<D, false, 2>(a, stream)) return rc;
    return (x32 == 1 || x32 == 3) ? launch_pass<D, true, 2, true>(a, stream) : launch_pass<D, true, 2>(a, stream);
  }
#endif
  if (int rc = launch_pass<D, false, 2>(a, stream)) return rc;
  return launch_pass<D, true, (D > 80 ? 1 : 2)>(a, stream);
}

}  // namespace

namespace dfh {

int attention_delta_launch(const bf16_t* O, const bf16_t* dO, int ld, float* delta, int B, int H, int D, int Nq, hipStream_t stream) {
  const long rows = (long)B * Nq;
  DFH_REQUIRE(D % 8 == 0 && ld % 8 == 0 && (size_t)DELTA_ROWS * H * D / 8 * 4 <= 64 * 1024, "attention delta: head dim / row stride must be multiples of 8");
  hipLaunchKernelGGL(attention_delta_kernel, dim3((unsigned)((rows + DELTA_ROWS - 1) / DELTA_ROWS)), dim3(256), DELTA_ROWS * H * D / 8 * 4, stream,
                     O, dO, ld, delta, B, H, D, Nq);
  return check_launch("attention_delta_kernel");
}

int attention_bwd_launch(const AttnBwdArgs& a, hipStream_t stream) {
  DFH_REQUIRE(a.Nq > 0 && a.Nk > 0 && a.B > 0 && a.H > 0, "empty attention");
  DFH_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0, "leading dims must be 16-byte aligned");
  DFH_REQUIRE(a.lddq % 4 == 0 && a.lddk % 4 == 0 && a.lddv % 4 == 0, "gradient leading dims must be 8-byte aligned");
  DFH_REQUIRE(a.lse && a.delta && a.dQ && a.dK && a.dV, "null pointer");
  switch (a.D) {
    case 32: return launch_bwd<32>(a, stream);
    case 40: return launch_bwd<40>(a, stream);
    case 64: return launch_bwd<64>(a, stream);
    case 80: return launch_bwd<80>(a, stream);
    case 128: return launch_bwd<128>(a, stream);
    case 160: return launch_bwd<160>(a, stream);
    default: break;
  }
  set_error("attention_bwd_launch: unsupported head dim " + std::to_string(a.D));
  return -1;
}

}  // namespace dfh
