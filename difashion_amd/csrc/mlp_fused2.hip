// The GEGLU feed-forward + proj_out of a C = 320 transformer block as one kernel, SECOND form: eight waves of 16 tokens (two per SIMD).
//
// Same algorithm and the same memory traffic as mlp_fused.hip (read that header first): X in registers as the MFMA B operand, the hidden
// units never leave the wave, [pout . ff2 | pout] as the second GEMM with the raw rows as its last K segment, the weights as a
// fragment-major image staged through registers into a double-buffered LDS ring.  What changes is the shape of a wave.  The first form
// gave a wave 32 tokens on v_mfma_f32_32x32x16_bf16: 80 + 160 accumulator / operand registers, one wave per SIMD -- and measured
// (profiles/r05/mlp_fused_v1_microbench.txt) that ONE wave does not overlap its own MFMAs with its own VALU / LDS / VMEM issue: an
// iteration took the SUM of its 1920 matrix-pipe cycles and its ~3000 cycles of other instructions (257 us per launch, 39 % matrix-pipe
// occupancy).  Here a wave owns 16 tokens on v_mfma_f32_16x16x32_bf16 -- X 40 + D2 80 + D1 2 x 16 registers of ~250, ALL of them VGPRs:
// a function that touches AGPRs gets its 256-register budget split 128 / 128 by the compiler, and 128 VGPRs do not hold the rest -- so
// a 512-thread workgroup puts TWO waves on every SIMD and one's MFMAs run under the other's GEGLU slices, fragment reads and staging.
// The price: a 16-row weight fragment (1 KB from LDS) feeds one 16-cycle MFMA, so at full matrix-pipe rate the four SIMDs would ask
// the LDS for its whole 256 B/clk -- the kernel is LDS-read bound at ~480 KB per iteration, not MFMA bound.
//
// Layouts (v_mfma_f32_16x16x32_bf16; lane = (c = lane & 15, g = lane >> 4)): A row c, k = 8 g .. 8 g + 7 of the 32-deep step; B column c,
// same k; D column c, rows 4 g + r.  Packed W1' rows come as [16 values | 16 gates] per 16 hidden units, so a (value tile, gate tile)
// pair leaves value and gate of hidden units 4 g + r in the same lane; two such pairs = 32 hidden units = one k-step of the second
// GEMM, whose B operand (k-slot 8 g + q) is {units 4 g + q of pair 0, q < 4; units 16 + 4 g + (q - 4) of pair 1}: W2's columns are
// stored in that order (kPerm32).
#include "mlp_fused.h"

#include <cstdlib>
#include <cstring>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef const __attribute__((address_space(1))) u32x4_t* gptr16_t;

constexpr int MC = 320, MHID = 4 * MC;
constexpr int NCHUNK = MHID / 32;            // 40 chunks of 32 hidden units
constexpr int KS = MC / 32;                  // 10 k-steps over C
constexpr int CT = MC / 16;                  // 20 output row tiles
constexpr int W1_BYTES = 4 * KS * 1024, W2_BYTES = CT * 1024, VEC_BYTES = 1024;      // 40 KB + 20 KB + 1 KB per chunk
constexpr int CHUNK_BYTES = W1_BYTES + W2_BYTES + VEC_BYTES;
constexpr int G3_SLICES = 5, G3_BYTES = CT * 2 * 1024;                               // h2 segment: five slices of [20 row tiles][2 k-steps]
constexpr long G3_OFF = (long)NCHUNK * CHUNK_BYTES;
constexpr long IMG_BYTES = G3_OFF + (long)G3_SLICES * G3_BYTES;
static_assert(W1_BYTES == G3_BYTES, "the h2 slices go through the W1 ring");
static_assert(128 * MC * 2 + 16 * 32 * 2 * 4 <= 2 * W1_BYTES + 2 * W2_BYTES, "output tile + statistics scratch fit the dead weight ring");
constexpr int NWAVE = 8;
constexpr int LDS_W1 = 0, LDS_W2 = 2 * W1_BYTES, LDS_VEC = LDS_W2 + 2 * W2_BYTES, LDS_DUMP = LDS_VEC + 2 * VEC_BYTES, LDS_TOTAL = LDS_DUMP + NWAVE * 1024;

// ---------------------------------------------------------------------------------------------------------------- weight image
__global__ __launch_bounds__(256) void mlp2_pack_kernel(const bf16_t* __restrict__ w1, const float* __restrict__ s1, const float* __restrict__ b1,
                                                        const bf16_t* __restrict__ w2p, unsigned char* __restrict__ img) {
  const long slot = (long)blockIdx.x * 256 + threadIdx.x;
  const long byte = slot * 16;
  if (byte >= IMG_BYTES) return;
  uint4 v = uint4{0u, 0u, 0u, 0u};
  if (byte < G3_OFF) {
    const int c = (int)(byte / CHUNK_BYTES), off = (int)(byte - (long)c * CHUNK_BYTES);
    if (off < W1_BYTES) {
      const int blk = off >> 10, lane = (off & 1023) >> 4;
      const int t = blk / KS, ks = blk - t * KS;               // tile t: packed rows 64 c + 16 t .. + 15 (v0, g0, v1, g1)
      const int row = c * 64 + t * 16 + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
      v = *(const uint4*)(w1 + (long)row * MC + k0);
    } else if (off < W1_BYTES + W2_BYTES) {
      const int o2 = off - W1_BYTES, ct = o2 >> 10, lane = (o2 & 1023) >> 4;
      const int n = 16 * ct + (lane & 15), g = lane >> 4;
      const bf16_t* src = w2p + (long)n * (5 * MC) + c * 32;
      const uint2 lo = *(const uint2*)(src + 4 * g), hi = *(const uint2*)(src + 16 + 4 * g);      // units 4 g .. + 3 | 16 + 4 g .. + 3
      v = uint4{lo.x, lo.y, hi.x, hi.y};
    } else {
      const int p = (off - W1_BYTES - W2_BYTES) >> 4;          // row pair: (s_r, s_r+1, b_r, b_r+1) of packed rows 64 c + 2 p, + 1
      if (p < 32) {
        const int r = c * 64 + 2 * p;
        v.x = __float_as_uint(s1[r]); v.y = __float_as_uint(s1[r + 1]); v.z = __float_as_uint(b1[r]); v.w = __float_as_uint(b1[r + 1]);
      }
    }
  } else {
    const int o3 = (int)(byte - G3_OFF), q = o3 / G3_BYTES, o4 = o3 - q * G3_BYTES;
    const int blk = o4 >> 10, lane = (o4 & 1023) >> 4;
    const int ct = blk >> 1, kk = blk & 1;
    const int n = 16 * ct + (lane & 15), k0 = 32 * (2 * q + kk) + 8 * (lane >> 4);
    v = *(const uint4*)(w2p + (long)n * (5 * MC) + 4 * MC + k0);
  }
  *(uint4*)(img + byte) = v;
}

// ---------------------------------------------------------------------------------------------------------------- the kernel
DFH_DEVICE void fence() { __builtin_amdgcn_sched_barrier(0); }

struct Mlp2State {
  f32x4_t d1[2][4];           // [chunk parity][tile v0, g0, v1, g1]: first-GEMM accumulators (VGPRs: the GEGLU reads them)
  f32x4_t d2[CT];             // output accumulators (AGPRs)
  bf16x8_t xf[KS];            // X fragments (AGPRs)
  u32x4_t hreg;               // B operand of the second GEMM: the gated 32 hidden units of the previous chunk
  float rstd, ms;
  GeluK gk;
  f32x2_t vv, gg, ax, rl, pp;
  float4 cv, cg;
};

// pair pr (0..3) of a chunk = (unit block b = pr >> 1, row pair rp = pr & 1): value accumulators d1[2 b][2 rp, + 1], gates d1[2 b + 1][..]
template <int PP>
DFH_DEVICE void pair_consts2(Mlp2State& st, unsigned vb, int pr) {
  const int b = pr >> 1, rp = pr & 1;
  typedef const __attribute__((address_space(3))) f32x4_t* lds_f4;
  const unsigned p = vb + (unsigned)(LDS_VEC - 65536 + PP * VEC_BYTES + (b * 16 + rp) * 16);      // packed rows 32 b + 4 g + 2 rp, + 1
  const f32x4_t cv = *(lds_f4)(uintptr_t)p, cg = *(lds_f4)(uintptr_t)(p + 128u);                   // gate rows: + 16 rows = + 8 pairs
  st.cv = float4{cv[0], cv[1], cv[2], cv[3]};
  st.cg = float4{cg[0], cg[1], cg[2], cg[3]};
}

template <int PP>
DFH_DEVICE void geglu_slice2(Mlp2State& st, unsigned vb, int pr, int k) {
  const int b = pr >> 1, rp = pr & 1;
  if (k == 0) {
    pair_consts2<PP>(st, vb, pr);            // requested here, used one slice later: the partner wave covers the LDS latency
  } else if (k == 1) {
    const f32x2_t r2 = f32x2_t{st.rstd, st.rstd}, m2 = f32x2_t{st.ms, st.ms};
    const f32x2_t fv = __builtin_elementwise_fma(m2, f32x2_t{st.cv.x, st.cv.y}, f32x2_t{st.cv.z, st.cv.w});
    const f32x2_t fg = __builtin_elementwise_fma(m2, f32x2_t{st.cg.x, st.cg.y}, f32x2_t{st.cg.z, st.cg.w});
    st.vv = __builtin_elementwise_fma(r2, f32x2_t{st.d1[PP][2 * b][2 * rp], st.d1[PP][2 * b][2 * rp + 1]}, fv);
    st.gg = __builtin_elementwise_fma(r2, f32x2_t{st.d1[PP][2 * b + 1][2 * rp], st.d1[PP][2 * b + 1][2 * rp + 1]}, fg);
    const float clampv = 5.65685424949f;
    asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(st.ax[0]) : "v"(st.gg[0]), "s"(clampv));
    asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(st.ax[1]) : "v"(st.gg[1]), "s"(clampv));
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(st.rl[0]) : "v"(st.gg[0]));
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(st.rl[1]) : "v"(st.gg[1]));
  } else if (k == 2) {
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(st.pp) : "s"(st.gk.k65), "v"(st.ax));
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(st.pp) : "v"(st.ax), "s"(st.gk.k43));
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(st.pp) : "v"(st.ax), "s"(st.gk.k43));
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(st.pp) : "v"(st.ax), "s"(st.gk.k21));
  } else if (k == 3) {
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(st.pp) : "v"(st.ax), "s"(st.gk.k21));
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(st.pp) : "v"(st.ax), "s"(st.gk.k0));
    st.pp = f32x2_t{__builtin_amdgcn_exp2f(st.pp[0]), __builtin_amdgcn_exp2f(st.pp[1])};
  } else {
    const f32x2_t hh = st.gg * st.pp;
    f32x2_t r;
    asm("v_sub_f32_e64 %0, %1, |%2|" : "=v"(r[0]) : "v"(st.rl[0]), "v"(hh[0]));
    asm("v_sub_f32_e64 %0, %1, |%2|" : "=v"(r[1]) : "v"(st.rl[1]), "v"(hh[1]));
    const f32x2_t o = st.vv * r;
    st.hreg[2 * b + rp] = pack2bf(o[0], o[1]);
  }
}

// One pipeline iteration (see mlp_fused.hip mlp_iter): KIND 0 = first GEMM of a chunk (parity PAR) into st.d1[PAR], KIND 1 = h2 slice Q;
// PREV: GEGLU + second GEMM of the previous chunk; off_w1 / off_w2: image offsets of the 40-piece W1-ring set (-> slot 1 - PAR) and of
// the 21-piece W2 + vector set of the current chunk (-> slot PAR) staged by this iteration, negative = none.
template <int KIND, int PAR, bool PREV, int Q>
DFH_DEVICE void mlp2_iter(Mlp2State& st, const unsigned char* smem, const unsigned char* img, long off_w1, long off_w2, int wave, int lane) {
  constexpr int PP = 1 - PAR;
  constexpr int WIN = 6;                                    // fragment reads in flight
  // two per-lane LDS bases 64 KB apart, both opaque 32-bit LDS addresses: every fragment access is then `ds_read_b128 v, base offset:imm16`.
  // Left to itself the compiler materialised some forty distinct address VGPRs for the offsets beyond 65535 and kept them live across the
  // loop (9 spills).  (The bases must stay LDS-typed: laundered as generic pointers the reads became flat loads -- 424 instead of 239 us.)
  typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
  typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
  unsigned fl_u = (unsigned)(uintptr_t)(lds_cptr)smem + (unsigned)lane * 16u, fh_u = fl_u + 65536u;
  asm volatile("" : "+v"(fl_u), "+v"(fh_u));
  auto lds = [&](int off) -> lds_cptr { return (lds_cptr)(uintptr_t)(off < 65536 ? fl_u + (unsigned)off : fh_u + (unsigned)(off - 65536)); };
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned smem_u = (unsigned)(uintptr_t)(lds_cptr)smem;
  auto lds_dyn = [&](int off) -> lds_ptr { return (lds_ptr)(uintptr_t)(smem_u + (unsigned)off + lane16); };      // wave-dependent offsets (staging stores)
  unsigned vb = smem_u + 65536u + (unsigned)(lane >> 4) * 32u;      // vector-slot base of this lane group, same trick
  asm volatile("" : "+v"(vb));
  auto g1_off = [&](int i) {
    if (KIND == 0) return LDS_W1 + PAR * W1_BYTES + ((i & 3) * KS + (i >> 2)) * 1024;          // (tile i & 3, k-step i >> 2)
    return LDS_W1 + PAR * W1_BYTES + ((i % CT) * 2 + i / CT) * 1024;                            // (row tile i % 20, k-step i / 20 of the slice)
  };
  auto g2_off = [&](int j) { return LDS_W2 + PP * W2_BYTES + j * 1024; };
  // this wave's eight pieces of the iteration: 0..4 = W1-ring pieces wave + 8 k, 5..7 = W2 + vector pieces wave + 8 (k - 5) < 21
  auto piece_off = [&](int k) -> long {
    if (k < 5) return (off_w1 >= 0 ? off_w1 : 0) + (long)(wave + 8 * k) * 1024;
    const int p2 = wave + 8 * (k - 5);
    return (off_w2 >= 0 && p2 < 21) ? off_w2 + (long)p2 * 1024 : 0;
  };
  auto piece_dst = [&](int k) -> int {
    if (k < 5) return off_w1 >= 0 ? LDS_W1 + PP * W1_BYTES + (wave + 8 * k) * 1024 : LDS_DUMP + wave * 1024;
    const int p2 = wave + 8 * (k - 5);
    if (!(off_w2 >= 0 && p2 < 21)) return LDS_DUMP + wave * 1024;
    return p2 < 20 ? LDS_W2 + PAR * W2_BYTES + p2 * 1024 : LDS_VEC + PAR * VEC_BYTES;
  };
  u32x4_t sg[8];
  auto stage_ld = [&](int k) -> u32x4_t {
    const unsigned char* base = img + piece_off(k);
    asm volatile("" : "+s"(base));
    return *(gptr16_t)(base + lane16);
  };
  bf16x8_t fr[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) fr[i] = *(const __attribute__((address_space(3))) bf16x8_t*)lds(g1_off(i));
  fence();
#pragma unroll
  for (int i = 0; i < 40; ++i) {
    if (KIND == 0) {
      const int t = i & 3, ks = i >> 2;
      if (ks == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(st.d1[PAR][t]) : "v"(fr[i % WIN]), "v"(st.xf[ks]));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(st.d1[PAR][t]) : "v"(fr[i % WIN]), "v"(st.xf[ks]));
    } else {
      const int ct = i % CT, kk = i / CT;
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(st.d2[ct]) : "v"(fr[i % WIN]), "v"(st.xf[2 * Q + kk]));
    }
    fence();
    if (i + WIN < 40) fr[i % WIN] = *(const __attribute__((address_space(3))) bf16x8_t*)lds(g1_off(i + WIN));
    else if (PREV) fr[i % WIN] = *(const __attribute__((address_space(3))) bf16x8_t*)lds(g2_off(i + WIN - 40));
    if (PREV && (i & 1) == 0) geglu_slice2<PP>(st, vb, (i >> 1) / 5, (i >> 1) % 5);
    // the wave's eight pieces: ALL requested behind the first sixteen MFMAs (eight 16-byte loads per lane = 64 KB per CU in flight), written
    // to LDS behind the last sixteen.  With two staging registers (16 KB per CU in flight) an iteration took ~6000 cycles whatever it
    // computed -- 61 KB at one L2 round trip (~0.75 us under load) per 16 KB: the stream was latency-bound (profiles/r05/mlp_fused_steps.md)
    if ((i & 1) && i < 16) sg[i >> 1] = stage_ld(i >> 1);
    else if ((i & 1) && i >= 24) *(__attribute__((address_space(3))) u32x4_t*)lds_dyn(piece_dst((i - 24) >> 1)) = sg[(i - 24) >> 1];
    fence();
  }
  if (PREV) {
    asm volatile("s_nop 1" ::: "memory");                  // the gated hidden units are VALU results read by the MFMAs below
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(st.d2[j]) : "v"(fr[(40 + j) % WIN]), "v"(__builtin_bit_cast(bf16x8_t, st.hreg)));
      fence();
      if (j + WIN < CT) fr[(40 + j) % WIN] = *(const __attribute__((address_space(3))) bf16x8_t*)lds(g2_off(j + WIN));
      fence();
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(512, 2) void mlp2_fused_kernel(const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int m = blockIdx.x * 128 + wave * 16 + (lane & 15);          // this lane's token (M % 128 == 0)
  const unsigned char* img = a.img;

  Mlp2State st;
  // the seven GELU coefficients live in SGPR pairs (one scalar source per v_pk_fma_f32): eight VGPRs the two-wave budget does not have
  st.gk.k65 = f32x2_t{1.917432119e-05f, -6.586021110e-04f}; st.gk.k43 = f32x2_t{7.754402186e-03f, -5.296538429e-02f};
  st.gk.k21 = f32x2_t{-4.590602584e-01f, -1.151122051e+00f}; st.gk.k0 = f32x2_t{3.063254510e-07f - 1.0f, 0.0f};
  asm volatile("" : "+s"(st.gk.k65), "+s"(st.gk.k43), "+s"(st.gk.k21), "+s"(st.gk.k0));
#pragma unroll
  for (int k = 0; k < 5; ++k)                                        // W1 of chunk 0 -> W1 slot 0
    *(u32x4_t*)(smem + LDS_W1 + (wave + 8 * k) * 1024 + lane * 16) = *(gptr16_t)(img + (long)(wave + 8 * k) * 1024 + lane * 16);
  {
    const bf16_t* xr = a.x + (long)m * MC + 8 * g;                   // X fragments: token m, k = 32 ks + 8 g .. + 7
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) st.xf[ks] = *(const bf16x8_t*)(xr + 32 * ks);
  }
  {
    GemmArgs gg; gg.ln_stat = a.ln_stat; gg.ln_parts = a.ln_parts; gg.ln_cnt = a.ln_cnt; gg.ln_eps = a.ln_eps; gg.M = a.M;
    const float2 mr = ln_row_stats(gg, m);
    st.rstd = mr.y; st.ms = -mr.x * mr.y;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) st.d2[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  mlp2_iter<0, 0, false, 0>(st, smem, img, (long)CHUNK_BYTES, (long)W1_BYTES, wave, lane);
  for (int c = 1; c < NCHUNK - 1; c += 2) {
    mlp2_iter<0, 1, true, 0>(st, smem, img, (long)(c + 1) * CHUNK_BYTES, (long)c * CHUNK_BYTES + W1_BYTES, wave, lane);
    mlp2_iter<0, 0, true, 0>(st, smem, img, (long)(c + 2) * CHUNK_BYTES, (long)(c + 1) * CHUNK_BYTES + W1_BYTES, wave, lane);
  }
  mlp2_iter<0, 1, true, 0>(st, smem, img, G3_OFF, (long)(NCHUNK - 1) * CHUNK_BYTES + W1_BYTES, wave, lane);
  mlp2_iter<1, 0, true, 0>(st, smem, img, G3_OFF + 1 * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 1, false, 1>(st, smem, img, G3_OFF + 2 * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 0, false, 2>(st, smem, img, G3_OFF + 3 * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 1, false, 3>(st, smem, img, G3_OFF + 4 * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 0, false, 4>(st, smem, img, -1, -1, wave, lane);

  // ---- epilogue: out[m][n] = D2 + bias[n] + resid[m][n], n = 16 ct + 4 g + r.  The wait for the last MFMAs is tied to the accumulator
  //      tuples as operands (their AGPR -> VGPR copies are plain moves the register allocator would otherwise place right behind the
  //      last MFMA of each tuple), and the addresses are derived from an opaque copy of the thread id so that they are computed HERE.
  asm volatile("s_nop 15\n\ts_nop 7"
               : "+v"(st.d2[0]), "+v"(st.d2[1]), "+v"(st.d2[2]), "+v"(st.d2[3]), "+v"(st.d2[4]), "+v"(st.d2[5]), "+v"(st.d2[6]), "+v"(st.d2[7]),
                 "+v"(st.d2[8]), "+v"(st.d2[9]), "+v"(st.d2[10]), "+v"(st.d2[11]), "+v"(st.d2[12]), "+v"(st.d2[13]), "+v"(st.d2[14]), "+v"(st.d2[15]),
                 "+v"(st.d2[16]), "+v"(st.d2[17]), "+v"(st.d2[18]), "+v"(st.d2[19])
               :: "memory");
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int m2 = blockIdx.x * 128 + (tid2 >> 6) * 16 + (tid2 & 15), g2 = (tid2 >> 4) & 3;
  const long row = (long)m2 * MC;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int n = 16 * ct + 4 * g2;
    const float4 b4 = *(const float4*)(a.bias + n);
    const uint2 rr = *(const uint2*)(a.resid + row + n);
    const float v0 = st.d2[ct][0] + b4.x + __uint_as_float(rr.x << 16), v1 = st.d2[ct][1] + b4.y + __uint_as_float(rr.x & 0xffff0000u);
    const float v2 = st.d2[ct][2] + b4.z + __uint_as_float(rr.y << 16), v3 = st.d2[ct][3] + b4.w + __uint_as_float(rr.y & 0xffff0000u);
    uint2 o; o.x = pack2bf(v0, v1); o.y = pack2bf(v2, v3);
    *(uint2*)(a.out + row + n) = o;
    // the rounded tile also goes to LDS (the weight ring is dead: every wave passed the last iteration's barrier) for the statistics below
    if (a.gstat) *(uint2*)(smem + ((tid2 >> 6) * 16 + (tid2 & 15)) * (MC * 2) + n * 2) = o;
  }
  if (a.gstat) {
    // GroupNorm statistics of the output tile for the consuming GroupNorm (unet_model.h: the next resnet's norm1): thread (group j, token
    // slice sl) sums its cpg channels over 8 tokens of the staged tile, the 16 slices are folded in order -- fixed orders, no atomics
    __syncthreads();
    const int cpg = a.gstat_cpg, G = MC / cpg;
    float* red = (float*)(smem + 128 * MC * 2);                       // [16 slices][G][2] behind the tile
    const int j = tid2 % G, sl = tid2 / G;
    if (sl < 16) {
      float ss = 0.f, qq = 0.f;
      for (int t = sl * 8; t < sl * 8 + 8; ++t) {
        const bf16_t* src = (const bf16_t*)(smem + t * (MC * 2)) + j * cpg;
        for (int c = 0; c < cpg; ++c) { const float f = bf2f(src[c]); ss += f; qq = fmaf(f, f, qq); }
      }
      red[(sl * G + j) * 2] = ss; red[(sl * G + j) * 2 + 1] = qq;
    }
    __syncthreads();
    if (tid2 < G) {
      float ss = 0.f, qq = 0.f;
      for (int q = 0; q < 16; ++q) { ss += red[(q * G + tid2) * 2]; qq += red[(q * G + tid2) * 2 + 1]; }
      const int m0 = blockIdx.x * 128, b = m0 / a.gstat_hw, chunk = (m0 - b * a.gstat_hw) / 128, chunks = a.gstat_hw / 128;
      float* dst = a.gstat + (((long)b * G + tid2) * chunks + chunk) * 2;
      dst[0] = ss; dst[1] = qq;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------- token linear
// out = x . W^T (+ bias) (+ resid), K = N = 320: the proj_in / to_out / cross-attention-query projections of a C = 320 transformer block.
// Exactly the h2-segment phase of the kernel above run on its own: a wave holds its 16 tokens' rows as the MFMA B operand (never
// staged through LDS), the 200 KB of weights come as five fragment-major 40-KB slices through the same register-staged ring.  Against the
// tile-per-workgroup GEMM (gemm.hip, 128 x 160 tiles: 184 KB of LDS fill per 128 x 160 outputs, paced by the ~23 B/clk a CU can fill)
// the fill per output halves and the activations bypass LDS altogether.  Epilogue options: a folded-LayerNorm consumer (rstd, -mean rstd
// per token from the producer's records; s and b' per channel), residual, and the per-token LayerNorm statistics of the ROUNDED output
// for the next folded consumer -- a wave owns whole rows, so that is one record per token over all 320 columns, exact two-pass.
__global__ __launch_bounds__(512, 2) void token_linear_kernel(const TokLinArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int m = blockIdx.x * 128 + wave * 16 + (lane & 15);
  const unsigned char* img = a.img;
  Mlp2State st;
#pragma unroll
  for (int k = 0; k < 5; ++k)                                        // slice 0 -> W1 slot 0
    *(u32x4_t*)(smem + LDS_W1 + (wave + 8 * k) * 1024 + lane * 16) = *(gptr16_t)(img + (long)(wave + 8 * k) * 1024 + lane * 16);
  {
    const bf16_t* xr = a.x + (long)m * MC + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) st.xf[ks] = *(const bf16x8_t*)(xr + 32 * ks);
  }
  st.rstd = 1.f; st.ms = 0.f;
  if (a.ln_stat) {
    GemmArgs gg; gg.ln_stat = a.ln_stat; gg.ln_parts = a.ln_parts; gg.ln_cnt = a.ln_cnt; gg.ln_eps = a.ln_eps; gg.M = a.M;
    const float2 mr = ln_row_stats(gg, m);
    st.rstd = mr.y; st.ms = -mr.x * mr.y;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) st.d2[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  mlp2_iter<1, 0, false, 0>(st, smem, img, 1L * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 1, false, 1>(st, smem, img, 2L * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 0, false, 2>(st, smem, img, 3L * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 1, false, 3>(st, smem, img, 4L * G3_BYTES, -1, wave, lane);
  mlp2_iter<1, 0, false, 4>(st, smem, img, -1, -1, wave, lane);
  asm volatile("s_nop 15\n\ts_nop 7"
               : "+v"(st.d2[0]), "+v"(st.d2[1]), "+v"(st.d2[2]), "+v"(st.d2[3]), "+v"(st.d2[4]), "+v"(st.d2[5]), "+v"(st.d2[6]), "+v"(st.d2[7]),
                 "+v"(st.d2[8]), "+v"(st.d2[9]), "+v"(st.d2[10]), "+v"(st.d2[11]), "+v"(st.d2[12]), "+v"(st.d2[13]), "+v"(st.d2[14]), "+v"(st.d2[15]),
                 "+v"(st.d2[16]), "+v"(st.d2[17]), "+v"(st.d2[18]), "+v"(st.d2[19])
               :: "memory");
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int m2 = blockIdx.x * 128 + (tid2 >> 6) * 16 + (tid2 & 15), g2 = (tid2 >> 4) & 3;
  const long row = (long)m2 * MC;
  const bool lnf = a.ln_stat != nullptr, rst = a.rowstat != nullptr;
  float sum = 0.f;
  uint2 ov[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int n = 16 * ct + 4 * g2;
    const float4 b4 = a.bias ? *(const float4*)(a.bias + n) : float4{0.f, 0.f, 0.f, 0.f};
    float v[4] = {st.d2[ct][0], st.d2[ct][1], st.d2[ct][2], st.d2[ct][3]};
    if (lnf) {                                   // rstd * (acc - mean * s) + b'
      const float4 s4 = *(const float4*)(a.ln_s + n);
      v[0] = fmaf(st.rstd, v[0], fmaf(st.ms, s4.x, b4.x)); v[1] = fmaf(st.rstd, v[1], fmaf(st.ms, s4.y, b4.y));
      v[2] = fmaf(st.rstd, v[2], fmaf(st.ms, s4.z, b4.z)); v[3] = fmaf(st.rstd, v[3], fmaf(st.ms, s4.w, b4.w));
    } else { v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
    if (a.resid) {
      const uint2 rr = *(const uint2*)(a.resid + row + n);
      v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
      v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
    }
    uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
    *(uint2*)(a.out + row + n) = o;
    ov[ct] = o;
    if (rst) sum += (__uint_as_float(o.x << 16) + __uint_as_float(o.x & 0xffff0000u)) + (__uint_as_float(o.y << 16) + __uint_as_float(o.y & 0xffff0000u));
  }
  if (rst) {
    // per-token statistics of the rounded row: the four lane groups of a token hold 80 columns each (fixed-order combine)
    const float mean = rows_sum(sum) * (1.0f / MC);
    float m2s = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float d0 = __uint_as_float(ov[ct].x << 16) - mean, d1 = __uint_as_float(ov[ct].x & 0xffff0000u) - mean;
      const float d2 = __uint_as_float(ov[ct].y << 16) - mean, d3 = __uint_as_float(ov[ct].y & 0xffff0000u) - mean;
      m2s = fmaf(d0, d0, m2s); m2s = fmaf(d1, d1, m2s); m2s = fmaf(d2, d2, m2s); m2s = fmaf(d3, d3, m2s);
    }
    m2s = rows_sum(m2s);
    if (g2 == 0) *(float2*)(a.rowstat + (long)m2 * 2) = float2{mean, m2s};
  }
}

// W [320][ldw] bf16 row-major -> five fragment-major slices (blocks (row tile ct, k-step kk) of 1 KB)
__global__ __launch_bounds__(256) void token_linear_pack_kernel(const bf16_t* __restrict__ W, int ldw, unsigned char* __restrict__ img) {
  const long byte = ((long)blockIdx.x * 256 + threadIdx.x) * 16;
  if (byte >= (long)G3_SLICES * G3_BYTES) return;
  const int q = (int)(byte / G3_BYTES), o4 = (int)(byte - (long)q * G3_BYTES);
  const int blk = o4 >> 10, lane = (o4 & 1023) >> 4;
  const int ct = blk >> 1, kk = blk & 1;
  const int n = 16 * ct + (lane & 15), k0 = 32 * (2 * q + kk) + 8 * (lane >> 4);
  *(uint4*)(img + byte) = *(const uint4*)(W + (long)n * ldw + k0);
}

}  // namespace

namespace dfh {

int mlp2_pack_launch(const bf16_t* w1, const float* s1, const float* b1, const bf16_t* w2p, void* img, hipStream_t stream) {
  DFH_REQUIRE(w1 && s1 && b1 && w2p && img, "null argument");
  static_assert(IMG_BYTES == 2703360, "both forms of the fused MLP share one image size (mlp_fused_image_bytes)");
  const long slots = IMG_BYTES / 16;
  hipLaunchKernelGGL(mlp2_pack_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, w1, s1, b1, w2p, (unsigned char*)img);
  return check_launch("mlp2_pack_kernel");
}

int mlp2_fused_launch(const MlpArgs& a, hipStream_t stream) {
  DFH_REQUIRE(a.x && a.resid && a.img && a.ln_stat && a.bias && a.out, "null argument");
  DFH_REQUIRE(a.M > 0 && a.M % 128 == 0, "fused MLP: whole 128-token tiles");
  DFH_REQUIRE(a.ln_parts > 0 && a.ln_cnt > 0 && a.ln_parts * a.ln_cnt == MC, "fused MLP: row statistics of a 320-channel producer");
  if (a.gstat) DFH_REQUIRE(a.gstat_cpg > 0 && MC % a.gstat_cpg == 0 && MC / a.gstat_cpg <= 32 && a.gstat_hw % 128 == 0 && a.M % a.gstat_hw == 0,
                           "fused MLP statistics: groups of channels dividing 320 (at most 32), whole 128-token chunks per image");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)mlp2_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    attr_set = true;
  }
  ProfScope ps(PC_LINEAR, 2.0 * a.M * (8.0 * MC * MC + 5.0 * MC * MC), 3.0 * a.M * MC * 2.0 + 13.0 * MC * MC * 2.0, stream);
  census(CK_MLP_FUSED);
  hipLaunchKernelGGL(mlp2_fused_kernel, dim3(a.M / 128), dim3(512), LDS_TOTAL, stream, a);
  return check_launch("mlp2_fused_kernel");
}


size_t token_linear_image_bytes() { return (size_t)G3_SLICES * G3_BYTES; }
bool token_linear_eligible(int N, int K, long M) { return N == MC && K == MC && M > 0 && M % 128 == 0; }

int token_linear_pack_launch(const bf16_t* W, int ldw, void* img, hipStream_t stream) {
  DFH_REQUIRE(W && img && ldw >= MC && ldw % 8 == 0, "token linear pack: a [320][ldw] bf16 matrix, 16-byte aligned rows");
  const long slots = (long)G3_SLICES * G3_BYTES / 16;
  hipLaunchKernelGGL(token_linear_pack_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, W, ldw, (unsigned char*)img);
  return check_launch("token_linear_pack_kernel");
}

int token_linear_launch(const TokLinArgs& a, hipStream_t stream) {
  DFH_REQUIRE(a.x && a.img && a.out, "null argument");
  DFH_REQUIRE(a.M > 0 && a.M % 128 == 0, "token linear: whole 128-token tiles");
  if (a.ln_stat) DFH_REQUIRE(a.ln_s && a.bias && a.ln_parts > 0 && a.ln_cnt > 0 && a.ln_parts * a.ln_cnt == MC, "token linear: folded LayerNorm needs s, b' and 320-column records");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)token_linear_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    attr_set = true;
  }
  ProfScope ps(PC_LINEAR, 2.0 * a.M * MC * MC, (a.resid ? 3.0 : 2.0) * a.M * MC * 2.0 + 2.0 * MC * MC, stream);
  census(CK_TOKEN_LINEAR);
  hipLaunchKernelGGL(token_linear_kernel, dim3(a.M / 128), dim3(512), LDS_TOTAL, stream, a);
  return check_launch("token_linear_kernel");
}

}  // namespace dfh
