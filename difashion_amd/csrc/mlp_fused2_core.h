// Device machinery shared by the fused feed-forward kernel (mlp_fused2.hip) and the probe kernels built on it (scripts/probes/kernels/
// token_linear.hip): geometry of the weight image and of the LDS ring, the per-wave state, the GEGLU slices and one pipeline iteration.
// Include inside an anonymous namespace of a .hip file.
#pragma once

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

// ---------------------------------------------------------------------------------------------------------------- the kernel
DFH_DEVICE void fence() { __builtin_amdgcn_sched_barrier(0); }

struct Mlp2State {
  f32x4_t d1[2][4];           // [chunk parity][tile v0, g0, v1, g1]: first-GEMM accumulators (VGPRs: the GEGLU reads them)
  f32x4_t d2[CT];             // output accumulators
  bf16x8_t xf[KS];            // X fragments (the MFMA B operand of the first GEMM and of the h2 segment)
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

// One pipeline iteration: KIND 0 = first GEMM of a chunk (parity PAR) into st.d1[PAR], KIND 1 = h2 slice Q;
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

