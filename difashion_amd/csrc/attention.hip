// Fused softmax(Q K^T / sqrt(d)) V for the U-Net's self- and cross-attention on gfx950.
// Reference call sites: diffusers Attention (attn1 / attn2 of BasicTransformerBlock) reached via
// DiFashion/models/difashion.py:249-253,518-523; the reference runs it through xformers
// memory_efficient_attention (difashion.py:118).  SURVEY.md A.3: heads split C, scale d^-0.5,
// (N, d) = (4096,40) (1024,80) (256,160) (64,160) self; Nk = 77 cross.
//
// Structure (DESIGN.md "Kernels/attention"), all 16x16x32 bf16 MFMA, 64-wide waves:
//   * workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries (two 16-wide
//     column tiles) so every K / V^T fragment read from LDS feeds two MFMAs.
//   * scores are computed TRANSPOSED, S^T = K . Q^T: a lane then holds, for ONE query (lane&15),
//     4 keys per 16-key tile -> running max / sum are lane-local plus two xor-shuffles (16, 32).
//   * the K rows of each 16-key tile are chosen as key = (rho>>2)*8 + u*4 + (rho&3), so after two
//     tiles a lane holds 8 CONSECUTIVE keys = exactly the B fragment of the P.V MFMA: the
//     probabilities never leave registers (no LDS round trip, no permutes).
//   * O is accumulated transposed, O^T = V^T . P^T, with V^T read from a [d][key] tile; V^T is
//     produced directly by the V projection GEMM's transposed epilogue (gemm.hip OUT_BF16_T), so
//     nothing is transposed here.  The accumulator column is the lane's own query -> the softmax
//     rescale and the final 1/l need no cross-lane traffic either; output is 8-byte bf16x4 stores.
//   * K/V^T tiles are double-buffered in LDS: the next tile's global loads are issued into
//     registers before the current tile's MFMAs and written to the other buffer afterwards
//     (issue-early / write-late), so HBM/L2 latency hides under compute; one barrier per tile.
//   * deferred max: the running max (and the O / l rescale that costs an AGPR round trip) is only
//     updated when some query of the wave sees its tile max exceed the running max by more than
//     RESCALE_THR (wave-uniform branch); probabilities then stay below 2^RESCALE_THR.
//   * LDS tiles are XOR-swizzled on the 16-byte slot so ds_read_b128 fragment reads are
//     conflict-free; the N x N score matrix never touches HBM.
#include "dfh_common.h"
#include "attention.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int KV_TILE = 64;
constexpr float RESCALE_THR = 6.0f;   // in log2 units of the scaled scores: p <= 2^6 between rescales

template <int D> struct AttnGeom {
  static constexpr int KS = (D + 31) / 32;                       // 32-deep k-steps over the head dim
  static constexpr int DF = (D + 15) / 16;                       // 16-row fragments of O^T
  static constexpr int KSTR = D <= 64 ? 128 : (D <= 128 ? 256 : 512);  // K tile row stride (bytes)
  static constexpr int K_BYTES = KV_TILE * KSTR;
  static constexpr int V_BYTES = DF * 16 * 128;
  static constexpr int NV = (DF * 128 + 255) / 256;              // V^T staging chunks per thread
  static constexpr int BUF = K_BYTES + V_BYTES;
};

// rho = position of a key inside its 16-row MFMA tile (see header); swizzle bits derive from it
DFH_DEVICE int key_rho(int key) { return (((key >> 3) & 3) << 2) | (key & 3); }

template <int D>
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
  using G = AttnGeom<D>;
  constexpr int KS = G::KS, DF = G::DF, KSTR = G::KSTR, NV = G::NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;

  const bf16_t* Qb = a.Q + (long)b * a.Nq * a.ldq + h * D;
  const bf16_t* Kb = a.K + (long)b * a.Nk * a.ldk + h * D;
  const bf16_t* Vb = a.Vt + (long)b * (a.vt_bstride ? a.vt_bstride : (long)a.H * D * a.ldvt) + (long)h * D * a.ldvt;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q = fr][d = ks*32 + fg*8 ..+8]
  bf16x8_t qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + qt * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 32 + fg * 8;
      uint4 v = uint4{0, 0, 0, 0};
      if (q < a.Nq && d0 < D) v = *(const uint4*)(Qb + (long)q * a.ldq + d0);
      qf[qt][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  // ---- staging bookkeeping (fixed per thread): K chunk i -> (key, slot); V^T chunk i -> (row, slot)
  int k_key[KS], k_slot[KS], k_lds[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int idx = tid + i * 256;
    k_key[i] = idx / (KS * 4);
    k_slot[i] = idx - k_key[i] * (KS * 4);
    const int rho = key_rho(k_key[i]);
    const int sw = (KSTR == 128) ? ((rho >> 1) & 7) : rho;
    k_lds[i] = k_key[i] * KSTR + ((k_slot[i] ^ sw) << 4);
  }
  uint4 kreg[KS], vreg[NV];

  // When D is not a multiple of 16 the V^T tile has spare rows: row D holds ONES, so the P.V MFMA
  // also accumulates the softmax denominator l = sum_k p (in O^T row D) -- no per-score VALU add.
  constexpr bool ONES_ROW = (D % 16) != 0;
  // full_c = std::true_type: the tile lies entirely inside Nk -- no per-key bounds predicates, no ragged-tail masking
  auto load_tile = [&](int kv0, auto full_c) {   // global -> registers (zero beyond D / beyond Nk)
    constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      kreg[i] = uint4{0, 0, 0, 0};
      if ((FULL || kv0 + k_key[i] < a.Nk) && k_slot[i] * 8 < D)
        kreg[i] = *(const uint4*)(Kb + (long)(kv0 + k_key[i]) * a.ldk + k_slot[i] * 8);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, slot = idx & 7;
      const int k0 = kv0 + slot * 8;
      uint4 v = uint4{0, 0, 0, 0};
      if (ONES_ROW && row == D) v = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};   // bf16 1.0 x8
      if (idx < DF * 128 && row < D && (FULL || k0 < a.Nk)) {
        v = *(const uint4*)(Vb + (long)row * a.ldvt + k0);
        if (!FULL && k0 + 8 > a.Nk) {   // ragged tail (cross-attention, Nk = 77): zero the padding keys
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
  auto store_tile = [&](int buf) {   // registers -> swizzled LDS image
    unsigned char* Ks = smem + buf * G::BUF;
    unsigned char* Vs = Ks + G::K_BYTES;
#pragma unroll
    for (int i = 0; i < KS; ++i) *(uint4*)(Ks + k_lds[i]) = kreg[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (idx < DF * 128) {
        const int row = idx >> 3, slot = idx & 7;
        *(uint4*)(Vs + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = vreg[i];
      }
    }
  };

  f32x4_t oacc[2][DF];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int f = 0; f < DF; ++f) oacc[qt][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY};   // running max in units of s*c (log2 domain)
  float l_run[2] = {0.f, 0.f};
  const float c = a.scale * 1.44269504088896340736f;   // fold log2(e): p = 2^(s*c - m)

  const int ntiles = (a.Nk + KV_TILE - 1) / KV_TILE;
  load_tile(0, std::false_type{});
  store_tile(0);
  __syncthreads();

  // One key tile.  fast_c = std::true_type: this tile AND the next are entirely inside Nk -- the instantiation carries no
  // ragged-tail code at all (left in one body, hipcc hoists the 40 key-index adds of the masking branch into every
  // iteration: a quarter of the loop's VALU instructions at d = 40, where the loop is VALU-bound).
  auto tile = [&](int t, auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    const int kv0 = t * KV_TILE;
    const bool more = FAST || t + 1 < ntiles;
    if (FAST) load_tile(kv0 + KV_TILE, std::true_type{});      // in flight while this tile computes
    else if (more) load_tile(kv0 + KV_TILE, std::false_type{});
    const unsigned char* Ks = smem + (t & 1) * G::BUF;
    const unsigned char* Vs = Ks + G::K_BYTES;

    // ---- S^T tiles: 4 x (16 keys) for each of the wave's two query tiles
    f32x4_t s[2][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      s[0][tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      s[1][tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int key = (tt >> 1) * 32 + (fr >> 2) * 8 + (tt & 1) * 4 + (fr & 3);
      const int sw = (KSTR == 128) ? ((fr >> 1) & 7) : fr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(Ks + key * KSTR + (((ks * 4 + fg) ^ sw) << 4));
        s[0][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[0][tt], 0, 0, 0);
        s[1][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[1][tt], 0, 0, 0);
      }
    }
    // lane (fr = query, fg) holds for tile tt, reg r: key kv0 + (tt>>1)*32 + fg*8 + (tt&1)*4 + r
    const bool ragged = !FAST && kv0 + KV_TILE > a.Nk;
    bf16x8_t pf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float sc[4][4];      // raw scores; the scale c is folded into the exp2 argument (one FMA per score)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[tt][r] = s[qt][tt][r];
      if (ragged) {   // kernel-uniform: only the last tile of a ragged key range (cross-attention, Nk = 77)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kv0 + (tt >> 1) * 32 + fg * 8 + (tt & 1) * 4 + r;
            if (key >= a.Nk) sc[tt][r] = -INFINITY;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[tt][r]);
      mx = rows_max(mx);          // over the lane's four key groups (lanes l, l^16, l^32, l^48): two VALU row swaps
      mx *= c;             // c > 0: max commutes with the scale
      // deferred max (wave-uniform): rescale only if some query's max grew by more than the threshold
      if (__any(!(mx - m_run[qt] <= RESCALE_THR))) {
        const float m_new = fmaxf(m_run[qt], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        m_run[qt] = m_new;
        l_run[qt] *= alpha;
#pragma unroll
        for (int f = 0; f < DF; ++f) {
          oacc[qt][f][0] *= alpha; oacc[qt][f][1] *= alpha; oacc[qt][f][2] *= alpha; oacc[qt][f][3] *= alpha;
        }
      }
      const float nmr = -m_run[qt];
      float psum = 0.f;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[tt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[tt][r], c, nmr));
          if (!ONES_ROW) psum += sc[tt][r];
        }
      if (!ONES_ROW) l_run[qt] += psum;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        uint4 w;
        w.x = pack2bf(sc[2 * ch][0], sc[2 * ch][1]);
        w.y = pack2bf(sc[2 * ch][2], sc[2 * ch][3]);
        w.z = pack2bf(sc[2 * ch + 1][0], sc[2 * ch + 1][1]);
        w.w = pack2bf(sc[2 * ch + 1][2], sc[2 * ch + 1][3]);
        pf[qt][ch] = __builtin_bit_cast(bf16x8_t, w);
      }
    }
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int f = 0; f < DF; ++f) {
      const int row = f * 16 + fr;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const bf16x8_t vf = *(const bf16x8_t*)(Vs + row * 128 + (((ch * 4 + fg) ^ ((row >> 1) & 7)) << 4));
        oacc[0][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][ch], oacc[0][f], 0, 0, 0);
        oacc[1][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][ch], oacc[1][f], 0, 0, 0);
      }
    }
    if (more) store_tile((t + 1) & 1);   // the other buffer: last read one barrier ago
    __syncthreads();
  };
  const int nfast = a.Nk / KV_TILE - 1;      // tiles whose successor is a full tile too
  int t = 0;
  for (; t < nfast; ++t) tile(t, std::true_type{});
  for (; t < ntiles; ++t) tile(t, std::false_type{});

  // ---- normalise and store: lane holds O[q = fr][d = f*16 + fg*4 + r]
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l;
    if (ONES_ROW) {        // O^T row D (the ones row of V^T) lives in lane group fg = (D%16)/4, register (D%16)%4
      l = __shfl(oacc[qt][D / 16][(D % 16) % 4], ((D % 16) / 4) * 16 + fr, 64);
    } else {
      l = l_run[qt];
      l = rows_sum(l);
    }
    const float inv = attn_qmul(a, b) / l;
    const int q = q0 + qt * 16 + fr;
    if (q >= a.Nq) continue;
    if (a.lse && fg == 0) a.lse[((long)b * a.H + h) * a.Nq + q] = m_run[qt] + __builtin_amdgcn_logf(l);   // v_log_f32 = log2
    const long orow = ((long)b * a.Nq + q) * a.ldo + h * D;
#pragma unroll
    for (int f = 0; f < DF; ++f) {
      const int d0 = f * 16 + fg * 4;
      if (d0 < D) attn_store4(a, orow, d0, oacc[qt][f][0] * inv, oacc[qt][f][1] * inv, oacc[qt][f][2] * inv, oacc[qt][f][3] * inv);
    }
  }
}

template <int D>
int launch(const AttnArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * AttnGeom<D>::BUF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attention_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  dim3 grid((a.Nq + 127) / 128, a.H, a.B);
  dfh::ProfScope ps(dfh::PC_ATTN, 4.0 * a.B * a.H * (double)a.Nq * a.Nk * D,
                    2.0 * a.B * a.H * D * (2.0 * a.Nq + 2.0 * a.Nk), stream);
  hipLaunchKernelGGL(attention_kernel<D>, grid, dim3(256), lds, stream, a);
  return dfh::check_launch("attention_kernel");
}

}  // namespace

namespace dfh {

// attention_x32.hip: the 32x32x16 kernel for the long self-attention launches (d = 40 / 80)
bool attention_x32_eligible(const AttnArgs& a);
int attention_x32_launch(const AttnArgs& a, hipStream_t stream);
// attention_fp8.hip: both products on the e4m3 MFMA (operand factors given, whole 64-key tiles)
bool attention_fp8_eligible(const AttnArgs& a);
int attention_fp8_launch(const AttnArgs& a, hipStream_t stream);

int attention_launch(const AttnArgs& a, hipStream_t stream) {
  DFH_REQUIRE(a.Nq > 0 && a.Nk > 0 && a.B > 0 && a.H > 0, "empty attention");
  DFH_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 4 == 0, "leading dims must be 16-byte aligned");
  DFH_REQUIRE(a.ldvt >= ((a.Nk + 7) / 8) * 8, "V^T rows must be padded to a multiple of 8 keys");
  DFH_REQUIRE(a.O8 ? a.o_amax != nullptr : a.O != nullptr, "attention: no output (bf16 O, or e4m3 O8 with the per-batch maxima of V)");
  if (attention_fp8_eligible(a)) return attention_fp8_launch(a, stream);
  static const bool x32_off = [] { const char* e = getenv("DFH_ATTN_X32"); return e && e[0] == '0'; }();   // A/B switch for the microbenchmarks
  if (!x32_off && attention_x32_eligible(a)) return attention_x32_launch(a, stream);
  census(CK_ATTN_16);
  switch (a.D) {
    case 32: return launch<32>(a, stream);
    case 40: return launch<40>(a, stream);
    case 64: return launch<64>(a, stream);
    case 80: return launch<80>(a, stream);
    case 128: return launch<128>(a, stream);
    case 160: return launch<160>(a, stream);
    default: break;
  }
  set_error("attention_launch: unsupported head dim " + std::to_string(a.D) + " (have 32,40,64,80,128,160)");
  return -1;
}

}  // namespace dfh
