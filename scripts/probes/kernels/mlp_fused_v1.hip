// PROBE kernel: the FIRST form of the fused feed-forward (four waves of 32 tokens, one per SIMD).  Lost its A/B against the second form
// (difashion_amd/csrc/mlp_fused2.hip): 265 vs 239 us per launch in isolation, equal inside the step (profiles/r05/mlp_fused_forms_*.txt).  Kept as the
// measurement of what ONE wave per SIMD can overlap (nothing: an iteration cost the sum of its MFMA and its other issue cycles).
//
// The GEGLU feed-forward of a transformer block as ONE kernel (C = 320: the 64x64 level of the SD U-Nets):
//
//     out = proj_out( ff.net.2( GEGLU( ff.net.0( LayerNorm3(x) ) ) ) + x ) + resid
//
// Reference call site: DiFashion/models/difashion.py:518-523 -> diffusers BasicTransformerBlock.ff (GEGLU FeedForward) + Transformer2DModel.proj_out.
// As three launches the hidden tensor ([tokens][4 C] bf16: 168 MB at batch 16) makes a round trip through HBM and the short-K GEMMs
// around it run at 650-880 TFLOP/s behind their epilogues (profiles/r04/launch_table.txt: 164 + 93 us per block).  Here a wave owns 32
// tokens for the whole chain and the hidden units never leave its registers:
//
//   * X (raw rows, LayerNorm folded into the weights: lnfold.hip) sits in the AGPR half of the register file as the B operand of
//     v_mfma_f32_32x32x16_bf16 for all 20 k-steps (80 registers), read once from HBM;
//   * per chunk of 32 hidden units: D1[64 packed rows][32 tokens] = W1'[chunk] . X^T (40 MFMAs, two accumulator chains), the GEGLU
//     gate on the accumulators (the packed rows put value and gate of a hidden unit into the same lane), and the rounded product IS
//     the B operand of the second GEMM -- the C layout of a 32x32 tile (rows 8 j + 4 h + r per lane half h) is the B layout of a
//     16-deep k-step once the weight columns are permuted to match (mlp_pack_kernel) -- D2[320][32 tokens] += W2[:, chunk] . H
//     (20 MFMAs, ten accumulator chains, 160 AGPRs);
//   * the [h2 | hidden] K-segment trick of the unfused walk (AttL::fffp: ff.net.2 and proj_out as one matrix) carries over: after the
//     last chunk D2 += Wp . X^T (200 MFMAs on the same X registers), then bias + residual + store.
//
// Four waves of one workgroup (one per SIMD, ~230 VGPRs + 240 AGPRs each) share the weight stream: every chunk's W1' / W2 slices are
// stored in HBM as the LDS image the fragment reads want (1-KB blocks, lane-linear: block b = the 64 x 16 bytes one ds_read_b128 of a
// wave fetches), copied by LDS-DMA one chunk ahead, double-buffered (122 KB).  Nothing else on a SIMD can cover for the wave, so the
// loop is software-pipelined inside the wave, in source order pinned by scheduling barriers: the GEGLU of chunk c - 1 is cut into
// 40 slices of 4-5 VALU instructions, one behind each MFMA of chunk c's first GEMM; the LDS-DMA issue rides behind the first sixteen.
#include "mlp_fused.h"

#include <cstdlib>
#include <cstring>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef const __attribute__((address_space(1))) u32x4_t* gptr16_t;      // 16-byte global loads (explicit address space: no flat loads)

constexpr int MC = 320, MHID = 4 * MC;
constexpr int NCHUNK = MHID / 32;            // 40 chunks of 32 hidden units
constexpr int KS1 = MC / 16;                 // 20 k-steps of the first GEMM / of the h2 segment
constexpr int CT = MC / 32;                  // 10 output row tiles
constexpr int W1_BYTES = 2 * KS1 * 1024, W2_BYTES = 2 * CT * 1024, VEC_BYTES = 1024;
constexpr int CHUNK_BYTES = W1_BYTES + W2_BYTES + VEC_BYTES;        // 62464
constexpr int G3_SLICES = 5, G3_BYTES = CT * 4 * 1024;              // h2 segment: five slices of [10 row tiles][4 k-steps]
constexpr long G3_OFF = (long)NCHUNK * CHUNK_BYTES;
constexpr long IMG_BYTES = G3_OFF + (long)G3_SLICES * G3_BYTES;
static_assert(W1_BYTES == G3_BYTES, "the h2 slices go through the W1 ring");
constexpr int LDS_W1 = 0, LDS_W2 = 2 * W1_BYTES, LDS_VEC = LDS_W2 + 2 * W2_BYTES, LDS_DUMP = LDS_VEC + 2 * VEC_BYTES, LDS_TOTAL = LDS_DUMP + 4 * 1024;      // 129024
// k-position p of a 16-deep k-step of the second GEMM <-> hidden unit (inside its 16-unit tile) the GEGLU leaves there
__device__ __constant__ const int kPerm[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};

// ---------------------------------------------------------------------------------------------------------------- weight image
// One thread per 16-byte slot of the image.  w1: folded-LayerNorm GEGLU projection [8 C][C] (rows packed 16 values | 16 gates),
// s1 / b1: its LayerNorm-fold vectors [8 C]; w2p: [C][5 C] = [pout . ff2 | pout] (unet_model.h AttL::fffp).
__global__ __launch_bounds__(256) void mlp_pack_kernel(const bf16_t* __restrict__ w1, const float* __restrict__ s1, const float* __restrict__ b1,
                                                       const bf16_t* __restrict__ w2p, unsigned char* __restrict__ img) {
  const long slot = (long)blockIdx.x * 256 + threadIdx.x;
  const long byte = slot * 16;
  if (byte >= IMG_BYTES) return;
  uint4 v = uint4{0u, 0u, 0u, 0u};
  if (byte < G3_OFF) {
    const int c = (int)(byte / CHUNK_BYTES), off = (int)(byte - (long)c * CHUNK_BYTES);
    if (off < W1_BYTES) {
      const int blk = off >> 10, lane = (off & 1023) >> 4;
      const int t = blk / KS1, ks = blk - t * KS1;
      const int row = c * 64 + t * 32 + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
      v = *(const uint4*)(w1 + (long)row * MC + k0);
    } else if (off < W1_BYTES + W2_BYTES) {
      const int o2 = off - W1_BYTES, blk = o2 >> 10, lane = (o2 & 1023) >> 4;
      const int t2 = blk / CT, ct = blk - t2 * CT;
      const int n = 32 * ct + (lane & 31), hbase = c * 32 + t2 * 16;
      bf16_t e[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) e[q] = w2p[(long)n * (5 * MC) + hbase + kPerm[8 * (lane >> 5) + q]];
      v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16); v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
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
    const int ct = blk >> 2, kk = blk & 3;
    const int n = 32 * ct + (lane & 31), k0 = 16 * (4 * q + kk) + 8 * (lane >> 5);
    v = *(const uint4*)(w2p + (long)n * (5 * MC) + 4 * MC + k0);
  }
  *(uint4*)(img + byte) = v;
}

// ---------------------------------------------------------------------------------------------------------------- the kernel
DFH_DEVICE void fence() { __builtin_amdgcn_sched_barrier(0); }

struct MlpState {
  f32x16_t d1[2][2];          // [chunk parity][tile]: first-GEMM accumulators (VGPRs: the GEGLU reads them)
  f32x16_t d2[CT];            // output accumulators (AGPRs)
  bf16x8_t xf[KS1];           // X fragments (AGPRs)
  uint32_t hreg[2][4];        // B operand of the second GEMM: the gated hidden units of the previous chunk, two 16-unit tiles
  float rstd, ms;             // LayerNorm statistics of this lane's token: rstd, -mean * rstd
  GeluK gk;
  // GEGLU pair in flight
  f32x2_t vv, gg, ax, rl, pp, ee;
  float4 cv, cg;              // (s_r, s_r+1, b_r, b_r+1) of the pair's value / gate rows
};

// consts of pair pr (0..7) of the chunk whose vectors sit in slot PP: issue the two LDS reads
template <int PP>
DFH_DEVICE void pair_consts(MlpState& st, const unsigned char* smem, int vec_lane, int pr) {
  const int t = pr >> 2, j = (pr >> 1) & 1, rp = pr & 1;
  const unsigned char* p = smem + LDS_VEC + PP * VEC_BYTES + vec_lane + (t * 16 + 4 * j + rp) * 16;
  st.cv = *(const float4*)p;
  st.cg = *(const float4*)(p + 128);       // gate rows: + 16 rows = + 8 pairs
}

// slice k (0..4) of the GEGLU of pair pr of the chunk with parity PP (accumulators st.d1[PP])
template <int PP>
DFH_DEVICE void geglu_slice(MlpState& st, const unsigned char* smem, int vec_lane, int pr, int k) {
  const int t = pr >> 2, j = (pr >> 1) & 1, rp = pr & 1;
  const int iv = 4 * j + 2 * rp, ig = 8 + iv;
  if (k == 0) {
    const f32x2_t r2 = f32x2_t{st.rstd, st.rstd}, m2 = f32x2_t{st.ms, st.ms};
    const f32x2_t fv = __builtin_elementwise_fma(m2, f32x2_t{st.cv.x, st.cv.y}, f32x2_t{st.cv.z, st.cv.w});
    const f32x2_t fg = __builtin_elementwise_fma(m2, f32x2_t{st.cg.x, st.cg.y}, f32x2_t{st.cg.z, st.cg.w});
    st.vv = __builtin_elementwise_fma(r2, f32x2_t{st.d1[PP][t][iv], st.d1[PP][t][iv + 1]}, fv);
    st.gg = __builtin_elementwise_fma(r2, f32x2_t{st.d1[PP][t][ig], st.d1[PP][t][ig + 1]}, fg);
  } else if (k == 1) {
    const float clampv = 5.65685424949f;
    asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(st.ax[0]) : "v"(st.gg[0]), "s"(clampv));
    asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(st.ax[1]) : "v"(st.gg[1]), "s"(clampv));
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(st.rl[0]) : "v"(st.gg[0]));
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(st.rl[1]) : "v"(st.gg[1]));
  } else if (k == 2) {
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(st.pp) : "v"(st.gk.k65), "v"(st.ax));
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(st.pp) : "v"(st.ax), "v"(st.gk.k43));
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(st.pp) : "v"(st.ax), "v"(st.gk.k43));
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(st.pp) : "v"(st.ax), "v"(st.gk.k21));
  } else if (k == 3) {
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(st.pp) : "v"(st.ax), "v"(st.gk.k21));
    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(st.pp) : "v"(st.ax), "v"(st.gk.k0));
    st.ee = f32x2_t{__builtin_amdgcn_exp2f(st.pp[0]), __builtin_amdgcn_exp2f(st.pp[1])};
  } else {
    const f32x2_t hh = st.gg * st.ee;
    f32x2_t r;
    asm("v_sub_f32_e64 %0, %1, |%2|" : "=v"(r[0]) : "v"(st.rl[0]), "v"(hh[0]));
    asm("v_sub_f32_e64 %0, %1, |%2|" : "=v"(r[1]) : "v"(st.rl[1]), "v"(hh[1]));
    const f32x2_t o = st.vv * r;
    st.hreg[t][2 * j + rp] = pack2bf(o[0], o[1]);
    if (pr < 7) pair_consts<PP>(st, smem, vec_lane, pr + 1);
  }
}

// One pipeline iteration.
//   KIND 0: first GEMM of chunk `c` (parity PAR = c & 1) into st.d1[PAR]       KIND 1: h2 slice Q (PAR = Q & 1) into st.d2
//   PREV: the GEGLU and the second GEMM of the previous chunk (parity 1 - PAR) ride along
//   dma_w1 / dma_w2: image offsets of the W1-ring piece set (40 pieces -> slot 1 - PAR) and of the W2 + vector set of the CURRENT chunk
//   (21 pieces -> slot PAR) this iteration requests; negative = nothing to request
template <int KIND, int PAR, bool PREV, int Q>
DFH_DEVICE void mlp_iter(MlpState& st, const unsigned char* smem, const unsigned char* img, long dma_w1, long dma_w2, int wave, int lane) {
  constexpr int PP = 1 - PAR;
  constexpr int WIN = 6;                                    // fragment reads in flight
  const unsigned char* fl = smem + lane * 16;               // this lane's 16 bytes of every fragment block
  const unsigned lane16 = (unsigned)lane * 16u;             // unsigned 32-bit lane offset: scalar base + VGPR offset addressing of the staging loads
  const int vec_lane = (lane >> 5) * 32;
  auto g1_off = [&](int i) {                                // fragment of MFMA i of this iteration's 40
    if (KIND == 0) return LDS_W1 + PAR * W1_BYTES + ((i & 1) * KS1 + (i >> 1)) * 1024;          // (tile i & 1, k-step i >> 1)
    return LDS_W1 + PAR * W1_BYTES + ((i % CT) * 4 + i / CT) * 1024;                             // (row tile i % 10, k-step i / 10 of the slice)
  };
  auto g2_off = [&](int j) { return LDS_W2 + PP * W2_BYTES + j * 1024; };                        // (tile j / 10, row tile j % 10)
  // The weight stream: piece k (0..15) of this wave in this iteration -- pieces 0..9 belong to the W1 ring (image pieces wave + 4 k of the
  // NEXT chunk's first-GEMM slices -> slot 1 - PAR), 10..15 to the current chunk's W2 + vector set (pieces wave + 4 (k - 10) < 21 ->
  // slot PAR).  Staged through REGISTERS (global_load_dwordx4 early, ds_write_b128 sixteen MFMAs later), not by LDS-DMA: a
  // global_load_lds piece holds its issuing wave for ~250 cycles (profiles/DESIGN_LOG: "costs its issuing wave ~250 cycles per KiB"),
  // and with one wave per SIMD nothing else can issue meanwhile -- sixteen pieces per iteration were 4000 of its 5900 cycles (measured
  // with the LDS-DMA form: 277 us per launch; profiles/r05/mlp_fused_steps.txt).
  // (no branches: an iteration with nothing to stage for a piece reads image offset 0 and writes the wave's 1-KB dump region)
  auto piece_off = [&](int k) -> long {
    if (k < 10) return (dma_w1 >= 0 ? dma_w1 : 0) + (long)(wave + 4 * k) * 1024;
    const int p2 = wave + 4 * (k - 10);
    return (dma_w2 >= 0 && p2 < 21) ? dma_w2 + (long)p2 * 1024 : 0;
  };
  auto piece_dst = [&](int k) -> int {
    if (k < 10) return dma_w1 >= 0 ? LDS_W1 + PP * W1_BYTES + (wave + 4 * k) * 1024 : LDS_DUMP + wave * 1024;
    const int p2 = wave + 4 * (k - 10);
    if (!(dma_w2 >= 0 && p2 < 21)) return LDS_DUMP + wave * 1024;
    return p2 < 20 ? LDS_W2 + PAR * W2_BYTES + p2 * 1024 : LDS_VEC + PAR * VEC_BYTES;
  };
  u32x4_t sg[8];
  // staging load of piece k: the piece's image address is wave-uniform -- pinned in SGPRs (opaque to reassociation) so that the load is
  // `global_load_dwordx4 v, v_lane16, s[base]` and no 64-bit per-lane address is built, kept or spilled
  auto stage_ld = [&](int k) -> u32x4_t {
    const unsigned char* base = img + piece_off(k);
    asm volatile("" : "+s"(base));
    return *(gptr16_t)(base + lane16);
  };
  bf16x8_t fr[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) fr[i] = *(const bf16x8_t*)(fl + g1_off(i));
  if (PREV) pair_consts<PP>(st, smem, vec_lane, 0);
  fence();
#pragma unroll
  for (int i = 0; i < 40; ++i) {
    if (KIND == 0) {
      const int t = i & 1, ks = i >> 1;
      if (ks == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(st.d1[PAR][t]) : "v"(fr[i % WIN]), "a"(st.xf[ks]));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(st.d1[PAR][t]) : "v"(fr[i % WIN]), "a"(st.xf[ks]));
    } else {
      const int ct = i % CT, kk = i / CT;
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(st.d2[ct]) : "v"(fr[i % WIN]), "a"(st.xf[4 * Q + kk]));
    }
    fence();
    // refill the window: the rest of this GEMM's fragments, then the first ones of the second GEMM
    if (i + WIN < 40) fr[i % WIN] = *(const bf16x8_t*)(fl + g1_off(i + WIN));
    else if (PREV) fr[i % WIN] = *(const bf16x8_t*)(fl + g2_off(i + WIN - 40));
    if (PREV) geglu_slice<PP>(st, smem, vec_lane, i / 5, i % 5);
    // pieces 0..7: requested behind MFMAs 0..7, written to LDS behind MFMAs 16..23; pieces 8..15: requested there, written behind 32..39
    if (i < 8) sg[i] = stage_ld(i);
    else if (i >= 16 && i < 24) {
      *(u32x4_t*)(const_cast<unsigned char*>(fl) + piece_dst(i - 16)) = sg[i - 16];
      sg[i - 16] = stage_ld(i - 8);
    } else if (i >= 32) *(u32x4_t*)(const_cast<unsigned char*>(fl) + piece_dst(i - 24)) = sg[i - 32];
    fence();
  }
  if (PREV) {
    // the gated hidden units are VALU results: two wait states before an MFMA may read them as its B operand
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      const int t2 = j / CT, ct = j % CT;
      const uint4 hb = uint4{st.hreg[t2][0], st.hreg[t2][1], st.hreg[t2][2], st.hreg[t2][3]};
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(st.d2[ct]) : "v"(fr[(40 + j) % WIN]), "v"(__builtin_bit_cast(bf16x8_t, hb)));
        fence();
      if (j + WIN < 20) fr[(40 + j) % WIN] = *(const bf16x8_t*)(fl + g2_off(j + WIN));
      fence();
    }
  }
  // every piece this wave staged is written, then every wave's: the next iteration reads them
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + ql;          // this lane's token (the launcher guarantees M % 128 == 0)
  const unsigned char* img = a.img;

  MlpState st;
  st.gk = gelu_consts();
  // first pieces of the weight stream: W1 of chunk 0 -> W1 slot 0 (the first iteration requests chunk 1 and W2 / vectors of chunk 0)
#pragma unroll
  for (int k = 0; k < 10; ++k)
    *(u32x4_t*)(smem + LDS_W1 + (wave + 4 * k) * 1024 + lane * 16) = *(gptr16_t)(img + (long)(wave + 4 * k) * 1024 + lane * 16);
  // X fragments: token m, k = 16 ks + 8 h .. + 7
  {
    const bf16_t* xr = a.x + (long)m * MC + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) st.xf[ks] = *(const bf16x8_t*)(xr + 16 * ks);
  }
  // LayerNorm statistics of the token from its producer's per-column-tile records (gemm.h ln_row_stats)
  {
    GemmArgs g; g.ln_stat = a.ln_stat; g.ln_parts = a.ln_parts; g.ln_cnt = a.ln_cnt; g.ln_eps = a.ln_eps; g.M = a.M;
    const float2 mr = ln_row_stats(g, m);
    st.rstd = mr.y; st.ms = -mr.x * mr.y;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) st.d2[ct][r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // chunk 0: first GEMM only
  mlp_iter<0, 0, false, 0>(st, smem, img, (long)CHUNK_BYTES, (long)W1_BYTES, wave, lane);
  // chunks 1 .. 39 in pairs (odd, even): the accumulator buffers alternate at compile time
  for (int c = 1; c < NCHUNK - 1; c += 2) {
    mlp_iter<0, 1, true, 0>(st, smem, img, (long)(c + 1) * CHUNK_BYTES, (long)c * CHUNK_BYTES + W1_BYTES, wave, lane);
    mlp_iter<0, 0, true, 0>(st, smem, img, (long)(c + 2) * CHUNK_BYTES, (long)(c + 1) * CHUNK_BYTES + W1_BYTES, wave, lane);
  }
  // chunk 39 (odd): requests the first h2 slice into W1 slot 0
  mlp_iter<0, 1, true, 0>(st, smem, img, G3_OFF, (long)(NCHUNK - 1) * CHUNK_BYTES + W1_BYTES, wave, lane);
  // h2 slices 0 .. 4 (slot Q & 1); slice 0 carries the GEGLU + second GEMM of chunk 39
  mlp_iter<1, 0, true, 0>(st, smem, img, G3_OFF + 1 * G3_BYTES, -1, wave, lane);
  mlp_iter<1, 1, false, 1>(st, smem, img, G3_OFF + 2 * G3_BYTES, -1, wave, lane);
  mlp_iter<1, 0, false, 2>(st, smem, img, G3_OFF + 3 * G3_BYTES, -1, wave, lane);
  mlp_iter<1, 1, false, 3>(st, smem, img, G3_OFF + 4 * G3_BYTES, -1, wave, lane);
  mlp_iter<1, 0, false, 4>(st, smem, img, -1, -1, wave, lane);

  // ---- epilogue: out[m][n] = D2 + bias[n] + resid[m][n], n = 32 ct + 8 j + 4 h + r
  // (the compiler does not see the XDL writes of the inline-asm MFMAs: 18 wait states before the accumulators are read)
  // ... and the wait is tied to the ten accumulator tuples as operands: their AGPR -> VGPR copies are plain register moves that the
  // register allocator otherwise places right behind the last MFMA of each tuple (the last-written tile came out 15 % wrong)
  asm volatile("s_nop 15\n\ts_nop 7"
               : "+a"(st.d2[0]), "+a"(st.d2[1]), "+a"(st.d2[2]), "+a"(st.d2[3]), "+a"(st.d2[4]), "+a"(st.d2[5]), "+a"(st.d2[6]), "+a"(st.d2[7]),
                 "+a"(st.d2[8]), "+a"(st.d2[9])
               :: "memory");
  // the token index is re-derived through an opaque copy of the thread id: the per-lane output / residual addresses are then computed
  // HERE instead of being kept in six VGPRs across the whole kernel (they were spilled to scratch)
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int m2 = blockIdx.x * 128 + (tid2 >> 6) * 32 + (tid2 & 31), h2 = (tid2 >> 5) & 1;
  const long row = (long)m2 * MC;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = 32 * ct + 8 * j + 4 * h2;
      const float4 b4 = *(const float4*)(a.bias + n);
      const uint2 rr = *(const uint2*)(a.resid + row + n);
      const float v0 = st.d2[ct][4 * j] + b4.x + __uint_as_float(rr.x << 16), v1 = st.d2[ct][4 * j + 1] + b4.y + __uint_as_float(rr.x & 0xffff0000u);
      const float v2 = st.d2[ct][4 * j + 2] + b4.z + __uint_as_float(rr.y << 16), v3 = st.d2[ct][4 * j + 3] + b4.w + __uint_as_float(rr.y & 0xffff0000u);
      uint2 o; o.x = pack2bf(v0, v1); o.y = pack2bf(v2, v3);
      *(uint2*)(a.out + row + n) = o;
    }
}

}  // namespace

namespace dfh {


int mlp_pack_launch(const bf16_t* w1, const float* s1, const float* b1, const bf16_t* w2p, void* img, hipStream_t stream) {
  DFH_REQUIRE(w1 && s1 && b1 && w2p && img, "null argument");
  const long slots = IMG_BYTES / 16;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, w1, s1, b1, w2p, (unsigned char*)img);
  return check_launch("mlp_pack_kernel");
}

int mlp_fused_launch(const MlpArgs& a, hipStream_t stream) {
  DFH_REQUIRE(a.x && a.resid && a.img && a.ln_stat && a.bias && a.out, "null argument");
  DFH_REQUIRE(a.M > 0 && a.M % 128 == 0, "fused MLP: whole 128-token tiles");
  DFH_REQUIRE(a.ln_parts > 0 && a.ln_cnt > 0 && a.ln_parts * a.ln_cnt == MC, "fused MLP: row statistics of a 320-channel producer");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)mlp_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    attr_set = true;
  }
  // algorithmic work: ff.net.0 (2 M 8C C) + [ff.net.2 | proj_out] (2 M C 5C); bytes: x, resid in, out, the weights once
  ProfScope ps(PC_LINEAR, 2.0 * a.M * (8.0 * MC * MC + 5.0 * MC * MC), 3.0 * a.M * MC * 2.0 + 13.0 * MC * MC * 2.0, stream);
  census(CK_MLP_FUSED);
  hipLaunchKernelGGL(mlp_fused_kernel, dim3(a.M / 128), dim3(256), LDS_TOTAL, stream, a);
  return check_launch("mlp_fused_kernel");
}

}  // namespace dfh
