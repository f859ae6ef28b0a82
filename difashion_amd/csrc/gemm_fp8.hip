// fp8 (OCP e4m3fn) linear layers on the block-scaled MFMA of gfx950 -- BASELINE.json configs[4]: "fp8 MFMA attention +
// 1x1-conv path".  Covers the LayerNorm-fed projections of every transformer block (self-attention q | k and v, cross-attention
// q, the GEGLU feed-forward input projection: 60 % of the linear-class FLOPs; reference call sites
// DiFashion/models/difashion.py:249-253,518-523 -> diffusers BasicTransformerBlock attn1 / attn2 / ff.net.0).  The reference runs
// these in fp16 autocast (run_inf4eval.sh:1); fp8 is this project's own target, parity-tested against the fp32 oracle.
//
// Scaling: activations per TOKEN (the LayerNorm kernel that produces them sees the whole row: amax -> scale, one extra float
// per row), weights per OUTPUT CHANNEL (quantised once per weight update from the packed bf16 matrix); the product of the two
// scales is applied to the fp32 accumulator in the epilogue.  The hardware block scales (E8M0 per 32 contraction elements) are
// held at 1.0: v_mfma_scale_f32_32x32x64_f8f6f4 is used for its 2x rate (2048 flop / cycle / SIMD, 4.6 PFLOP/s measured chip
// wide) -- there is no non-scaled fp8 MFMA with K = 64.
//
// Kernel: 4 waves side by side along the pixels (wave = PB x 32 pixels x 160 channels, 5 x PB 32x32 accumulator blocks), 64-byte
// (= 64-element) k-steps, 3-stage LDS ring filled by global_load_lds_dwordx4 (lane-linear image, 16-byte slots XOR-swizzled by
// (row >> 2) & 3 on the SOURCE address and on the fragment read: conflict-free for the 32-row fragments), counted vmcnt + one raw
// barrier per k-step, weights as the MFMA A operand so a lane ends with 4 consecutive output channels of one pixel; epilogue in
// registers (scales, bias, GEGLU on the (value, gate) rows a lane holds of one 32-row block, residual), bf16 tile staged per wave
// through LDS for full-row 16-byte stores.
#include "gemm.h"

#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int KB8 = 64;          // bytes (= fp8 elements) per k-step = one MFMA contraction
constexpr int NST8 = 3;

template <int N> DFH_DEVICE void f8_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

DFH_DEVICE unsigned pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
constexpr float FP8_MAX = 448.0f;

// E8M0 block scale of a group whose largest magnitude is am: the smallest power of two 2^e with am / 2^e <= 448.  Returns the biased
// exponent byte (e + 127, kept inside [1, 253] so that its reciprocal is a normal float) and, through inv, 2^-e.
DFH_DEVICE unsigned e8m0_of(float am, float* inv) {
  const unsigned bits = __float_as_uint(am * (1.0f / FP8_MAX));
  unsigned e = (bits >> 23) + ((bits & 0x7fffffu) ? 1u : 0u);       // ceil(log2(am / 448)) + 127
  e = am > 0.f ? min(max(e, 1u), 253u) : 127u;
  *inv = __uint_as_float((254u - e) << 23);
  return e;
}

// PB x 32 pixel rows per wave (4 waves side by side along the pixels), BN output channels per tile; MX: the activations carry E8M0 block
// scales (one byte per row and 32 contraction elements, Fp8GemmArgs::sx), fed to the MFMA's scale operand
// TOUT: the transposed (V^T) output mode, its own instantiation (the 256 x 160 tile is at the register limit: compiled into the row-major
// kernel it spilled)
template <int PB, int BN, bool MX, bool TOUT>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(const Fp8GemmArgs a) {
  constexpr int BM = 4 * PB * 32;
  constexpr int NCI = BN / 32;                               // 32-channel accumulator blocks per wave
  constexpr int PA = BM / 16, PW = BN / 16;                  // 1-KiB staging pieces (16 rows x 64 B)
  constexpr int IA = PA / 4, IW = (PW + 3) / 4;
  constexpr int A_BYTES = BM * KB8, W_BYTES = BN * KB8;
  constexpr int S_WAVE = PB * 64;                            // MX: per wave, the scale bytes of ITS rows for one k-step: [2 k-halves][PB * 32]
  constexpr int S_BYTES = MX ? 4 * S_WAVE : 0;
  constexpr int STAGE = A_BYTES + W_BYTES + S_BYTES;
  constexpr int N_LO = IA + PW / 4 + (MX ? 1 : 0), N_HI = N_LO + 1, PW_REM = PW % 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, kh = lane >> 5;
  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  int m0_ = m0, n0_ = n0;
  const int nk = a.K / KB8;
  const bool hi_wave = wave < PW_REM;

  // ---- staging: piece p = i * 4 + wave holds tile rows p*16 + lane/4; slot s of row r holds source chunk s ^ ((r >> 2) & 3)
  const int srow = lane >> 2;
  const int schunk = (lane & 3) ^ ((srow >> 2) & 3);
  // 32-bit byte offsets from the two SGPR bases (rows beyond M / N clamped to the last valid row: their accumulator rows / columns are
  // never stored) -- as 64-bit pointers + strides the seven pieces cost 21 VGPRs and the 256 x 160 transposed-output tile spilled
  unsigned lo_a[IA], lo_w[IW];
  const uint8_t* Ab = a.A; const uint8_t* Wb8 = a.W;
  asm volatile("" : "+s"(Ab), "+s"(Wb8));
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int m = min(m0 + (i * 4 + wave) * 16 + srow, a.M - 1);
    lo_a[i] = (unsigned)m * (unsigned)a.lda + (unsigned)schunk * 16u;
  }
#pragma unroll
  for (int i = 0; i < IW; ++i) {
    const int n = min(n0 + (i * 4 + wave) * 16 + srow, a.N - 1);
    lo_w[i] = (unsigned)n * (unsigned)a.K + (unsigned)schunk * 16u;
  }
  // MX scales: lanes 0 .. 16 PB - 1 of every wave fetch 4 bytes each = rows 4 j .. 4 j + 3 of the wave's PB * 32 rows, k-half lane / (8 PB)
  // (sx is [K / 32][M]: the bytes of one k-block are contiguous over the rows); one LDS-DMA instruction per wave and k-step
  const bool s_lane = MX && lane < 16 * PB;
  const uint8_t* lp_s = a.zero; unsigned ls_s = 0;
  if (s_lane) {
    const int khs = lane / (8 * PB), m = m0 + wave * PB * 32 + (lane % (8 * PB)) * 4;
    if (m < a.M) { lp_s = a.sx + ((size_t)khs * a.M + m); ls_s = 2u * (unsigned)a.M; }
  }
  auto glds = [&](const uint8_t* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue = [&](int buf) {
    unsigned char* As = smem + buf * STAGE + wave * 1024;
    unsigned char* Ws = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < IA; ++i) { glds(Ab + lo_a[i], As + i * 4096); lo_a[i] += KB8; }
#pragma unroll
    for (int i = 0; i < IW; ++i) {
      if (i * 4 + wave >= PW) continue;                      // wave-uniform
      glds(Wb8 + lo_w[i], Ws + i * 4096); lo_w[i] += KB8;
    }
    if (MX) {
      if (s_lane)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)lp_s,
                                         (__attribute__((address_space(3))) void*)(smem + buf * STAGE + A_BYTES + W_BYTES + wave * S_WAVE), 4, 0, 0);
      lp_s += ls_s;
    }
  };

  f32x16_t acc[NCI][PB];
#pragma unroll
  for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
    for (int pj = 0; pj < PB; ++pj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ci][pj][r] = 0.f;

  // fragment read offsets inside a stage: row * 64 + (chunk ^ ((row >> 2) & 3)) * 16; (row >> 2) & 3 depends on ql only.  The 64 contraction
  // elements of the f8f6f4 MFMA sit as k = 32 * (register half) + 16 * (lane half) + byte (measured: scripts/probes/mx_scale_probe.hip --
  // registers 0-3 of BOTH lane halves form the first 32-element scale block, whose scale comes from lanes 0-31; registers 4-7 the second,
  // scaled by lanes 32-63): lane half kh reads the 16-byte chunks kh and kh + 2 of its row.  (Without block scales any consistent
  // assignment of chunks gives the same dot product; with them the chunks must be the hardware's.)
  const int sw = (ql >> 2) & 3;
  const int f0 = (kh ^ sw) << 4, f1 = ((kh + 2) ^ sw) << 4;
  const int x_row = (wave * PB * 32 + ql) * KB8;
  const int w_row = A_BYTES + ql * KB8;
  const int s_off = A_BYTES + W_BYTES + wave * S_WAVE + kh * (PB * 32) + ql;
  const int unit_scale = 0x7f7f7f7f;                         // E8M0 127 = 2^0 in every block-scale slot

  if (nk > 0) {
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NST8 - 1; ++s)
      if (s < nk) { issue(s); ++issued; }
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
      const int ahead = issued - 1 - t;
      if (ahead == 0) f8_vmcnt<0>();
      else if (ahead == 1) { if (hi_wave) f8_vmcnt<N_HI>(); else f8_vmcnt<N_LO>(); }
      else { if (hi_wave) f8_vmcnt<2 * N_HI>(); else f8_vmcnt<2 * N_LO>(); }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (issued < nk) {
        int nb = buf - 1; if (nb < 0) nb += NST8;
        issue(nb);
        ++issued;
      }
      const unsigned char* S = smem + buf * STAGE;
      i32x8_t xf[PB];
      int xs[PB];
#pragma unroll
      for (int pj = 0; pj < PB; ++pj) {
        const uint4 lo = *(const uint4*)(S + x_row + pj * 32 * KB8 + f0), hi4 = *(const uint4*)(S + x_row + pj * 32 * KB8 + f1);
        xf[pj] = i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi4.x, (int)hi4.y, (int)hi4.z, (int)hi4.w};
        xs[pj] = MX ? (int)S[s_off + pj * 32] : unit_scale;  // the E8M0 scale of this lane's 32 contraction elements of its row
      }
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci) {
        const uint4 lo = *(const uint4*)(S + w_row + ci * 32 * KB8 + f0), hi4 = *(const uint4*)(S + w_row + ci * 32 * KB8 + f1);
        const i32x8_t wf = i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi4.x, (int)hi4.y, (int)hi4.z, (int)hi4.w};
#pragma unroll
        for (int pj = 0; pj < PB; ++pj)
          acc[ci][pj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf, xf[pj], acc[ci][pj], 0, 0, 0, unit_scale, 0, xs[pj]);
      }
      if (++buf == NST8) buf = 0;
    }
  }

  // ---------------------------------------------------------------- epilogue
  // lane (pixel ql, half kh) holds of block (ci, pj): channels ci*32 + 8 (r >> 2) + 4 kh + (r & 3), r = 0..15
  __syncthreads();                                           // pipeline buffers free: per-wave staging regions below
  // the tile origin is re-read through an opaque copy: everything the epilogue derives from it (row / column indices, output addresses,
  // the per-row activation scale) is then computed HERE -- hoisted above the k-loop those values stayed live across it beside the 160
  // accumulator registers and were spilled (13 VGPRs of the transposed-output tile)
  asm volatile("" : "+s"(m0_), "+s"(n0_));
  const bool geglu = a.act == ACT_GEGLU;
  const bool out8 = a.out_mode == OUT_FP8_MX;
  constexpr int RS = BN * 2 + 16;                            // byte stride of a wave's staged 32-row block (bf16 rows; e4m3 rows use half of it)
  unsigned char* stage = smem + wave * (32 * RS);
  // per-channel weight scales and bias of the tile's columns go to LDS once: read per fragment behind `if (n < a.N)` they
  // were 10-20 dependent L2 round trips per 32-row block (same finding as the bf16 kernel's epilogue, gemm.hip)
  static_assert(4 * 32 * RS + 2 * BN * 4 <= NST8 * STAGE, "staging regions + scale / bias slices must fit the pipeline buffers");
  float* swl = (float*)(smem + 4 * 32 * RS);
  float* bl = swl + BN;
  for (int c = tid; c < BN; c += 256) {
    const int n = n0_ + c;
    swl[c] = n < a.N ? a.sW[n] : 0.f;
    bl[c] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int pj = 0; pj < PB; ++pj) {
    const int m = m0_ + wave * PB * 32 + pj * 32 + ql;
    const bool m_ok = m < a.M;
    const float sa = m_ok ? (a.sA ? a.sA[m / a.sa_div] : 1.0f) * a.sa_mul : 0.f;
    if constexpr (TOUT) {                                    // attention V^T: out[b][n][mm]
      float am = 0.f;
      const int b = m_ok ? m / a.rows_per_b : 0, mm = m - b * a.rows_per_b;
      const int m_first = m0_ + wave * PB * 32 + pj * 32;
      // whole 32-row blocks inside one image: the block is staged TRANSPOSED through the wave's LDS region and leaves as 16-byte
      // pieces along the pixel axis (10 store instructions per lane instead of 80 two-byte ones)
      if (a.rows_per_b % 32 == 0 && m_first + 32 <= a.M && a.ld_out % 8 == 0 && n0_ + BN <= a.N) {
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {                      // four consecutive channels per (block, g): one 16-byte read of the scale / bias slices
            const int c = ci * 32 + 8 * g + 4 * kh;
            const float4 sw4 = *(const float4*)(swl + c), b4 = *(const float4*)(bl + c);
            const float sw[4] = {sw4.x, sw4.y, sw4.z, sw4.w}, bb4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bf16_t o = f2bf(fmaf(acc[ci][pj][4 * g + j] * sa, sw[j], bb4[j]));      // bl is zero-filled without a bias
              am = fmaxf(am, fabsf(bf2f(o)));
              *(bf16_t*)(stage + (c + j) * 64 + ql * 2) = o;  // [BN columns][32 pixels]
            }
          }
          asm volatile("" ::: "memory");                      // one block's reads and stores at a time: no 80-deep hoist of the slice reads
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int bb = m_first / a.rows_per_b, mm0 = m_first - bb * a.rows_per_b;
        for (int q = lane; q < BN * 4; q += 64) {
          const int c = q >> 2, part = q & 3;
          *(uint4*)((bf16_t*)a.out + ((long)bb * a.N + n0_ + c) * a.ld_out + mm0 + part * 8) = *(const uint4*)(stage + c * 64 + part * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else
      if (m_ok) {
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = n0_ + ci * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
            if (n < a.N) {
              float v = acc[ci][pj][r] * sa * swl[n - n0_];
              if (a.bias) v += bl[n - n0_];
              const bf16_t o = f2bf(v);
              am = fmaxf(am, fabsf(bf2f(o)));
              ((bf16_t*)a.out)[((long)b * a.N + n) * a.ld_out + mm] = o;
            }
          }
      }
      if (a.amax) {
        // largest |V| per batch element (the bound of the attention output that is quantised against it): one atomic per wave when the
        // wave's 32 rows lie in one image, else one per lane.  Non-negative floats order like their bit patterns.
        if (a.rows_per_b % 32 == 0) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
          if (lane == 0 && m_first < a.M) atomicMax((unsigned*)a.amax + m_first / a.rows_per_b, __float_as_uint(am));
        } else if (m_ok) {
          atomicMax((unsigned*)a.amax + b, __float_as_uint(am));
        }
      }
      continue;
    } else {
    if (geglu) {
      // rows 0..15 of a 32-row block are values, 16..31 the gates of the same 16 hidden units (packed in 16-row blocks): lane (ql, kh)
      // ends with hidden units ci*16 + 8 g + 4 kh + j (g = 0, 1; j = 0..3) of pixel ql
      float hv[NCI][8];
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int cv = ci * 32 + 8 * g + 4 * kh;           // packed column of the value; its gate is 16 columns further
          const float4 swv = *(const float4*)(swl + cv), swg = *(const float4*)(swl + cv + 16);
          const float4 bv = *(const float4*)(bl + cv), bg = *(const float4*)(bl + cv + 16);
          const float v0 = acc[ci][pj][4 * g] * sa * swv.x + bv.x, v1 = acc[ci][pj][4 * g + 1] * sa * swv.y + bv.y;
          const float v2 = acc[ci][pj][4 * g + 2] * sa * swv.z + bv.z, v3 = acc[ci][pj][4 * g + 3] * sa * swv.w + bv.w;
          const float g0 = acc[ci][pj][8 + 4 * g] * sa * swg.x + bg.x, g1 = acc[ci][pj][9 + 4 * g] * sa * swg.y + bg.y;
          const float g2 = acc[ci][pj][10 + 4 * g] * sa * swg.z + bg.z, g3 = acc[ci][pj][11 + 4 * g] * sa * swg.w + bg.w;
          hv[ci][4 * g] = v0 * gelu_erf_f(g0); hv[ci][4 * g + 1] = v1 * gelu_erf_f(g1);
          hv[ci][4 * g + 2] = v2 * gelu_erf_f(g2); hv[ci][4 * g + 3] = v3 * gelu_erf_f(g3);
        }
      if (!out8) {
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            uint2 o;
            o.x = pack2bf(hv[ci][4 * g], hv[ci][4 * g + 1]); o.y = pack2bf(hv[ci][4 * g + 2], hv[ci][4 * g + 3]);
            *(uint2*)(stage + ql * RS + (ci * 16 + 8 * g + 4 * kh) * 2) = o;
          }
      } else if constexpr (BN % 64 == 0) {
        // e4m3 hidden tensor with one E8M0 scale per 32 hidden units: a 32-unit block = the blocks ci = 2 c, 2 c + 1 of this lane and
        // of its partner in the other k-half (lane ^ 32)
#pragma unroll
        for (int c = 0; c < NCI / 2; ++c) {
          float am = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) am = fmaxf(am, fmaxf(fabsf(hv[2 * c][k]), fabsf(hv[2 * c + 1][k])));
          am = fmaxf(am, __shfl_xor(am, 32, 64));
          float inv;
          const unsigned e = e8m0_of(am, &inv);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g)
              *(unsigned*)(stage + ql * RS + c * 32 + h * 16 + 8 * g + 4 * kh) =
                  pack4_fp8(hv[2 * c + h][4 * g] * inv, hv[2 * c + h][4 * g + 1] * inv, hv[2 * c + h][4 * g + 2] * inv, hv[2 * c + h][4 * g + 3] * inv);
          const int kb = (n0_ >> 6) + c;                      // 32-unit block index along the hidden axis
          if (kh == 0 && m_ok && (kb << 6) < a.N) a.out_sx[(size_t)kb * a.M + m] = (uint8_t)e;
        }
      }
    } else {
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci) {
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = ci * 32 + 8 * g + 4 * kh, n = n0_ + col;
          const float4 sw4 = *(const float4*)(swl + col), b4 = *(const float4*)(bl + col);
          v[4 * g] = acc[ci][pj][4 * g] * sa * sw4.x + b4.x; v[4 * g + 1] = acc[ci][pj][4 * g + 1] * sa * sw4.y + b4.y;
          v[4 * g + 2] = acc[ci][pj][4 * g + 2] * sa * sw4.z + b4.z; v[4 * g + 3] = acc[ci][pj][4 * g + 3] * sa * sw4.w + b4.w;
          if (a.resid && m_ok && n < a.N) {
            const uint2 rr = *(const uint2*)(a.resid + (long)m * a.ld_res + n);
            v[4 * g] += __uint_as_float(rr.x << 16); v[4 * g + 1] += __uint_as_float(rr.x & 0xffff0000u);
            v[4 * g + 2] += __uint_as_float(rr.y << 16); v[4 * g + 3] += __uint_as_float(rr.y & 0xffff0000u);
          }
        }
        if (!out8) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 o; o.x = pack2bf(v[4 * g], v[4 * g + 1]); o.y = pack2bf(v[4 * g + 2], v[4 * g + 3]);
            *(uint2*)(stage + ql * RS + (ci * 32 + 8 * g + 4 * kh) * 2) = o;
          }
        } else {
          float am = 0.f;
#pragma unroll
          for (int k = 0; k < 16; ++k) am = fmaxf(am, fabsf(v[k]));
          am = fmaxf(am, __shfl_xor(am, 32, 64));
          float inv;
          const unsigned e = e8m0_of(am, &inv);
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(unsigned*)(stage + ql * RS + ci * 32 + 8 * g + 4 * kh) = pack4_fp8(v[4 * g] * inv, v[4 * g + 1] * inv, v[4 * g + 2] * inv, v[4 * g + 3] * inv);
          const int kb = (n0_ >> 5) + ci;
          if (kh == 0 && m_ok && (kb << 5) < a.N) a.out_sx[(size_t)kb * a.M + m] = (uint8_t)e;
        }
      }
    }
    // the wave's own 32-row block -> full rows, 16 bytes per lane (LDS accesses of one wave complete in order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int ocols = geglu ? BN / 2 : BN;                   // output columns of this tile
    const int obase = geglu ? (n0_ >> 1) : n0_, olim = geglu ? (a.N >> 1) : a.N;
    if (!out8) {
      const int cpr = ocols / 8;
      for (int c = lane; c < 32 * cpr; c += 64) {
        const int row = c / cpr, cc = c - row * cpr;
        const int mr = m0_ + wave * PB * 32 + pj * 32 + row, oc = obase + cc * 8;
        if (mr < a.M && oc < olim)
          *(uint4*)((bf16_t*)a.out + (long)mr * a.ld_out + oc) = *(const uint4*)(stage + row * RS + cc * 16);
      }
    } else {
      const int cpr = ocols / 16;
      for (int c = lane; c < 32 * cpr; c += 64) {
        const int row = c / cpr, cc = c - row * cpr;
        const int mr = m0_ + wave * PB * 32 + pj * 32 + row, oc = obase + cc * 16;
        if (mr < a.M && oc < olim)
          *(uint4*)((uint8_t*)a.out + (long)mr * a.ld_out + oc) = *(const uint4*)(stage + row * RS + cc * 16);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next block overwrites the region
    }
  }
}

// ---- quantisers -------------------------------------------------------------------------------------------------------------
// one wave per row of a bf16 [R][K] matrix: scale[r] = amax / 448 (1 for an all-zero row), q = round(x / scale) as e4m3
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int ldx, uint8_t* __restrict__ q,
                                                             float* __restrict__ scale, int R, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const bf16_t* xr = x + (long)row * ldx;
  float amax = 0.f;
  for (int o = lane * 8; o < K; o += 512) {
    float f[8];
    unpack8(*(const uint4*)(xr + o), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(f[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  for (int o = lane * 8; o < K; o += 512) {
    float f[8];
    unpack8(*(const uint4*)(xr + o), f);
    uint2 w;
    w.x = pack4_fp8(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
    w.y = pack4_fp8(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
    *(uint2*)(q + (long)row * K + o) = w;
  }
}

// LayerNorm with the fp8 quantisation of its output fused: one wave per R token rows (C <= 2048), exact two-pass variance in
// registers like layernorm_kernel (norm.hip, incl. its rows-in-flight scheme: all R rows are loaded before the first is reduced);
// the row maximum of |y| (of the fp32 normalised value: it is quantised directly) gives the token's scale.
template <int MAXO, int R>
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, uint8_t* __restrict__ q,
                                                            float* __restrict__ scale, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  const int C8 = C >> 3;
  uint4 raw[R][MAXO];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      const int o = lane + i * 64;
      raw[r][i] = make_uint4(0u, 0u, 0u, 0u);
      if (o < C8 && row0 + r < M) raw[r][i] = *(const uint4*)(x + (long)(row0 + r) * C + o * 8);
    }
  float gg[MAXO][8], bb[MAXO][8];
#pragma unroll
  for (int i = 0; i < MAXO; ++i) {
    const int o = lane + i * 64;
    if (o < C8) {
      const float4 g0 = *(const float4*)(gamma + o * 8), g1 = *(const float4*)(gamma + o * 8 + 4);
      const float4 b0 = *(const float4*)(beta + o * 8), b1 = *(const float4*)(beta + o * 8 + 4);
      gg[i][0] = g0.x; gg[i][1] = g0.y; gg[i][2] = g0.z; gg[i][3] = g0.w; gg[i][4] = g1.x; gg[i][5] = g1.y; gg[i][6] = g1.z; gg[i][7] = g1.w;
      bb[i][0] = b0.x; bb[i][1] = b0.y; bb[i][2] = b0.z; bb[i][3] = b0.w; bb[i][4] = b1.x; bb[i][5] = b1.y; bb[i][6] = b1.z; bb[i][7] = b1.w;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    // the kernel is VALU-issue bound (40 of 64 lanes hold data at C = 320): the centred values of the variance pass are kept and
    // normalised with one multiply + one fma per element; the fp32 value is quantised directly (no bf16 round trip)
    float v[MAXO][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      unpack8(raw[r][i], v[i]);
      if (lane + i * 64 < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[i][k];
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      if (lane + i * 64 < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[i][k] -= mean; qq = fmaf(v[i][k], v[i][k], qq); }
      }
    }
    const float rstd = rsqrtf(wave_sum(qq) / (float)C + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      if (lane + i * 64 < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[i][k] = fmaf(v[i][k], rstd * gg[i][k], bb[i][k]);
          amax = fmaxf(amax, fabsf(v[i][k]));
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      const int o = lane + i * 64;
      if (o < C8) {
        uint2 w;
        w.x = pack4_fp8(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
        w.y = pack4_fp8(v[i][4] * inv, v[i][5] * inv, v[i][6] * inv, v[i][7] * inv);
        *(uint2*)(q + (long)row * C + o * 8) = w;
      }
    }
  }
}

template <int PB, int BN, bool MX, bool TOUT = false>
int launch_fp8(const Fp8GemmArgs& a, hipStream_t s) {
  constexpr int BM = 4 * PB * 32;
  constexpr int lds = NST8 * ((BM + BN) * KB8 + (MX ? 4 * PB * 64 : 0));
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_fp8_kernel<PB, BN, MX, TOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_fp8_kernel<PB, BN, MX, TOUT>), dim3(tiles), dim3(256), lds, s, a);
  return dfh::check_launch("gemm_fp8_kernel");
}

// largest |x| of each (batch element, row range) slab of a bf16 [B][rows][ld] tensor: the cross-attention V^T of one transformer
// layer, whose maximum bounds that layer's attention output (unet_model.h).  One workgroup per (slab, batch element).
__global__ __launch_bounds__(256) void amax_slabs_kernel(const bf16_t* __restrict__ x, long bstride, int ld, int cols, const int* __restrict__ row0,
                                                         const int* __restrict__ nrows, float* __restrict__ out, int nslab) {
  const int sl = blockIdx.x, b = blockIdx.y;
  const bf16_t* base = x + (long)b * bstride + (long)row0[sl] * ld;
  const int c8 = (cols + 7) >> 3;                 // rows are padded to whole 16-byte vectors (ld % 8 == 0); pad elements are masked out
  const long total = (long)nrows[sl] * c8;
  float am = 0.f;
  for (long i = threadIdx.x; i < total; i += 256) {
    const long r = i / c8; const int c = (int)(i - r * c8);
    float f[8];
    unpack8(*(const uint4*)(base + r * ld + c * 8), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) if (c * 8 + k < cols) am = fmaxf(am, fabsf(f[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
  __syncthreads();
  if (threadIdx.x == 0) out[(long)sl * gridDim.y + b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

}  // namespace

namespace dfh {

int gemm_fp8_launch(Fp8GemmArgs a, hipStream_t stream) {
  if (a.lda == 0) a.lda = a.K;
  if (a.sa_mul == 0.f) a.sa_mul = 1.0f;
  if (a.sa_div <= 0) a.sa_div = 1;
  const bool mx = a.sx != nullptr, out8 = a.out_mode == OUT_FP8_MX, geglu = a.act == ACT_GEGLU;
  DFH_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "empty fp8 GEMM");
  DFH_REQUIRE(a.K % KB8 == 0 && a.lda % 16 == 0 && a.lda >= a.K, "fp8 GEMM: K must be a multiple of 64, the row stride of A a multiple of 16");
  DFH_REQUIRE(a.N % 8 == 0, "fp8 GEMM: N must be a multiple of 8");
  DFH_REQUIRE((double)a.M * a.lda < 4.0e9 && (double)a.N * a.K < 4.0e9, "fp8 GEMM: operands must be smaller than 4 GB (32-bit staging offsets)");
  DFH_REQUIRE(a.A && a.W && a.sW && a.zero && a.out, "fp8 GEMM: null operand");
  DFH_REQUIRE(a.out_mode == OUT_BF16 || a.out_mode == OUT_BF16_T || out8, "fp8 GEMM: bf16, transposed bf16 or e4m3 + E8M0 outputs");
  DFH_REQUIRE(a.act == ACT_NONE || geglu, "fp8 GEMM: no activation or GEGLU");
  if (mx) DFH_REQUIRE(a.M % 4 == 0, "fp8 GEMM with E8M0 block scales: M must be a multiple of 4");
  if (geglu) DFH_REQUIRE(a.N % 32 == 0 && !a.resid && a.out_mode != OUT_BF16_T, "fp8 GEGLU: N % 32 == 0, bias only, row-major output");
  if (a.out_mode == OUT_BF16) DFH_REQUIRE(a.ld_out % 8 == 0, "fp8 GEMM: output rows must be 16-byte aligned");
  if (a.out_mode == OUT_BF16_T) DFH_REQUIRE(a.rows_per_b > 0 && !a.resid, "fp8 GEMM: transposed output needs rows_per_b, no residual");
  if (a.amax) DFH_REQUIRE(a.out_mode == OUT_BF16_T, "fp8 GEMM: the output maximum is tracked for the transposed (V^T) output only");
  if (out8) DFH_REQUIRE(a.out_sx && a.ld_out % 16 == 0 && a.N % (geglu ? 128 : 32) == 0,
                        "fp8 GEMM, e4m3 output: needs out_sx, 16-byte aligned rows, N % 32 == 0 (GEGLU: N % 128 == 0)");
  census(CK_GEMM_FP8);
  const double out_bytes = (out8 ? 1.0 + 1.0 / 32 : 2.0) * a.M * (geglu ? a.N / 2 : a.N);
  ProfScope ps(PC_LINEAR_FP8, 2.0 * a.M * a.N * (double)a.K,
               (double)a.M * a.K * (mx ? 1.0 + 1.0 / 32 : 1.0) + (double)a.N * a.K + out_bytes + (a.resid ? 2.0 * a.M * a.N : 0.0), stream);
  // column tile: 128 where the GEGLU epilogue quantises 32-unit blocks of the hidden tensor (64 packed columns each) or 160 does not divide N
  const int bn = (geglu && out8) ? 128 : ((a.N % 160 == 0 || a.N % 128 != 0) ? 160 : 128);
  // 256-row tiles while they still give every CU two workgroups, 128-row tiles below
  const long tiles256 = (long)((a.M + 255) / 256) * ((a.N + bn - 1) / bn);
  const bool big = tiles256 >= 384;
  if (a.out_mode == OUT_BF16_T) {
    DFH_REQUIRE(!mx, "fp8 GEMM: the transposed output takes per-row activation scales (no E8M0 block scales)");
    // 160-wide transposed tiles always on 128 rows: the 256 x 160 instantiation holds 160 accumulator registers through a transposed
    // staging epilogue and spilled 8-13 VGPRs whichever way it was written; the V projection is a K = C launch (13 GFLOP) whose time does
    // not depend on the row tile
    if (bn == 160) return launch_fp8<1, 160, false, true>(a, stream);
    return big ? launch_fp8<2, 128, false, true>(a, stream) : launch_fp8<1, 128, false, true>(a, stream);
  }
  if (bn == 160) {
    if (mx) return big ? launch_fp8<2, 160, true>(a, stream) : launch_fp8<1, 160, true>(a, stream);
    return big ? launch_fp8<2, 160, false>(a, stream) : launch_fp8<1, 160, false>(a, stream);
  }
  if (mx) return big ? launch_fp8<2, 128, true>(a, stream) : launch_fp8<1, 128, true>(a, stream);
  return big ? launch_fp8<2, 128, false>(a, stream) : launch_fp8<1, 128, false>(a, stream);
}

int amax_slabs_launch(const bf16_t* x, long bstride, int ld, int cols, const int* row0, const int* nrows, float* out, int nslab, int B,
                      hipStream_t stream) {
  DFH_REQUIRE(cols > 0 && cols <= ld && ld % 8 == 0 && nslab > 0 && B > 0, "amax_slabs: rows of ld elements, ld a multiple of 8");
  hipLaunchKernelGGL(amax_slabs_kernel, dim3(nslab, B), dim3(256), 0, stream, x, bstride, ld, cols, row0, nrows, out, nslab);
  return check_launch("amax_slabs_kernel");
}

int quant_rows_fp8_launch(const bf16_t* x, int ldx, uint8_t* q, float* scale, int R, int K, hipStream_t stream) {
  DFH_REQUIRE(R > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0, "fp8 row quantiser: K and the row stride must be multiples of 8");
  hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, x, ldx, q, scale, R, K);
  return check_launch("quant_rows_fp8_kernel");
}

int layernorm_fp8_launch(const bf16_t* x, const float* gamma, const float* beta, uint8_t* q, float* scale, int M, int C, float eps,
                         hipStream_t stream) {
  DFH_REQUIRE(C % 8 == 0 && C <= 8 * 64 * 4, "LayerNorm width must be a multiple of 8 and <= 2048");
  const dim3 block(256);
  ProfScope ps(PC_LNORM, 0.0, 3.0 * (double)M * C + 4.0 * M, stream);
  if (C <= 512) hipLaunchKernelGGL((layernorm_fp8_kernel<1, 4>), dim3((M + 15) / 16), block, 0, stream, x, gamma, beta, q, scale, M, C, eps);
  else if (C <= 1024) hipLaunchKernelGGL((layernorm_fp8_kernel<2, 2>), dim3((M + 7) / 8), block, 0, stream, x, gamma, beta, q, scale, M, C, eps);
  else hipLaunchKernelGGL((layernorm_fp8_kernel<4, 1>), dim3((M + 3) / 4), block, 0, stream, x, gamma, beta, q, scale, M, C, eps);
  return check_launch("layernorm_fp8_kernel");
}

}  // namespace dfh
