// Shared device/host helpers for the DiFashion gfx950 kernels.
// Layout conventions (DESIGN.md "Data layout in HBM"):
//   activations  : bf16, NHWC == [B][H*W][C] token-major (so convs, 1x1 convs and the
//                  transformer linears all see the same row-major [M][K] A operand)
//   GEMM weights : bf16, [N][K] row-major (K contiguous); conv3x3 K index = (ky*3+kx)*Cin + c
//   bias / temb  : fp32
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define DFH_DEVICE __device__ __forceinline__

DFH_DEVICE float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even, through the native __bf16 conversion: hipcc lowers it to
// gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; a hand-rolled RNE costs ~8 VALU + a NaN branch)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
DFH_DEVICE bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
DFH_DEVICE uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

DFH_DEVICE void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

DFH_DEVICE uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division: the division expands to ~10 VALU
// instructions (v_div_scale / v_rcp / 4 x v_fma / v_div_fmas / v_div_fixup), which made GroupNorm+SiLU VALU-bound -- 42 M
// elements per 64x64-level launch at ~70 wave-cycles per 64 of them is 19 us of a 27 us kernel.  The result is rounded to bf16
// (8 mantissa bits) right after; limits as before: -0 for x -> -inf side, x for x -> +inf.
DFH_DEVICE float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
// exact-erf GELU (diffusers GEGLU uses F.gelu, erf form).  erf by Abramowitz-Stegun 7.1.26:
// |error| <= 1.5e-7 absolute -- three orders below the bf16 rounding of the result -- at 13 VALU ops
// (one v_rcp, one v_exp) instead of libm erff's ~33.
DFH_DEVICE float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
  return copysignf(1.0f - p * t * e, x);
}
// GELU (erf form) as used by every GEGLU epilogue: 84 M gate elements per ff.net.0 launch, all on the VALU while the MFMA pipe
// idles, so the op count is launch time.  erfc(|x|/sqrt2) = 2^P(|x|) with P a degree-6 polynomial (minimax fit of log2 erfc on
// [0, 4 sqrt2], the 1/sqrt2 folded into the coefficients; beyond the clamp erfc < 1.6e-8): one v_exp and six FMAs, no
// reciprocal, and gelu(x) = max(x, 0) - |x * 2^(P - 1)| covers both signs without a select.  |error| <= 5.5e-7 absolute over all x
// (the same as the Abramowitz-Stegun form above at |x| ~ 4, three orders below the bf16 rounding of the product) at ~10 VALU
// slots against ~19.
DFH_DEVICE float gelu_erf_f(float x) {
  const float ax = fminf(fabsf(x), 5.65685424949f);
  float p = 1.917432119e-05f;
  p = fmaf(p, ax, -6.586021110e-04f);
  p = fmaf(p, ax, 7.754402186e-03f);
  p = fmaf(p, ax, -5.296538429e-02f);
  p = fmaf(p, ax, -4.590602584e-01f);
  p = fmaf(p, ax, -1.151122051e+00f);
  p = fmaf(p, ax, 3.063254510e-07f - 1.0f);          // - 1: the 0.5 of x / 2 * erfc rides in the exponent
  const float h = x * __builtin_amdgcn_exp2f(p);
  return fmaxf(x, 0.0f) - fabsf(h);
}

// The same GELU on a PAIR of values with packed fp32 arithmetic (v_pk_fma_f32: two fp32 lanes per VALU slot).  The GEGLU epilogues are
// VALU-bound (in-kernel stamps: ~22 issue slots per output element, profiles/r03/geglu_phases.txt), and left to itself hipcc either keeps
// the Horner chain scalar (v_fmaak_f32, one slot per element and step) or packs it and re-materialises every constant pair with two
// v_mov per use.  Here the seven coefficients sit in four register pairs and each v_pk_fma_f32 broadcasts the half it needs with
// op_sel: three slots per element for the polynomial instead of six.  Every lane computes exactly gelu_erf_f (fma is fma).
struct GeluK { f32x2_t k65, k43, k21, k0; };
DFH_DEVICE GeluK gelu_consts() {
  GeluK k;
  k.k65 = f32x2_t{1.917432119e-05f, -6.586021110e-04f}; k.k43 = f32x2_t{7.754402186e-03f, -5.296538429e-02f};
  k.k21 = f32x2_t{-4.590602584e-01f, -1.151122051e+00f}; k.k0 = f32x2_t{3.063254510e-07f - 1.0f, 0.0f};
  asm volatile("" : "+v"(k.k65), "+v"(k.k43), "+v"(k.k21), "+v"(k.k0));      // live register pairs, not literals to re-materialise
  return k;
}
DFH_DEVICE f32x2_t gelu_erf_f2(f32x2_t x, const GeluK& k) {
  // min(|x|, 4 sqrt2) and max(x, 0) as ONE instruction each (fminf / fmaxf / fabsf cost a canonicalising v_max apiece through the compiler)
  const float clampv = 5.65685424949f;
  f32x2_t ax, rl;
  asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(ax[0]) : "v"(x[0]), "s"(clampv));
  asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(ax[1]) : "v"(x[1]), "s"(clampv));
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(rl[0]) : "v"(x[0]));
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(rl[1]) : "v"(x[1]));
  f32x2_t p;
  // p = c6 * ax + c5   (src0 = low half of k65 in both lanes, src2 = high half in both lanes)
  asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(p) : "v"(k.k65), "v"(ax));
  // p = p * ax + c   (src2 = low / high half of the pair in both lanes)
  asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(p) : "v"(ax), "v"(k.k43));
  asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(p) : "v"(ax), "v"(k.k43));
  asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(p) : "v"(ax), "v"(k.k21));
  asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(p) : "v"(ax), "v"(k.k21));
  asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(p) : "v"(ax), "v"(k.k0));
  const f32x2_t e = f32x2_t{__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
  const f32x2_t h = x * e;
  f32x2_t r;
  asm("v_sub_f32_e64 %0, %1, |%2|" : "=v"(r[0]) : "v"(rl[0]), "v"(h[0]));
  asm("v_sub_f32_e64 %0, %1, |%2|" : "=v"(r[1]) : "v"(rl[1]), "v"(h[1]));
  return r;
}
// GEGLU of four (value, gate) accumulator pairs of one lane: (v + bias-or-fix-up) * gelu(g + ...) -> four bf16 in a uint2.
//   lnf: v' = rstd * v + (ms * s + b),  ms = -mean * rstd  (folded LayerNorm, gemm.h);  else v' = v + b
DFH_DEVICE uint2 geglu4(const f32x4_t v, const f32x4_t g, const float4 bv, const float4 bg, const float4 sv, const float4 sg, const bool lnf,
                        const float rstd, const float ms, const GeluK& k) {
  f32x2_t v0 = f32x2_t{v[0], v[1]}, v1 = f32x2_t{v[2], v[3]}, g0 = f32x2_t{g[0], g[1]}, g1 = f32x2_t{g[2], g[3]};
  const f32x2_t bv0 = f32x2_t{bv.x, bv.y}, bv1 = f32x2_t{bv.z, bv.w}, bg0 = f32x2_t{bg.x, bg.y}, bg1 = f32x2_t{bg.z, bg.w};
  if (lnf) {
    const f32x2_t r2 = f32x2_t{rstd, rstd}, m2 = f32x2_t{ms, ms};
    v0 = __builtin_elementwise_fma(r2, v0, __builtin_elementwise_fma(m2, f32x2_t{sv.x, sv.y}, bv0));
    v1 = __builtin_elementwise_fma(r2, v1, __builtin_elementwise_fma(m2, f32x2_t{sv.z, sv.w}, bv1));
    g0 = __builtin_elementwise_fma(r2, g0, __builtin_elementwise_fma(m2, f32x2_t{sg.x, sg.y}, bg0));
    g1 = __builtin_elementwise_fma(r2, g1, __builtin_elementwise_fma(m2, f32x2_t{sg.z, sg.w}, bg1));
  } else {
    v0 = v0 + bv0; v1 = v1 + bv1; g0 = g0 + bg0; g1 = g1 + bg1;
  }
  const f32x2_t o0 = v0 * gelu_erf_f2(g0, k), o1 = v1 * gelu_erf_f2(g1, k);
  uint2 o;
  o.x = pack2bf(o0[0], o0[1]); o.y = pack2bf(o1[0], o1[1]);
  return o;
}

// geglu4 that also hands back the four (value, gate) PRE-activations as bf16 (training: the backward needs them, gemm.h GemmArgs::pre_out)
DFH_DEVICE uint2 geglu4p(const f32x4_t v, const f32x4_t g, const float4 bv, const float4 bg, const GeluK& k, uint2& pre_v, uint2& pre_g) {
  const f32x2_t v0 = f32x2_t{v[0] + bv.x, v[1] + bv.y}, v1 = f32x2_t{v[2] + bv.z, v[3] + bv.w};
  const f32x2_t g0 = f32x2_t{g[0] + bg.x, g[1] + bg.y}, g1 = f32x2_t{g[2] + bg.z, g[3] + bg.w};
  pre_v.x = pack2bf(v0[0], v0[1]); pre_v.y = pack2bf(v1[0], v1[1]);
  pre_g.x = pack2bf(g0[0], g0[1]); pre_g.y = pack2bf(g1[0], g1[1]);
  const f32x2_t o0 = v0 * gelu_erf_f2(g0, k), o1 = v1 * gelu_erf_f2(g1, k);
  uint2 o;
  o.x = pack2bf(o0[0], o0[1]); o.y = pack2bf(o1[0], o1[1]);
  return o;
}

// Value of lane (l ^ 16) / (l ^ 32): gfx950's v_permlane{16,32}_swap exchange 16- / 32-lane rows between two registers in
// the VALU (a few cycles); __shfl_xor goes through ds_bpermute_b32, an LDS round trip of ~100+ cycles on the critical path
// of every softmax tile.  swap(x, x) leaves {x of the even row, x of the odd row} of each row pair in both rows.
DFH_DEVICE float lane_xor16(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const bool odd_row = (__lane_id() & 16) != 0;
  return __uint_as_float(odd_row ? r[0] : r[1]);
}
DFH_DEVICE float lane_xor32(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const bool upper = (__lane_id() & 32) != 0;
  return __uint_as_float(upper ? r[0] : r[1]);
}
// max / sum over the four 16-lane rows (lanes l, l^16, l^32, l^48): no select needed, both halves of a swap are combined
DFH_DEVICE float rows_max(float x) {
  unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
DFH_DEVICE float rows_sum(float x) {
  unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
DFH_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return rows_sum(v);
}

// ---- host side ------------------------------------------------------------------------------
#include <string>
namespace dfh {
void set_error(const std::string& msg);  // api.cpp
int check_launch(const char* what);     // returns 0 or negative and records the message
// Optional per-launch timing with HIP events recorded on the launch stream (bench.py roofline).
enum ProfClass { PC_CONV3 = 0, PC_LINEAR = 1, PC_ATTN = 2, PC_GNORM = 3, PC_LNORM = 4, PC_SPLITK = 5, PC_OTHER = 6,
                 PC_WGRAD = 7, PC_ATTN_BWD = 8, PC_NORM_BWD = 9, PC_OPTIM = 10, PC_LINEAR_FP8 = 11, PC_COUNT = 12 };
// Launch census: one counter per kernel family, bumped by the launchers (tests assert through it WHICH kernels a walk ran, e.g. that
// the batch-16 forward really took the wide tile, the producer-statistics GroupNorm and the split-K path; dfh_census_* in the C ABI)
enum CensusId { CK_GEMM_WIDE = 0, CK_GEMM_8WAVE, CK_GEMM_LEAN, CK_GEMM_OTHER, CK_GEMM_ROW, CK_SPLITK, CK_SPLITK_FUSED, CK_GSTAT_WRITTEN,
                CK_GN_PRE, CK_GN_STATS, CK_GN_SMALL, CK_GN_MID, CK_LAYERNORM, CK_LN_FOLDED, CK_ATTN_X32, CK_ATTN_16, CK_GEMM_FP8,
                CK_TEXT_CACHED, CK_CONV_PHASE, CK_CONV_WINO, CK_GEMM_ROWS_GEGLU, CK_ATTN_FP8, CK_MLP_FUSED, CK_GN_FOLDED, CK_TOKEN_LINEAR, CK_DUP_PREFIX, CK_GEMM_PERSIST, CK_COUNT };
void census(int id);
bool prof_enabled();
void prof_open(int cls, double flops, double bytes, hipStream_t s);   // no-ops unless enabled
void prof_close(hipStream_t s);
void prof_note_saved(double flops);   // reference-algorithm flops a launch reports but does not execute (phase planes, Winograd)
struct ProfScope {
  hipStream_t s; bool on;
  ProfScope(int cls, double flops, double bytes, hipStream_t st) : s(st), on(prof_enabled()) { if (on) prof_open(cls, flops, bytes, s); }
  ~ProfScope() { if (on) prof_close(s); }
};
}  // namespace dfh
#define DFH_REQUIRE(cond, msg)                                   \
  do {                                                           \
    if (!(cond)) {                                               \
      dfh::set_error(std::string(__func__) + ": " + (msg));      \
      return -1;                                                 \
    }                                                            \
  } while (0)

// Bijective XCD-aware remap of a linear block id (guide T1): consecutive logical tiles land on
// the same XCD (= same L2), so neighbouring tiles that share an A row panel or a W column panel
// hit in L2.  Placement only affects speed, never results.
DFH_DEVICE int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Tile id -> (row tile, column tile) of an ntm x ntn GEMM grid.  What matters for the private 4-MiB L2 of an XCD is the set of
// tiles it runs CONCURRENTLY (~64: two workgroups on each of its 32 CUs, marching through K roughly in lock step): a set of
// gm row tiles x 64/gm column tiles fetches gm A panels + 64/gm W panels from the fabric, every other read is an L2 hit.
// Plain m-major ids make that set 1-3 row tiles x ALL column tiles, so with many column tiles (GEGLU projections: 40 / 80)
// every XCD re-streams the whole weight matrix once per row tile (measured with FETCH_SIZE: 11-12 x the algorithmic bytes at
// the 32x32 / 16x16 levels, profiles/r02/pmc_traffic_calib.txt).  Two knobs, both chosen by the launcher:
//   xm: the 8 XCDs form an xm x (8/xm) grid over the tile grid (only when xm | ntm and 8/xm | ntn; otherwise each XCD takes a
//       contiguous range of the m-major id list as before), so an XCD can own a rectangle with enough rows AND columns;
//   gm: inside its rectangle an XCD walks groups of gm row tiles column by column (ids run down a column of gm row tiles,
//       then step to the next column tile).
// gm == 0 keeps the legacy order (m-major, or n-major when n_major is set).  Placement never changes results.
DFH_DEVICE void tile_coords(int bid, int ntm, int ntn, int n_major, int xm, int gm, int& mt, int& nt) {
  if (gm <= 0) {
    const int tile = xcd_remap(bid, ntm * ntn);
    mt = n_major ? tile % ntm : tile / ntn;
    nt = n_major ? tile / ntm : tile % ntn;
    return;
  }
  int rm = ntm, rn = ntn, mbase = 0, nbase = 0, idx;
  const int xn = xm > 0 ? 8 / xm : 0;
  if (xm > 0 && ntm % xm == 0 && ntn % xn == 0) {
    const int xcd = bid & 7;
    rm = ntm / xm; rn = ntn / xn;
    mbase = (xcd / xn) * rm; nbase = (xcd % xn) * rn;
    idx = bid >> 3;
  } else {
    idx = xcd_remap(bid, ntm * ntn);
  }
  const int per = gm * rn, group = idx / per, first = group * gm;
  const int gsz = (rm - first) < gm ? (rm - first) : gm;
  const int w = idx - group * per;
  mt = mbase + first + w % gsz;
  nt = nbase + w / gsz;
}
