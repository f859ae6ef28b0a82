// PROBE kernel (lost its A/B: profiles/r05/token_linear_ab.txt -- 37 / 46 us against 27 / 34 us of the tile GEMM per launch at M = 65536,
// sampling step 15.7 -> 15.9 ms).  Built only into scripts/probes/build/libdifashion_probes.so; reached through dfh_gemm / dfh_gemm_ln with
// tile id 30 (K = N = 320, M a multiple of 128) and through the walk with DFH_TOKEN_LINEAR=1.
#include "mlp_fused.h"
#include "token_linear.h"

#include <cstdlib>
#include <cstring>
#include <map>

namespace {

#include "mlp_fused2_core.h"

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


// dfh_gemm tile id 30: the launch as a token linear (the image of W is packed on first use and cached per weight pointer)
int token_linear_from_gemm(const GemmArgs& g, hipStream_t stream) {
  DFH_REQUIRE(g.ntaps == 0 && g.nplain == 1 && g.p_c[0] == MC && token_linear_eligible(g.N, g.p_c[0], g.M) && g.out_mode == OUT_BF16 && g.ld_out == MC &&
              g.act == ACT_NONE && !g.rowvec && (!g.resid || g.ld_res == MC), "tile id 30: a K = N = 320 row-major linear over whole 128-token tiles");
  static std::map<const void*, void*> cache;
  void*& img = cache[g.W];
  if (!img) {
    if (hipMalloc(&img, token_linear_image_bytes()) != hipSuccess) { set_error("hipMalloc of a token-linear image failed"); return -1; }
  }
  if (int rc = token_linear_pack_launch(g.W, g.ldw, img, stream)) return rc;
  TokLinArgs t; std::memset(&t, 0, sizeof(t));
  t.x = g.p_src[0]; t.img = (const unsigned char*)img; t.bias = g.bias; t.resid = g.resid; t.ln_stat = g.ln_stat; t.ln_parts = g.ln_parts;
  t.ln_cnt = g.ln_cnt; t.ln_eps = g.ln_eps; t.ln_s = g.ln_s; t.rowstat = g.rowstat; t.out = (bf16_t*)g.out; t.M = g.M;
  return token_linear_launch(t, stream);
}

}  // namespace dfh
