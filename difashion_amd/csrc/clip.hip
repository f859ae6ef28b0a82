// CLIP text encoder: the frozen prompt encoder on either side of the denoising path.
//
// Replaces (arithmetic): transformers CLIPTextModel as the reference builds and calls it -- constructed at
// DiFashion/models/difashion.py:66-75, run on the category prompts of every training batch (:218-224, null prompt :226-234) and once
// per sampling call (:340-353); only ``[0]`` (last_hidden_state) is consumed.  The reference pins transformers 4.32.1 (README.md:24);
// the arithmetic is restated in oracle/clip_ref.py and PINNED against the installed transformers class (tests/golden/make_golden_clip.py).
//
// Design: the prompts are a closed set (one sentence per category + the empty prompt: <= 51 sequences of 77 tokens, data_utils.py:96-111),
// encoded ONCE per run and reused by every denoising step of every outfit (difashion_amd/prompts.py PromptTable).  The whole encoder is
// 0.67 TFLOP (CLIP-L) / 2.1 TFLOP (OpenCLIP-H) -- a few ms on any MFMA path -- while its output conditions all 50 x 16 U-Net forwards.
// So this file buys PRECISION, not speed: everything stays fp32, the linears run on the fp32 matrix instruction
// (v_mfma_f32_16x16x4_f32, 256 FLOP / clk / CU), weights are read in place from the fp32 master parameters (nn.Linear layout [N][K],
// K contiguous: no packed copy, no arena), and the result agrees with the fp32 reference class to summation-order noise (1e-6),
// not to bf16 noise.  No atomics: reruns are bit-identical.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/difashion_hip.h"
#include "dfh_common.h"

namespace {

// ------------------------------------------------------------------ embeddings: x[b][t] = token_embedding[ids[b][t]] + position_embedding[t]
// (CLIPTextEmbeddings.forward; position_ids = arange(T))
__global__ __launch_bounds__(256) void clip_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                         const float* __restrict__ pos, float* __restrict__ x, int T, int D, int vocab) {
  const int m = blockIdx.x, t = m % T;
  long id = ids[m];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // the host side refuses out-of-range ids before the launch
  const float4* a = (const float4*)(tok + id * (long)D);
  const float4* p = (const float4*)(pos + (long)t * D);
  float4* o = (float4*)(x + (long)m * D);
  for (int c = threadIdx.x; c < D / 4; c += 256) {
    const float4 u = a[c], v = p[c];
    o[c] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}

// ------------------------------------------------------------------ LayerNorm over the last dim, fp32 in / out, one wave per row
// (two-pass mean / biased variance like torch.nn.functional.layer_norm)
__global__ __launch_bounds__(256) void clip_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ b, float* __restrict__ y, int M, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float4* r = (const float4*)(x + (long)row * D);
  float s = 0.f;
  for (int c = lane; c < D / 4; c += 64) { const float4 v = r[c]; s += (v.x + v.y) + (v.z + v.w); }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int c = lane; c < D / 4; c += 64) {
    const float4 v = r[c];
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  float4* o = (float4*)(y + (long)row * D);
  for (int c = lane; c < D / 4; c += 64) {
    const float4 v = r[c], gg = ((const float4*)g)[c], bb = ((const float4*)b)[c];
    o[c] = make_float4((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y, (v.z - mean) * rstd * gg.z + bb.z,
                       (v.w - mean) * rstd * gg.w + bb.w);
  }
}

// ------------------------------------------------------------------ fp32 linear on the fp32 matrix pipe
//   out[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ resid[m][n])
// 64 x 64 output tile per workgroup (four waves, 32 x 32 each = 2 x 2 v_mfma_f32_16x16x4_f32 blocks), K in steps of 32.
// Operand tiles sit k-major in LDS ([k][row], row stride 80 floats): a fragment read (lane -> row l % 16, k l / 16) touches 64 distinct
// banks, and the transposing store of a float4 along k writes consecutive rows of one k.  The next k-tile's global loads are issued
// before the current tile's MFMAs (register prefetch).
constexpr int CBM = 64, CBN = 64, CBK = 32, CLD = 80;
enum { CLIP_ACT_NONE = 0, CLIP_ACT_QUICK_GELU = 1, CLIP_ACT_GELU = 2 };

DFH_DEVICE float clip_act(float v, int act) {
  if (act == CLIP_ACT_QUICK_GELU) return v / (1.0f + expf(-1.702f * v));           // x * sigmoid(1.702 x)
  if (act == CLIP_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

__global__ __launch_bounds__(256) void clip_gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, const float* resid, int ld_res,
                                                            float* out, int ld_out, int M, int N, int K, int act) {
  __shared__ float As[CBK * CLD];
  __shared__ float Ws[CBK * CLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * CBM, n0 = blockIdx.x * CBN;
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  // staging: thread -> (row = tid % 64, k-quads tid / 64 and tid / 64 + 4)
  const int srow = tid & 63, sq = tid >> 6;
  const int am = min(m0 + srow, M - 1), wr = min(n0 + srow, N - 1);
  const float* ap = A + (long)am * lda;
  const float* wp = W + (long)wr * ldw;
  float4 ra[2], rw[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + (sq + 4 * j) * 4;
      ra[j] = k < K ? *(const float4*)(ap + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      rw[j] = k < K ? *(const float4*)(wp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = (sq + 4 * j) * 4;
      As[(kk + 0) * CLD + srow] = ra[j].x; As[(kk + 1) * CLD + srow] = ra[j].y;
      As[(kk + 2) * CLD + srow] = ra[j].z; As[(kk + 3) * CLD + srow] = ra[j].w;
      Ws[(kk + 0) * CLD + srow] = rw[j].x; Ws[(kk + 1) * CLD + srow] = rw[j].y;
      Ws[(kk + 2) * CLD + srow] = rw[j].z; Ws[(kk + 3) * CLD + srow] = rw[j].w;
    }
  };
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fk = lane >> 4;
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += CBK) {
    __syncthreads();                       // the previous tile's fragment reads are done
    stash();
    __syncthreads();
    if (k0 + CBK < K) fetch(k0 + CBK);
#pragma unroll
    for (int kk = 0; kk < CBK; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(kk + fk) * CLD + wm + 16 * i + fr];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Ws[(kk + fk) * CLD + wn + 16 * j + fr];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // C layout of 16x16x4: acc[r] = C[4 * (lane / 16) + r][lane % 16]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + 16 * j + fr;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + 16 * i + 4 * fk + r;
        if (m >= M) continue;
        float v = clip_act(acc[i][j][r] + bv, act);
        if (resid) v += resid[(long)m * ld_res + n];
        out[(long)m * ld_out + n] = v;
      }
    }
}

// ------------------------------------------------------------------ causal self-attention over one (sequence, head): T <= 128 tokens
// qkv [B * T][3 D] fp32 (q | k | v), O [B * T][D].  K (row stride d + 1) and V of the head sit in LDS; a wave owns query rows
// wave, wave + 4, ...: lane j scores keys j and j + 64 (masked beyond the query position: CLIPTextModel's causal mask, no padding
// mask -- the reference passes input_ids only), softmax in fp32 with the row maximum, then lane c accumulates output channel c.
__global__ __launch_bounds__(256) void clip_attention_kernel(const float* __restrict__ qkv, float* __restrict__ O, int T, int D, int d,
                                                             float scale) {
  extern __shared__ float smem[];
  float* Ks = smem;                         // [T][d + 1]
  float* Vs = Ks + T * (d + 1);             // [T][d]
  float* Qw = Vs + T * d;                   // [4][d]
  float* Pw = Qw + 4 * d;                   // [4][128]
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* base = qkv + (long)b * T * 3 * D + h * d;
  for (int e = tid; e < T * d; e += 256) {
    const int t = e / d, c = e - t * d;
    Ks[t * (d + 1) + c] = base[(long)t * 3 * D + D + c];
    Vs[t * d + c] = base[(long)t * 3 * D + 2 * D + c];
  }
  __syncthreads();
  float* q = Qw + wave * d;
  float* p = Pw + wave * 128;
  for (int i = wave; i < T; i += 4) {
    for (int c = lane; c < d; c += 64) q[c] = base[(long)i * 3 * D + c];
    __builtin_amdgcn_wave_barrier();
    float s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane + 64 * u;
      float a = -INFINITY;
      if (j <= i) {
        a = 0.f;
        const float* kr = Ks + j * (d + 1);
        for (int c = 0; c < d; ++c) a = fmaf(q[c], kr[c], a);
        a *= scale;
      }
      s[u] = a;
    }
    float mx = fmaxf(s[0], s[1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e0 = lane <= i ? expf(s[0] - mx) : 0.f, e1 = lane + 64 <= i ? expf(s[1] - mx) : 0.f;
    float sum = e0 + e1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    p[lane] = e0 * inv; p[lane + 64] = e1 * inv;
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < d; c += 64) {
      float o = 0.f;
      for (int j = 0; j <= i; ++j) o = fmaf(p[j], Vs[j * d + c], o);
      O[((long)b * T + i) * D + h * d + c] = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// pooler_output[b] = last_hidden_state[b][pos]: pos = argmax(ids) when eos_token_id == 2 (the legacy rule of transformers 4.32.1's
// CLIPTextTransformer), else the first position holding eos_token_id (0 when there is none: argmax of an all-false row)
__global__ __launch_bounds__(256) void clip_pool_kernel(const int64_t* __restrict__ ids, const float* __restrict__ y, float* __restrict__ pooled,
                                                        int T, int D, int eos) {
  __shared__ int pos;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    int best = 0;
    if (eos == 2) {
      long mx = ids[(long)b * T];
      for (int t = 1; t < T; ++t) if (ids[(long)b * T + t] > mx) { mx = ids[(long)b * T + t]; best = t; }
    } else {
      for (int t = 0; t < T; ++t) if (ids[(long)b * T + t] == eos) { best = t; break; }
    }
    pos = best;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) pooled[(long)b * D + c] = y[((long)b * T + pos) * D + c];
}

struct ClipLayer { int kw, kb, vw, vb, qw, qb, ow, ob, ln1w, ln1b, f1w, f1b, f2w, f2b, ln2w, ln2b; };

}  // namespace

struct dfh_clip {
  dfh_clip_config cfg{};
  struct P { std::string name; std::vector<int> shape; };
  std::vector<P> params;
  int tok = 0, pos = 0, fw = 0, fb = 0;
  std::vector<ClipLayer> layers;
  int add(const std::string& n, std::vector<int> s) { params.push_back({n, std::move(s)}); return (int)params.size() - 1; }
};

extern "C" {

int dfh_clip_create(const dfh_clip_config* cfg, dfh_clip** out) {
  DFH_REQUIRE(cfg && out, "null argument");
  DFH_REQUIRE(cfg->hidden_size > 0 && cfg->hidden_size % 4 == 0 && cfg->intermediate_size % 4 == 0, "hidden / intermediate size must be multiples of 4");
  DFH_REQUIRE(cfg->num_attention_heads > 0 && cfg->hidden_size % cfg->num_attention_heads == 0, "hidden_size must divide into the heads");
  DFH_REQUIRE(cfg->max_position_embeddings > 0 && cfg->max_position_embeddings <= 128, "at most 128 positions (CLIP: 77)");
  DFH_REQUIRE(cfg->hidden_act == CLIP_ACT_QUICK_GELU || cfg->hidden_act == CLIP_ACT_GELU, "hidden_act: 1 = quick_gelu, 2 = gelu");
  DFH_REQUIRE(cfg->vocab_size > 0 && cfg->num_hidden_layers > 0, "vocab_size / num_hidden_layers");
  dfh_clip* c = new dfh_clip();
  c->cfg = *cfg;
  const int D = cfg->hidden_size, I = cfg->intermediate_size;
  // transformers 4.32.1 state-dict names and order (CLIPTextModel -> text_model.*)
  const std::string tm = "text_model.";
  c->tok = c->add(tm + "embeddings.token_embedding.weight", {cfg->vocab_size, D});
  c->pos = c->add(tm + "embeddings.position_embedding.weight", {cfg->max_position_embeddings, D});
  for (int l = 0; l < cfg->num_hidden_layers; ++l) {
    const std::string p = tm + "encoder.layers." + std::to_string(l) + ".";
    ClipLayer L;
    L.kw = c->add(p + "self_attn.k_proj.weight", {D, D}); L.kb = c->add(p + "self_attn.k_proj.bias", {D});
    L.vw = c->add(p + "self_attn.v_proj.weight", {D, D}); L.vb = c->add(p + "self_attn.v_proj.bias", {D});
    L.qw = c->add(p + "self_attn.q_proj.weight", {D, D}); L.qb = c->add(p + "self_attn.q_proj.bias", {D});
    L.ow = c->add(p + "self_attn.out_proj.weight", {D, D}); L.ob = c->add(p + "self_attn.out_proj.bias", {D});
    L.ln1w = c->add(p + "layer_norm1.weight", {D}); L.ln1b = c->add(p + "layer_norm1.bias", {D});
    L.f1w = c->add(p + "mlp.fc1.weight", {I, D}); L.f1b = c->add(p + "mlp.fc1.bias", {I});
    L.f2w = c->add(p + "mlp.fc2.weight", {D, I}); L.f2b = c->add(p + "mlp.fc2.bias", {D});
    L.ln2w = c->add(p + "layer_norm2.weight", {D}); L.ln2b = c->add(p + "layer_norm2.bias", {D});
    c->layers.push_back(L);
  }
  c->fw = c->add(tm + "final_layer_norm.weight", {D});
  c->fb = c->add(tm + "final_layer_norm.bias", {D});
  *out = c;
  return 0;
}
void dfh_clip_destroy(dfh_clip* c) { delete c; }
int dfh_clip_num_params(const dfh_clip* c) { return (int)c->params.size(); }
const char* dfh_clip_param_name(const dfh_clip* c, int i) { return c->params[i].name.c_str(); }
int dfh_clip_param_ndim(const dfh_clip* c, int i) { return (int)c->params[i].shape.size(); }
int dfh_clip_param_dim(const dfh_clip* c, int i, int d) { return c->params[i].shape[d]; }

static size_t clip_ws_floats(const dfh_clip* c, int batch, int T) {
  const size_t M = (size_t)batch * T, D = c->cfg.hidden_size, I = c->cfg.intermediate_size;
  return M * (D /* x */ + D /* ln */ + 3 * D /* qkv */ + D /* attention */ + I /* hidden */) + 64;
}
size_t dfh_clip_workspace_bytes(const dfh_clip* c, int batch, int seq_len) { return clip_ws_floats(c, batch, seq_len) * sizeof(float) + 256; }

static int clip_linear(const float* A, int lda, const float* W, int K, const float* bias, const float* resid, int ld_res, float* out,
                       int ld_out, int M, int N, int act, hipStream_t s) {
  dfh::ProfScope ps(dfh::PC_OTHER, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), s);
  const dim3 grid((N + CBN - 1) / CBN, (M + CBM - 1) / CBM);
  hipLaunchKernelGGL(clip_gemm_f32_kernel, grid, dim3(256), 0, s, A, lda, W, K, bias, resid, ld_res, out, ld_out, M, N, K, act);
  return dfh::check_launch("clip_gemm_f32_kernel");
}

int dfh_clip_encode(dfh_clip* c, const float* const* master_params, int count, const int64_t* input_ids, float* last_hidden_state,
                    float* pooler_output, int eos_token_id, float* hidden_states, void* workspace, size_t workspace_bytes, int batch,
                    int seq_len, void* stream) {
  DFH_REQUIRE(c && master_params && input_ids && last_hidden_state && workspace, "null argument");
  DFH_REQUIRE(count == (int)c->params.size(), "master_params count does not match dfh_clip_num_params");
  for (int i = 0; i < count; ++i) DFH_REQUIRE(master_params[i] != nullptr, "null parameter pointer: " + c->params[i].name);
  DFH_REQUIRE(batch > 0 && seq_len > 0 && seq_len <= c->cfg.max_position_embeddings,
              "sequence length must be in [1, max_position_embeddings] (CLIPTextEmbeddings raises too)");
  DFH_REQUIRE(workspace_bytes >= dfh_clip_workspace_bytes(c, batch, seq_len), "workspace smaller than dfh_clip_workspace_bytes");
  DFH_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int T = seq_len, D = c->cfg.hidden_size, I = c->cfg.intermediate_size, H = c->cfg.num_attention_heads, d = D / H;
  const int M = batch * T;
  const size_t lds = ((size_t)T * (2 * d + 1) + 4 * d + 4 * 128) * sizeof(float);
  DFH_REQUIRE(lds <= 64 * 1024, "head_dim x sequence length does not fit the attention kernel's LDS tile");
  float* x = (float*)workspace;
  float* ln = x + (size_t)M * D;
  float* qkv = ln + (size_t)M * D;
  float* att = qkv + (size_t)M * 3 * D;
  float* hid = att + (size_t)M * D;
  const float* const* P = master_params;
  const float eps = c->cfg.layer_norm_eps, scale = 1.0f / sqrtf((float)d);
  const dim3 ln_grid((M + 3) / 4);
  hipLaunchKernelGGL(clip_embed_kernel, dim3(M), dim3(256), 0, s, input_ids, P[c->tok], P[c->pos], x, T, D, c->cfg.vocab_size);
  if (int rc = dfh::check_launch("clip_embed_kernel")) return rc;
  const size_t hs_bytes = (size_t)M * D * sizeof(float);
  if (hidden_states && hipMemcpyAsync(hidden_states, x, hs_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    dfh::set_error("dfh_clip_encode: hidden_states copy failed");
    return -2;
  }
  for (size_t l = 0; l < c->layers.size(); ++l) {
    const ClipLayer& L = c->layers[l];
    hipLaunchKernelGGL(clip_layernorm_kernel, ln_grid, dim3(256), 0, s, x, P[L.ln1w], P[L.ln1b], ln, M, D, eps);
    if (int rc = dfh::check_launch("clip_layernorm_kernel")) return rc;
    if (int rc = clip_linear(ln, D, P[L.qw], D, P[L.qb], nullptr, 0, qkv, 3 * D, M, D, CLIP_ACT_NONE, s)) return rc;
    if (int rc = clip_linear(ln, D, P[L.kw], D, P[L.kb], nullptr, 0, qkv + D, 3 * D, M, D, CLIP_ACT_NONE, s)) return rc;
    if (int rc = clip_linear(ln, D, P[L.vw], D, P[L.vb], nullptr, 0, qkv + 2 * D, 3 * D, M, D, CLIP_ACT_NONE, s)) return rc;
    hipLaunchKernelGGL(clip_attention_kernel, dim3(H, batch), dim3(256), lds, s, qkv, att, T, D, d, scale);
    if (int rc = dfh::check_launch("clip_attention_kernel")) return rc;
    if (int rc = clip_linear(att, D, P[L.ow], D, P[L.ob], x, D, x, D, M, D, CLIP_ACT_NONE, s)) return rc;       // x += out_proj(attention)
    hipLaunchKernelGGL(clip_layernorm_kernel, ln_grid, dim3(256), 0, s, x, P[L.ln2w], P[L.ln2b], ln, M, D, eps);
    if (int rc = dfh::check_launch("clip_layernorm_kernel")) return rc;
    if (int rc = clip_linear(ln, D, P[L.f1w], D, P[L.f1b], nullptr, 0, hid, I, M, I, c->cfg.hidden_act, s)) return rc;
    if (int rc = clip_linear(hid, I, P[L.f2w], I, P[L.f2b], x, D, x, D, M, D, CLIP_ACT_NONE, s)) return rc;      // x += fc2(act(fc1(.)))
    if (hidden_states && hipMemcpyAsync(hidden_states + (l + 1) * (size_t)M * D, x, hs_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      dfh::set_error("dfh_clip_encode: hidden_states copy failed");
      return -2;
    }
  }
  hipLaunchKernelGGL(clip_layernorm_kernel, ln_grid, dim3(256), 0, s, x, P[c->fw], P[c->fb], last_hidden_state, M, D, eps);
  if (int rc = dfh::check_launch("clip_layernorm_kernel")) return rc;
  if (pooler_output) {
    hipLaunchKernelGGL(clip_pool_kernel, dim3(batch), dim3(256), 0, s, input_ids, last_hidden_state, pooler_output, T, D, eos_token_id);
    if (int rc = dfh::check_launch("clip_pool_kernel")) return rc;
  }
  return 0;
}

}  // extern "C"
