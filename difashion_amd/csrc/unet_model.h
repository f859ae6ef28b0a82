// Shared definition of the U-Net runtime object (parameter table, packing plan, inference walk).
// Included by unet.hip (inference + C ABI) and unet_train.hip (training forward / backward).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/difashion_hip.h"
#include "attention.h"
#include "dfh_common.h"
#include "elementwise.h"
#include "gemm.h"
#include "mlp_fused.h"
#ifdef DFH_PROBES
#include "token_linear.h"
#endif
#include "norm.h"
#include "packtab.h"
#include "bwd_elementwise.h"

namespace dfhm {

struct Mat { size_t off = 0; int N = 0, K = 0; };    // bf16 [N][K] at arena16 + off (elements)
struct Vec { size_t off = 0; int N = 0; };           // fp32 [N] at arena32 + off (elements)

enum PackKind { PK_VEC = 0, PK_MAT = 1, PK_CONV3 = 2 };
struct PackOp {
  int param, kind;
  size_t dst;
  int N, K, ldw, row_off, col_off, geglu, accumulate;
  int cin_pad = 0;   // PK_CONV3: channels per tap in the packed layout (conv_in pads 4 -> 8)
};

struct ParamDesc { std::string name; std::vector<int> shape; };

// transposed pack (training): master weight -> arena16t, see bwd_elementwise.hip pack_*_t kernels
struct TPackOp { int param, conv; size_t dst; int N, K, ldt, t_row_off, t_col_off, geglu, o_pad; };

// fp8 [N][K] at arena8 + off, per-row scales (floats) at arena8 + soff; boff: a derived bias (floats) for the matrices that fold a norm's
// affine into themselves (proj_in), else unused
struct Mat8 { size_t off = 0, soff = 0, boff = 0; int N = 0, K = 0; bool on = false; };

struct ResL {
  int cin = 0, cout = 0, temb_off = 0; bool shortcut = false;
  Vec n1w, n1b, b1, n2w, n2b, b2; Mat w1, w2;
  std::string pre; Mat w1t, w2t, wst;   // training: transposed packs for the data-gradient GEMMs
  // Winograd F(2x2, 3x3) weights U [16][cout][cin] of conv1 / conv2 in the fold region (winograd.hip), allocated for the deep levels
  size_t u1 = 0, u2 = 0; bool has_u = false;
};
// A LayerNorm-fed projection with the LayerNorm folded in (lnfold.hip): W' (bf16 elements) / s / b' (floats) offsets inside the fold
// region at the head of the workspace
struct Fold { size_t w = 0, s = 0, b = 0; int N = 0, K = 0; };

struct AttL {
  int C = 0, heads = 0, x_off = 0;   // x_off: this layer's row offset in the batched cross K / V matrices
  int idx = 0;                       // position in all_att() order (the per-layer slots of the fp8 path's V maxima)
  Fold fqkv, fqk, fv, fq2, fff1;
  // ff.net.2 and proj_out folded into ONE linear over [GEGLU output | h2] (inference walk): W = [pout . ff2 | pout] ([C][5C]), bias =
  // pout . ff2b + poutb -- proj_out(ff2(f) + ff2b + h2) + poutb as written, minus one launch, one bf16 rounding and one round trip
  // of a [tokens][C] tensor per transformer block
  Fold fffp;
  // C = 320 (the 64x64 level): the whole feed-forward + proj_out as ONE kernel (mlp_fused2.hip); its weight image (fragment-major LDS
  // image of fff1 and fffp, 2.7 MB) in the fold region
  size_t mlp_img = 0; bool has_mlp = false;
  // ... and its four K = N = C projections (proj_in, attn1.to_out, attn2.to_q folded, attn2.to_out) as register-resident token linears
  // (mlp_fused2.hip token_linear_kernel): their weight images in the fold region
  size_t tl_pin = 0, tl_o1 = 0, tl_q2 = 0, tl_o2 = 0; bool has_tl = false;
  Vec nw, nb, pinb, l1w, l1b, o1b, l2w, l2b, o2b, l3w, l3b, ff1b, ff2b, poutb;
  Mat pin, qk, v, o1, q2, o2, ff1, ff2, pout;
  std::string pre; Mat pint, qkvt, o1t, q2t, o2t, ff1t, ff2t, poutt;
  Mat8 qk8, v8, q28, ff18;           // fp8 copies of the LayerNorm-fed projections (gemm_fp8.hip), when enabled
  // round 4 (BASELINE configs[4] as named: "attention + 1x1-conv path"): to_out of both attentions, ff.net.2, proj_out, and proj_in with
  // the GroupNorm affine folded in (W . diag(gamma), bias + W . beta: the GroupNorm kernel then emits the normalised value as e4m3)
  Mat8 o18, o28, ff28, pout8, pin8;
  // fp8 attention products (attention_fp8.hip): operand factors rq | rk | rv ([C] each) and the per-head softmax factor ([heads]) as
  // floats at arena8 + f8a_off, derived from the LayerNorm-folded q | k | v weights at pack time; f8a: allocated for this layer
  size_t f8a_off = 0; bool f8a = false;
};
struct ConvL {
  Mat w; Vec b; int cin = 0, cout = 0; std::string pre; Mat wt;
  // upsamplers: summed phase weights [4][cout][4 * cin] in the fold region (gemm.h GemmArgs::phase2x), has_ph when allocated
  size_t ph = 0; bool has_ph = false;
};

struct Tensor {
  bf16_t* p = nullptr; int H = 0, W = 0, C = 0;
  // GroupNorm statistics of this tensor written by the GEMM epilogue that produced it (gemm.h GemmArgs::gstat); null when the
  // launch ran on a kernel that does not write them -- the consuming GroupNorm then computes its own
  const float* gst = nullptr; int gst_cpg = 0, gst_chunks = 0;
};

struct Bump {
  char* base = nullptr; size_t cap = 0, off = 0, peak = 0;
  void* alloc(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    char* p = base + off;
    off += bytes;
    if (off > peak) peak = off;
    return p;
  }
};

// host copy + device copy of a TabOp table; re-uploaded only when an entry (e.g. a master pointer) changed
struct OpTable {
  std::vector<TabOp> host, uploaded; TabOp* dev = nullptr; size_t cap = 0; unsigned blocks = 0;
  void clear() { host.clear(); blocks = 0; }
  void add(void* master, int kind, long dst, int N, int K, int ld, int p0, int p1, int p2, int p3, long /*elems*/) {
    TabOp op; std::memset(&op, 0, sizeof(op));
    op.master = master; op.dst = dst; op.kind = kind; op.N = N; op.K = K; op.ld = ld; op.p0 = p0; op.p1 = p1; op.p2 = p2; op.p3 = p3;
    op.first_block = blocks;
    blocks += dfh::tab_blocks(kind, N, K);
    host.push_back(op);
  }
  // a PACK2 op: the plain pack (dst .. p3 as in add) and the transposed pack (dst2, ld2, q0 = t_row_off, q1 = t_col_off, q3 = o_pad) of one master
  void add2(void* master, int kind, long dst, int N, int K, int ld, int p0, int p1, int p2, int p3, long dst2, int ld2, int q0, int q1, int q3) {
    add(master, kind, dst, N, K, ld, p0, p1, p2, p3, 0);
    TabOp& op = host.back();
    op.dst2 = dst2; op.ld2 = ld2; op.q0 = q0; op.q1 = q1; op.q3 = q3;
  }
  int launch(void* arena_vec, void* arena_mat, hipStream_t s, void* arena_mat2 = nullptr, float* sq_partials = nullptr) {
    if (host.empty()) return 0;
    const size_t bytes = host.size() * sizeof(TabOp);
    if (host.size() != uploaded.size() || std::memcmp(host.data(), uploaded.data(), bytes) != 0) {
      if (host.size() > cap) {
        if (dev) (void)hipFree(dev);
        if (hipMalloc((void**)&dev, bytes) != hipSuccess) { dfh::set_error("hipMalloc of an op table failed"); return -1; }
        cap = host.size();
      }
      // rare (first use / parameters re-homed): stream-ordered with respect to earlier launches that read the old table
      if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
        dfh::set_error("uploading an op table failed"); return -1;
      }
      uploaded = host;
    }
    return dfh::table_launch(dev, (int)host.size(), blocks, arena_vec, arena_mat, s, arena_mat2, sq_partials);
  }
  ~OpTable() { if (dev) (void)hipFree(dev); }
};

}  // namespace dfhm
using namespace dfhm;

struct dfh_unet {
  dfh_unet_config cfg{};
  std::vector<ParamDesc> params;
  std::vector<PackOp> packs;
  size_t a16 = 0, a32 = 0;         // arena sizes in elements
  // layers
  ConvL conv_in, conv_out;
  Mat te1, te2, tproj, kx_all, vx_all; Vec te1b, te2b, tprojb, cnw, cnb;
  int temb_total = 0, x_total = 0;   // x_total: sum of C over all transformer layers (batched text K/V)
  // folded-LayerNorm copies of the LayerNorm-fed projections (derived data, re-made by every pack): they live at the head of the
  // WORKSPACE, not in arena16 / arena32 -- the training path sizes its gradient arenas and its all-reduce by those
  size_t fold16 = 0, fold32 = 0; bool fold_valid = false, fold_dirty = false;
  size_t fold_bytes() const { return ((fold16 * 2 + 255) & ~(size_t)255) + ((fold32 * 4 + 255) & ~(size_t)255); }
  bf16_t* fold_w() const { return (bf16_t*)ws; }
  float* fold_v() const { return (float*)(ws + ((fold16 * 2 + 255) & ~(size_t)255)); }
  Fold fold_alloc(int N, int K) {
    Fold f; f.N = N; f.K = K;
    f.w = fold16; fold16 += ((size_t)N * K + 127) & ~(size_t)127;
    f.s = fold32; fold32 += (N + 63) & ~63;
    f.b = fold32; fold32 += (N + 63) & ~63;
    return f;
  }
  std::vector<std::vector<ResL>> down_res, up_res;
  std::vector<std::vector<AttL>> down_att, up_att;
  std::vector<ConvL> down_samp, up_samp;
  ResL mid_res[2]; AttL mid_att;
  // bound memory
  bf16_t* arena16 = nullptr; float* arena32 = nullptr;
  char* ws = nullptr; size_t ws_bytes = 0; int max_batch = 0;
  // planning results (bytes) for the last planned batch
  size_t plan_persist = 0, plan_temp = 0, plan_partial = 0, plan_total = 0; int plan_batch = 0;
  // fp8 linears (BASELINE configs[4]): e4m3 copies of qk / v / q2 / ff1 + per-row scales, in one caller-owned arena
  bool fp8 = false; unsigned char* arena8 = nullptr; size_t a8 = 0;
  // fp8 walk: also the self-attention products QK^T / PV on the e4m3 MFMA (attention_fp8.hip).  OFF by default: built, parity-tested and
  // measured slower than the bf16 kernels at every head dim of this model (profiles/r04/attn_fp8_microbench.txt)
  bool fp8_attention = false;
  // taps of the last forward
  std::map<std::string, Tensor> taps; int last_batch = 0;
  int dup_tail = 0;          // one-shot hint for the next forward (dfh_unet_set_dup_tail): trailing images that repeat the inputs of the ones before them
  // ---- training state (unet_train.hip)
  std::vector<TPackOp> tpacks; size_t a16t = 0; bool train_built = false;
  Mat te2t, tprojt;
  bf16_t* arena16t = nullptr; float* grad16 = nullptr; float* grad32 = nullptr;   // grad16/32: fp32, packed layouts of arena16/32
  char* tws = nullptr; size_t tws_bytes = 0; int train_max_batch = 0;
  size_t tplan_total = 0; int tplan_batch = 0;
  struct TrainRun; TrainRun* tr = nullptr;
  // backward walk: the weight-gradient GEMM of a layer runs on a second stream beside the data-gradient GEMM of the same layer
  // (both only read dY): the tail round of one is filled with blocks of the other (unet_train.hip TrainRun::wgrad / join)
  hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // dfh_unet_grad_sumsq: the un-pack of the backward also leaves the sum of the squares of every gradient value it wrote in *grad_sumsq_out
  // (the clip norm of the optimizer without another pass over 3.4 GB); per-block partials + a fixed-order reduce, no float atomics
  float* grad_sumsq_out = nullptr; float* sq_partials = nullptr; size_t sq_cap = 0; float* sq_scratch = nullptr;
  int build_train();
  size_t plan_train(int B);
  int forward_train(const void* sample, int sample_bf16, const float* timestep, const void* ehs, int ehs_bf16, float* out, int B,
                    hipStream_t s);
  int backward(const float* d_out, float* d_sample, float* const* master_grads, int count, hipStream_t s, int overwrite);
  // the same backward in pieces, for a data-parallel caller that all-reduces finished ranges of grad16 while the walk goes on
  int backward_begin(const float* d_out, float* d_sample, size_t bucket_floats, hipStream_t s);
  int backward_next(size_t* lo, size_t* hi, hipStream_t s);             // 1: grad16[lo, hi) is final; 0: tape done; < 0: error
  int backward_finish(float* const* master_grads, int count, hipStream_t s, int overwrite);
  int pack_train(const float* const* master, int count, hipStream_t s);
  ~dfh_unet();

  // ---------------------------------------------------------------- build
  int add_param(const std::string& name, std::vector<int> shape) {
    params.push_back({name, std::move(shape)});
    return (int)params.size() - 1;
  }
  size_t alloc16(size_t n) { size_t o = a16; a16 += (n + 127) & ~(size_t)127; return o; }
  size_t alloc32(size_t n) { size_t o = a32; a32 += (n + 63) & ~(size_t)63; return o; }

  Vec vec(const std::string& name, int N) {
    Vec v; v.N = N; v.off = alloc32(N);
    int p = add_param(name, {N});
    packs.push_back({p, PK_VEC, v.off, N, 0, 0, 0, 0, 0, 0});
    return v;
  }
  // packs a vector parameter into an existing fp32 range (batched biases / fused shortcut bias)
  void vec_into(const std::string& name, int N, size_t dst, int geglu, int accumulate) {
    int p = add_param(name, {N});
    packs.push_back({p, PK_VEC, dst, N, 0, 0, 0, 0, geglu, accumulate});
  }
  Mat mat_alloc(int N, int K) { Mat m; m.N = N; m.K = K; m.off = alloc16((size_t)N * K); return m; }
  void mat_into(const std::string& name, int N, int K, bool as_conv1x1, const Mat& dst, int row_off, int col_off, int geglu) {
    std::vector<int> shape = as_conv1x1 ? std::vector<int>{N, K, 1, 1} : std::vector<int>{N, K};
    int p = add_param(name, shape);
    packs.push_back({p, PK_MAT, dst.off, N, K, dst.K, row_off, col_off, geglu, 0});
  }
  Mat mat(const std::string& name, int N, int K, bool as_conv1x1 = false, int geglu = 0) {
    Mat m = mat_alloc(N, K);
    mat_into(name, N, K, as_conv1x1, m, 0, 0, geglu);
    return m;
  }
  void conv_into(const std::string& name, int cout, int cin, const Mat& dst, int col_off, int cin_pad = 0) {
    int p = add_param(name, {cout, cin, 3, 3});
    PackOp op{p, PK_CONV3, dst.off, cout, cin, dst.K, 0, col_off, 0, 0};
    op.cin_pad = cin_pad ? cin_pad : cin;
    packs.push_back(op);
  }

  // largest image (pixels) whose wide resnet convs take Winograd: 256 = the 16x16 level (default); DFH_WINO_MAXHW=1024 adds the 32x32 level
  // (the A/B of profiles/r04: measured, not the default)
  static int wino_max_hw() {
    static const int v = [] { const char* e = getenv("DFH_WINO_MAXHW"); return e ? atoi(e) : 256; }();
    return v;
  }
  // res: side of the (square) image this resnet runs on
  void build_resnet(const std::string& pre, int cin, int cout, ResL& r, int res) {
    const int temb = cfg.block_out_channels[0] * 4;
    // Winograd (winograd.hip) where it pays and where its bf16 transform-domain roundings are a small part of the layer's error budget:
    // the wide layers (>= 512 channels both ways) of the levels of at most 16 x 16 pixels
    if (res % 2 == 0 && res * res <= wino_max_hw() && cin % 8 == 0 && cout % 8 == 0 && std::min(cin, cout) >= 512) {
      r.has_u = true;
      r.u1 = fold16; fold16 += ((size_t)16 * cout * cin + 127) & ~(size_t)127;
      r.u2 = fold16; fold16 += ((size_t)16 * cout * cout + 127) & ~(size_t)127;
    }
    r.cin = cin; r.cout = cout; r.shortcut = cin != cout; r.pre = pre;
    r.n1w = vec(pre + ".norm1.weight", cin);
    r.n1b = vec(pre + ".norm1.bias", cin);
    r.w1 = mat_alloc(cout, 9 * cin);
    conv_into(pre + ".conv1.weight", cout, cin, r.w1, 0);
    r.b1 = vec(pre + ".conv1.bias", cout);
    r.temb_off = temb_total;
    temb_total += cout;
    // time_emb_proj rows are packed later into the batched matrix (needs the final total): remember via params
    add_param(pre + ".time_emb_proj.weight", {cout, temb});
    add_param(pre + ".time_emb_proj.bias", {cout});
    r.n2w = vec(pre + ".norm2.weight", cout);
    r.n2b = vec(pre + ".norm2.bias", cout);
    r.w2 = mat_alloc(cout, 9 * cout + (r.shortcut ? cin : 0));
    conv_into(pre + ".conv2.weight", cout, cout, r.w2, 0);
    r.b2 = vec(pre + ".conv2.bias", cout);
    if (r.shortcut) {
      mat_into(pre + ".conv_shortcut.weight", cout, cin, true, r.w2, 0, 9 * cout, 0);
      vec_into(pre + ".conv_shortcut.bias", cout, r.b2.off, 0, /*accumulate=*/1);
    }
  }

  void build_attn(const std::string& pre, int C, int heads, AttL& a) {
    const bool lin = cfg.use_linear_projection != 0;
    const int X = cfg.cross_attention_dim;
    a.C = C; a.heads = heads; a.pre = pre;
    a.nw = vec(pre + ".norm.weight", C);
    a.nb = vec(pre + ".norm.bias", C);
    a.pin = mat(pre + ".proj_in.weight", C, C, !lin);
    a.pinb = vec(pre + ".proj_in.bias", C);
    const std::string tb = pre + ".transformer_blocks.0";
    a.l1w = vec(tb + ".norm1.weight", C); a.l1b = vec(tb + ".norm1.bias", C);
    a.l2w = vec(tb + ".norm2.weight", C); a.l2b = vec(tb + ".norm2.bias", C);
    a.l3w = vec(tb + ".norm3.weight", C); a.l3b = vec(tb + ".norm3.bias", C);
    a.qk = mat_alloc(2 * C, C);
    mat_into(tb + ".attn1.to_q.weight", C, C, false, a.qk, 0, 0, 0);
    mat_into(tb + ".attn1.to_k.weight", C, C, false, a.qk, C, 0, 0);
    a.v = mat(tb + ".attn1.to_v.weight", C, C);
    a.o1 = mat(tb + ".attn1.to_out.0.weight", C, C);
    a.o1b = vec(tb + ".attn1.to_out.0.bias", C);
    a.q2 = mat(tb + ".attn2.to_q.weight", C, C);
    a.x_off = x_total;            // to_k / to_v of every layer are packed into two stacked matrices (below)
    x_total += C;
    add_param(tb + ".attn2.to_k.weight", {C, X});
    add_param(tb + ".attn2.to_v.weight", {C, X});
    a.o2 = mat(tb + ".attn2.to_out.0.weight", C, C);
    a.o2b = vec(tb + ".attn2.to_out.0.bias", C);
    a.ff1 = mat(tb + ".ff.net.0.proj.weight", 8 * C, C, false, /*geglu=*/1);
    a.ff1b.N = 8 * C; a.ff1b.off = alloc32(8 * C);
    vec_into(tb + ".ff.net.0.proj.bias", 8 * C, a.ff1b.off, 1, 0);
    a.ff2 = mat(tb + ".ff.net.2.weight", C, 4 * C);
    a.ff2b = vec(tb + ".ff.net.2.bias", C);
    a.pout = mat(pre + ".proj_out.weight", C, C, !lin);
    a.poutb = vec(pre + ".proj_out.bias", C);
    // q | k and v share one folded matrix [3C][C] (and one s / b' vector): ONE launch writes q | k and V^T (GemmArgs::out2); fqk / fv
    // are views of it for the two-launch fallback
    a.fqkv = fold_alloc(3 * C, C);
    a.fqk = a.fqkv; a.fqk.N = 2 * C;
    a.fv = a.fqkv; a.fv.N = C; a.fv.w += (size_t)2 * C * C; a.fv.s += 2 * C; a.fv.b += 2 * C;
    a.fq2 = fold_alloc(C, C); a.fff1 = fold_alloc(8 * C, C);
    a.fffp = fold_alloc(C, 5 * C);
    if (dfh::mlp_fused_eligible(C, 128)) {
      a.has_mlp = true; a.mlp_img = fold16;
      fold16 += (dfh::mlp_fused_image_bytes() / 2 + 127) & ~(size_t)127;
    }
#ifdef DFH_PROBES
    if (dfh::token_linear_eligible(C, C, 128)) {
      a.has_tl = true;
      for (size_t* o : {&a.tl_pin, &a.tl_o1, &a.tl_q2, &a.tl_o2}) { *o = fold16; fold16 += (dfh::token_linear_image_bytes() / 2 + 127) & ~(size_t)127; }
    }
#endif
  }

  void build_conv(const std::string& pre, int cout, int cin, ConvL& c) {
    // conv_out has 4 output channels; conv_in 8 (or 4, padded to 8 with zero weights) input channels:
    // both go through the same GEMM
    const int cp = (cin + 7) & ~7;
    c.cin = cp; c.cout = cout; c.pre = pre;
    c.w = mat_alloc(cout, 9 * cp);
    conv_into(pre + ".weight", cout, cin, c.w, 0, cp);
    c.b = vec(pre + ".bias", cout);
  }

  int build() {
    const int nb = cfg.num_blocks;
    const int* boc = cfg.block_out_channels;
    const int temb = boc[0] * 4;
    build_conv("conv_in", boc[0], cfg.in_channels, conv_in);
    te1 = mat("time_embedding.linear_1.weight", temb, boc[0]);
    te1b = vec("time_embedding.linear_1.bias", temb);
    te2 = mat("time_embedding.linear_2.weight", temb, temb);
    te2b = vec("time_embedding.linear_2.bias", temb);
    down_res.resize(nb); down_att.resize(nb); down_samp.resize(nb);
    up_res.resize(nb); up_att.resize(nb); up_samp.resize(nb);
    int ch = boc[0];
    for (int i = 0; i < nb; ++i) {
      const int oc = boc[i];
      down_res[i].resize(cfg.layers_per_block);
      if (cfg.down_attn[i]) down_att[i].resize(cfg.layers_per_block);
      for (int j = 0; j < cfg.layers_per_block; ++j) {
        const std::string b = "down_blocks." + std::to_string(i);
        build_resnet(b + ".resnets." + std::to_string(j), j == 0 ? ch : oc, oc, down_res[i][j], cfg.sample_size >> i);
      }
      for (int j = 0; j < cfg.layers_per_block && cfg.down_attn[i]; ++j)
        build_attn("down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), oc, cfg.num_heads[i], down_att[i][j]);
      if (i != nb - 1) build_conv("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", oc, oc, down_samp[i]);
      ch = oc;
    }
    const int mid = boc[nb - 1];
    build_resnet("mid_block.resnets.0", mid, mid, mid_res[0], cfg.sample_size >> (nb - 1));
    build_attn("mid_block.attentions.0", mid, cfg.num_heads[nb - 1], mid_att);
    build_resnet("mid_block.resnets.1", mid, mid, mid_res[1], cfg.sample_size >> (nb - 1));
    int out_ch = boc[nb - 1];
    for (int i = 0; i < nb; ++i) {
      const int prev = out_ch;
      out_ch = boc[nb - 1 - i];
      const int in_ch = boc[nb - 1 - std::min(i + 1, nb - 1)];
      const bool att = cfg.down_attn[nb - 1 - i] != 0;
      const int L = cfg.layers_per_block + 1;
      up_res[i].resize(L);
      if (att) up_att[i].resize(L);
      const std::string b = "up_blocks." + std::to_string(i);
      for (int j = 0; j < L; ++j) {
        const int skip = (j == L - 1) ? in_ch : out_ch;
        const int hid = (j == 0) ? prev : out_ch;
        build_resnet(b + ".resnets." + std::to_string(j), hid + skip, out_ch, up_res[i][j], cfg.sample_size >> (nb - 1 - i));
      }
      for (int j = 0; j < L && att; ++j)
        build_attn(b + ".attentions." + std::to_string(j), out_ch, cfg.num_heads[nb - 1 - i], up_att[i][j]);
      if (i != nb - 1) {
        build_conv(b + ".upsamplers.0.conv", out_ch, out_ch, up_samp[i]);
        if (out_ch % 8 == 0) {
          up_samp[i].ph = fold16; up_samp[i].has_ph = true;
          fold16 += ((size_t)16 * out_ch * out_ch + 127) & ~(size_t)127;
        }
      }
    }
    cnw = vec("conv_norm_out.weight", boc[0]);
    cnb = vec("conv_norm_out.bias", boc[0]);
    build_conv("conv_out", cfg.out_channels, boc[0], conv_out);
    // batched time_emb_proj: [temb_total][temb] + bias; rows of each resnet at its temb_off
    tproj = mat_alloc(temb_total, temb);
    tprojb.N = temb_total; tprojb.off = alloc32(temb_total);
    // batched cross-attention K / V projections of the text states: [x_total][cross_dim] each
    kx_all = mat_alloc(x_total, cfg.cross_attention_dim);
    vx_all = mat_alloc(x_total, cfg.cross_attention_dim);
    auto pack_cross = [&](const AttL& a, const std::string& pre) {
      const std::string tb = pre + ".transformer_blocks.0";
      for (int p = 0; p < (int)params.size(); ++p) {
        if (params[p].name == tb + ".attn2.to_k.weight")
          packs.push_back({p, PK_MAT, kx_all.off, a.C, cfg.cross_attention_dim, cfg.cross_attention_dim, a.x_off, 0, 0, 0});
        else if (params[p].name == tb + ".attn2.to_v.weight")
          packs.push_back({p, PK_MAT, vx_all.off, a.C, cfg.cross_attention_dim, cfg.cross_attention_dim, a.x_off, 0, 0, 0});
      }
    };
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j < (int)down_att[i].size(); ++j)
        pack_cross(down_att[i][j], "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j));
    pack_cross(mid_att, "mid_block.attentions.0");
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j < (int)up_att[i].size(); ++j)
        pack_cross(up_att[i][j], "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j));
    auto pack_tproj = [&](const ResL& r, const std::string& pre) {
      for (int p = 0; p < (int)params.size(); ++p) {
        if (params[p].name == pre + ".time_emb_proj.weight")
          packs.push_back({p, PK_MAT, tproj.off, r.cout, temb, temb, r.temb_off, 0, 0, 0});
        else if (params[p].name == pre + ".time_emb_proj.bias")
          packs.push_back({p, PK_VEC, tprojb.off + (size_t)r.temb_off, r.cout, 0, 0, 0, 0, 0, 0});
      }
    };
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j < (int)down_res[i].size(); ++j)
        pack_tproj(down_res[i][j], "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j));
    pack_tproj(mid_res[0], "mid_block.resnets.0");
    pack_tproj(mid_res[1], "mid_block.resnets.1");
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j < (int)up_res[i].size(); ++j)
        pack_tproj(up_res[i][j], "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j));
    return 0;
  }

  // ---------------------------------------------------------------- fp8
  std::vector<AttL*> all_att() {
    std::vector<AttL*> v;
    for (auto& lv : down_att) for (auto& a : lv) v.push_back(&a);
    v.push_back(&mid_att);
    for (auto& lv : up_att) for (auto& a : lv) v.push_back(&a);
    return v;
  }
  // fp8 copies exist for the transformer layers whose width the 64-deep contraction divides
  static constexpr float GN_Z = 32.0f;   // |normalised GroupNorm value| representable in the e4m3 proj_in operand (static scale 448 / GN_Z)
  size_t a8_slab_off = 0; int n_att = 0; std::vector<int> slab_host;
  int enable_fp8() {
    if (fp8) return 0;
    a8 = 0;
    auto take = [&](const Mat& m, Mat8& q, bool with_bias = false) {
      q.N = m.N; q.K = m.K; q.on = true;
      q.off = a8; a8 += ((size_t)m.N * m.K + 255) & ~(size_t)255;
      q.soff = a8; a8 += ((size_t)m.N * sizeof(float) + 255) & ~(size_t)255;
      if (with_bias) { q.boff = a8; a8 += ((size_t)m.N * sizeof(float) + 255) & ~(size_t)255; }
    };
    // DFH_FP8_EXT=0: only the round-2 set (the LayerNorm-fed projections) -- A/B switch
    static const bool ext_off = [] { const char* e = getenv("DFH_FP8_EXT"); return e && e[0] == '0'; }();
    int idx = 0;
    for (AttL* a : all_att()) {
      a->idx = idx++;
      if (a->C % 64) continue;
      take(a->qk, a->qk8); take(a->v, a->v8); take(a->q2, a->q28); take(a->ff1, a->ff18);
      if (ext_off) continue;
      take(a->o1, a->o18); take(a->o2, a->o28); take(a->ff2, a->ff28); take(a->pout, a->pout8); take(a->pin, a->pin8, true);
      // opt-in (dfh_unet_enable_fp8_attention / DFH_FP8_ATTN=1): see fp8_attention above
      static const bool attn_env = [] { const char* e = getenv("DFH_FP8_ATTN"); return e && e[0] == '1'; }();
      const bool attn_off = !(fp8_attention || attn_env);
      const int D = a->C / a->heads;
      const bool v_contig = a->v.off == a->qk.off + (size_t)2 * a->C * a->C && a->v.K == a->qk.K;
      if (!attn_off && v_contig && (D == 40 || D == 80 || D == 160)) {
        a->f8a = true; a->f8a_off = a8; a8 += ((size_t)(3 * a->C + a->heads) * sizeof(float) + 255) & ~(size_t)255;
      }
    }
    n_att = idx;
    // (row offset, row count) of every layer's slice of the batched cross-attention V^T: the slabs of amax_slabs_kernel
    a8_slab_off = a8; a8 += ((size_t)2 * n_att * sizeof(int) + 255) & ~(size_t)255;
    slab_host.assign(2 * n_att, 0);
    for (AttL* a : all_att()) { slab_host[a->idx] = a->x_off; slab_host[n_att + a->idx] = a->C; }
    fp8 = true;
    return 0;
  }
  const int* slab_row0() const { return (const int*)(arena8 + a8_slab_off); }
  const int* slab_rows() const { return slab_row0() + n_att; }
  int quantize_fp8(hipStream_t s) {
    if (!fp8 || !arena8) return 0;
    if (hipMemcpyAsync(arena8 + a8_slab_off, slab_host.data(), slab_host.size() * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) {
      dfh::set_error("uploading the V^T slab table failed"); return -2;
    }
    for (AttL* a : all_att()) {
      const Mat* src[8] = {&a->qk, &a->v, &a->q2, &a->ff1, &a->o1, &a->o2, &a->ff2, &a->pout};
      const Mat8* dst[8] = {&a->qk8, &a->v8, &a->q28, &a->ff18, &a->o18, &a->o28, &a->ff28, &a->pout8};
      for (int i = 0; i < 8; ++i) {
        if (!dst[i]->on) continue;
        if (int rc = dfh::quant_rows_fp8_launch(arena16 + src[i]->off, src[i]->K, arena8 + dst[i]->off, (float*)(arena8 + dst[i]->soff),
                                                src[i]->N, src[i]->K, s)) return rc;
      }
      if (a->pin8.on) {
        // proj_in behind the GroupNorm whose kernel emits the un-affined normalised value: W' = W . diag(gamma) (bf16, in the activation
        // workspace, which no walk is using while weights are derived), b' = bias + W . beta, then the per-channel quantisation of W'
        const int C = a->C;
        DFH_REQUIRE(ws && fold_bytes() + (size_t)C * C * 2 + (size_t)C * 4 + 512 <= ws_bytes, "workspace too small for the fp8 proj_in fold");
        bf16_t* wf = (bf16_t*)(ws + fold_bytes());
        float* sv = (float*)(ws + fold_bytes() + (((size_t)C * C * 2 + 255) & ~(size_t)255));
        if (int rc = dfh::ln_fold_launch(arena16 + a->pin.off, a->pin.K, arena32 + a->nw.off, arena32 + a->nb.off, arena32 + a->pinb.off, wf, sv,
                                         (float*)(arena8 + a->pin8.boff), C, C, s)) return rc;
        if (int rc = dfh::quant_rows_fp8_launch(wf, C, arena8 + a->pin8.off, (float*)(arena8 + a->pin8.soff), C, C, s)) return rc;
      }
      if (a->f8a) {
        // operand factors of the fp8 attention: bounds of q, k, v behind LayerNorm 1 from W . diag(gamma) and W . beta (bf16 / fp32 scratch)
        const int C = a->C;
        DFH_REQUIRE(ws && fold_bytes() + (size_t)3 * C * C * 2 + (size_t)6 * C * 4 + 1024 <= ws_bytes, "workspace too small for the fp8 attention scales");
        bf16_t* wf = (bf16_t*)(ws + fold_bytes());
        float* sv = (float*)(ws + fold_bytes() + (((size_t)3 * C * C * 2 + 255) & ~(size_t)255));
        float* bv = sv + 3 * C;
        if (int rc = dfh::ln_fold_launch(arena16 + a->qk.off, C, arena32 + a->l1w.off, arena32 + a->l1b.off, nullptr, wf, sv, bv, 3 * C, C, s)) return rc;
        float* f = (float*)(arena8 + a->f8a_off);
        if (int rc = dfh::attn_scales_launch(wf, bv, C, a->heads, f, f + C, f + 2 * C, f + 3 * C, s)) return rc;
      }
    }
    return 0;
  }

  // W' / s / b' of every LayerNorm-fed projection from the freshly packed bf16 matrices (needs the workspace: after dfh_unet_bind)
  int fold_layernorms(hipStream_t s) {
    fold_valid = false;
    if (!ws || !arena16 || !arena32) return 0;
    // a transformer width the 16-byte kernels cannot take (C % 8 != 0) has no folded weights: the walk must not read its (unwritten)
    // fold slots, so the whole inference walk then stays on the unfolded path (fold_valid stays false)
    for (AttL* a : all_att()) if (a->C % 8) return 0;
    for (AttL* a : all_att()) {
      if (fp8 && a->qk8.on) continue;                 // the fp8 walk of this layer reads none of the folded bf16 copies: not derived
      const Mat* src[4] = {&a->qk, &a->v, &a->q2, &a->ff1};
      const Fold* dst[4] = {&a->fqk, &a->fv, &a->fq2, &a->fff1};
      const Vec* gam[4] = {&a->l1w, &a->l1w, &a->l2w, &a->l3w};
      const Vec* bet[4] = {&a->l1b, &a->l1b, &a->l2b, &a->l3b};
      for (int i = 0; i < 4; ++i) {
        const float* bias = i == 3 ? arena32 + a->ff1b.off : nullptr;
        if (int rc = dfh::ln_fold_launch(arena16 + src[i]->off, src[i]->K, arena32 + gam[i]->off, arena32 + bet[i]->off, bias,
                                         fold_w() + dst[i]->w, fold_v() + dst[i]->s, fold_v() + dst[i]->b, src[i]->N, src[i]->K, s)) return rc;
      }
    }
#ifdef DFH_PROBES
    for (AttL* a : all_att()) {
      static const bool tl_on = [] { const char* e = getenv("DFH_TOKEN_LINEAR"); return e && e[0] == '1'; }();
      if (!tl_on || !a->has_tl || (fp8 && a->qk8.on)) continue;
      const int C = a->C;
      if (int rc = dfh::token_linear_pack_launch(arena16 + a->pin.off, a->pin.K, fold_w() + a->tl_pin, s)) return rc;
      if (int rc = dfh::token_linear_pack_launch(arena16 + a->o1.off, a->o1.K, fold_w() + a->tl_o1, s)) return rc;
      if (int rc = dfh::token_linear_pack_launch(fold_w() + a->fq2.w, C, fold_w() + a->tl_q2, s)) return rc;
      if (int rc = dfh::token_linear_pack_launch(arena16 + a->o2.off, a->o2.K, fold_w() + a->tl_o2, s)) return rc;
    }
#endif
    {
      std::vector<ResL*> rs;
      for (auto& lv : down_res) for (auto& r : lv) rs.push_back(&r);
      rs.push_back(&mid_res[0]); rs.push_back(&mid_res[1]);
      for (auto& lv : up_res) for (auto& r : lv) rs.push_back(&r);
      for (ResL* r : rs) {
        if (!r->has_u) continue;
        if (int rc = dfh::wino_weight_launch(arena16 + r->w1.off, r->w1.K, fold_w() + r->u1, r->cout, r->cin, dfh::wino_blocked(r->cout, r->cin), s)) return rc;
        if (int rc = dfh::wino_weight_launch(arena16 + r->w2.off, r->w2.K, fold_w() + r->u2, r->cout, r->cout, dfh::wino_blocked(r->cout, r->cout), s)) return rc;
      }
    }
    for (ConvL& c : up_samp)
      if (c.has_ph)
        if (int rc = dfh::ups_phase_fold_launch(arena16 + c.w.off, c.w.K, fold_w() + c.ph, c.cout, c.cin, s)) return rc;
    // ff.net.2 . proj_out: [pout . ff2 | pout] and its bias.  The product runs on the GEMM kernel itself (A = pout [C][C], the W operand
    // = ff2^T [4C][C], transposed into the activation workspace, which no walk is using while the weights are being derived)
    bf16_t* scratch = (bf16_t*)(ws + fold_bytes());
    bf16_t* zero = scratch;                                   // 256 zero bytes, then the transposed matrix
    if (hipMemsetAsync(zero, 0, 256, s) != hipSuccess) { dfh::set_error("hipMemsetAsync failed"); return -2; }
    for (AttL* a : all_att()) {
      const int C = a->C;
      if (C % 8) continue;
      if (fp8 && a->pout8.on) continue;               // fp8 walk: ff.net.2 and proj_out are two e4m3 launches, the folded matrix is unused
      DFH_REQUIRE(fold_bytes() + 256 + (size_t)4 * C * C * 2 <= ws_bytes, "workspace too small for the weight-fold scratch");
      bf16_t* w2t = scratch + 128;
      if (int rc = dfh::transpose_bf16_launch(arena16 + a->ff2.off, w2t, 1, C, 4 * C, 4 * C, C, 0, 0, s)) return rc;
      GemmArgs g; std::memset(&g, 0, sizeof(g));
      g.M = C; g.N = 4 * C; g.rows_per_b = C;
      g.p_src[0] = arena16 + a->pout.off; g.p_c[0] = C; g.nplain = 1;
      g.W = w2t; g.ldw = C; g.zero = zero;
      g.out = fold_w() + a->fffp.w; g.ld_out = 5 * C; g.out_mode = OUT_BF16;
      if (int rc = dfh::gemm_launch(g, s, 0, /*force_split=*/1)) return rc;
      if (hipMemcpy2DAsync(fold_w() + a->fffp.w + 4 * C, (size_t)5 * C * 2, arena16 + a->pout.off, (size_t)C * 2, (size_t)C * 2, C,
                           hipMemcpyDeviceToDevice, s) != hipSuccess) { dfh::set_error("hipMemcpy2DAsync failed"); return -2; }
      if (int rc = dfh::matvec_bias_launch(arena16 + a->pout.off, C, arena32 + a->ff2b.off, arena32 + a->poutb.off,
                                           fold_v() + a->fffp.b, C, C, s)) return rc;
      if (a->has_mlp && dfh::mlp_fused_form() > 0) {     // the fused feed-forward's weight image from the two folded matrices just derived
#ifdef DFH_PROBES
        auto pack = dfh::mlp_fused_form() == 1 ? dfh::mlp_pack_launch : dfh::mlp2_pack_launch;
#else
        auto pack = dfh::mlp2_pack_launch;
#endif
        if (int rc = pack(fold_w() + a->fff1.w, fold_v() + a->fff1.s, fold_v() + a->fff1.b, fold_w() + a->fffp.w, fold_w() + a->mlp_img, s)) return rc;
      }
    }
    fold_valid = true; fold_dirty = false;
    return 0;
  }

  // ---------------------------------------------------------------- run
  struct Run {
    dfh_unet* u; int B; hipStream_t s; bool dry;
    Bump persist, temp; size_t partial_need = 0;
    float* partial = nullptr; size_t partial_cap = 0;
    float* gn_partial = nullptr; bf16_t* zero = nullptr;
    int temb_ld = 0;                 // row stride of the time-embedding rows: temb_total, or 0 when the whole batch shares one cached row
    int rc = 0;
    // fp8 walk: per transformer layer and batch element the largest |V| of the self-attention (tracked by the V projection's epilogue,
    // zeroed at the start of the walk) and of the cross-attention (amax_slabs over the text V^T, once per forward or per run): [n_att][B]
    float* amax_self = nullptr; const float* amax_cross = nullptr;

    bf16_t* w16(const Mat& m) const { return u->arena16 + m.off; }
    float* v32(const Vec& v) const { return u->arena32 + v.off; }
    // Ba: the batch every tensor is ALLOCATED for (the call's batch); B: the batch the launches run on.  They differ only inside the
    // shared prefix of a guidance batch whose last `dup` images repeat the inputs of the `dup` images before them (dfh_unet::dup_tail):
    // there B = Ba - dup, and dup_images() then copies the repeated images' rows into place.
    int Ba = 0;
    Tensor palloc(int H, int W, int C) { return Tensor{(bf16_t*)persist.alloc((size_t)Ba * H * W * C * 2), H, W, C}; }
    Tensor talloc(int H, int W, int C) { return Tensor{(bf16_t*)temp.alloc((size_t)Ba * H * W * C * 2), H, W, C}; }
    void dup_bytes(void* p, size_t per_image, int n) {   // images [Ba - n, Ba) := images [Ba - 2n, Ba - n) of a [Ba][per_image bytes] buffer
      if (rc || dry || n <= 0) return;
      char* c = (char*)p;
      if (hipMemcpyAsync(c + (size_t)(Ba - n) * per_image, c + (size_t)(Ba - 2 * n) * per_image, (size_t)n * per_image, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        dfh::set_error("hipMemcpyAsync failed (dup_bytes)"); rc = -2;
      }
    }
    void dup_images(Tensor& t, int n) {               // the same for a tensor; its producer statistics no longer cover it
      t.gst = nullptr;
      dup_bytes(t.p, (size_t)t.H * t.W * t.C * 2, n);
    }

    // o / bump: the output tensor and the allocator it came from when the output feeds a GroupNorm -- the epilogue then leaves
    // that GroupNorm's statistics beside it (64x64 level: the launches the 256 x 160 tile takes).  DFH_GN_PRE=0 turns it off (A/B).
    // rs_bn (out): column tile of the row statistics the launch wrote into g.rowstat (0 = none)
    void gemm(GemmArgs g, Tensor* o = nullptr, Bump* bump = nullptr, int* rs_bn = nullptr) {
      if (rs_bn) *rs_bn = 0;
      if (rc) return;
      g.zero = zero; g.partial = partial;
      static const bool pre_off = [] { const char* e = getenv("DFH_GN_PRE"); return e && e[0] == '0'; }();
      float* gst = nullptr;
      const int G = u->cfg.norm_num_groups;
      // the consumers take at most GN_MAX_CHUNKS chunks per (image, group): a level fits when its 256-row chunk count does; gemm_launch
      // refuses the 128-row writer by itself when HW / 128 would exceed it (96x96 latents: 36 chunks of 256 rows, 72 of 128)
      if (o && bump && !pre_off && (o->H * o->W) % 128 == 0 && o->C % G == 0 &&
          (o->H * o->W) / ((o->H * o->W) % 256 == 0 ? 256 : 128) <= (int)GN_MAX_CHUNKS) {
        gst = (float*)bump->alloc((size_t)Ba * G * ((o->H * o->W) / 128) * 2 * sizeof(float));      // same in the dry run; chunks of 256 or 128 pixel rows
        g.gstat = gst; g.gstat_cpg = o->C / G; g.gstat_hw = o->H * o->W;
      }
      if (dry) { partial_need = std::max(partial_need, dfh::gemm_partial_floats(g) * sizeof(float)); return; }
      if (dfh::gemm_partial_floats(g) * sizeof(float) > partial_cap) { dfh::set_error("split-K partial buffer too small"); rc = -1; return; }
      int gst_rows = 0;
      rc = dfh::gemm_launch(g, s, 0, 0, -1, &gst_rows, rs_bn);
      if (o && gst_rows) { o->gst = gst; o->gst_cpg = g.gstat_cpg; o->gst_chunks = g.gstat_hw / gst_rows; }
    }
    static GemmArgs base(int M, int N) {
      GemmArgs g; std::memset(&g, 0, sizeof(g));
      g.M = M; g.N = N; g.rows_per_b = M; g.out_mode = OUT_BF16; g.ld_out = N;
      return g;
    }
    // out = act(x . W^T + bias) (+resid); x rows [M][K]
    // rowstat / rs_bn: ask the launch for the per-row statistics of its output (a LayerNorm folded into the consumers, gemm.h)
    void linear(const bf16_t* x, int M, int K, const Mat& W, const Vec* bias, int act, const bf16_t* resid, void* out,
                int N, int out_mode = OUT_BF16, int ld_out = -1, int rows_per_b = 0, Tensor* o = nullptr, Bump* bump = nullptr,
                float* rowstat = nullptr, int* rs_bn = nullptr) {
      GemmArgs g = base(M, N);
      g.p_src[0] = x; g.p_c[0] = K; g.nplain = 1;
      g.W = w16(W); g.ldw = W.K;
      g.bias = bias ? v32(*bias) : nullptr;
      g.act = act; g.resid = resid; g.ld_res = N;
      g.out = out; g.out_mode = out_mode; g.ld_out = ld_out < 0 ? (act == ACT_GEGLU ? N / 2 : N) : ld_out;
      if (rows_per_b) g.rows_per_b = rows_per_b;
      g.rowstat = rowstat;
      gemm(g, o, bump, rs_bn);
    }
    // the consumer of a folded LayerNorm: raw rows x [M][K] (K = C of the LayerNorm), statistics st ([C / st_bn][M][2]) from x's producer
    GemmArgs folded(const bf16_t* x, int M, const Fold& f, const float* st, int st_bn, int act, void* out, int out_mode, int ld_out,
                    int rows_per_b) const {
      GemmArgs g = base(M, f.N);
      g.p_src[0] = x; g.p_c[0] = f.K; g.nplain = 1;
      g.W = u->fold_w() + f.w; g.ldw = f.K;
      g.bias = u->fold_v() + f.b; g.ln_s = u->fold_v() + f.s;
      g.ln_stat = st; g.ln_cnt = st_bn; g.ln_parts = st_bn > 0 ? f.K / st_bn : 0; g.ln_eps = 1e-5f;
      g.act = act; g.out = out; g.out_mode = out_mode; g.ld_out = ld_out < 0 ? (act == ACT_GEGLU ? f.N / 2 : f.N) : ld_out;
      if (rows_per_b) g.rows_per_b = rows_per_b;
      return g;
    }
    void groupnorm(const Tensor& x0, const Tensor* x1, const Vec& w, const Vec& b, float eps, int silu, Tensor& out) {
      if (rc || dry) return;
      GnArgs a; std::memset(&a, 0, sizeof(a));
      a.src0 = x0.p; a.C0 = x0.C; a.src1 = x1 ? x1->p : nullptr; a.C1 = x1 ? x1->C : 0;
      a.B = B; a.HW = x0.H * x0.W; a.G = u->cfg.norm_num_groups;
      a.gamma = v32(w); a.beta = v32(b); a.eps = eps; a.silu = silu; a.out = out.p; a.partial = gn_partial;
      if (!x1 && x0.gst && x0.gst_cpg == x0.C / a.G) { a.pre = x0.gst; a.pre_chunks = x0.gst_chunks; }   // summed by its producer
      rc = dfh::groupnorm_launch(a, s);
    }
    bool use8(const Mat8& m) const { return u->fp8 && m.on; }        // the forward entry checks that arena8 is bound
    // LayerNorm whose output is quantised per token + the fp8 GEMM that consumes it (gemm_fp8.hip)
    void layernorm8(const bf16_t* x, const Vec& w, const Vec& b, uint8_t* q, float* sc, int M, int C) {
      if (rc || dry) return;
      rc = dfh::layernorm_fp8_launch(x, v32(w), v32(b), q, sc, M, C, 1e-5f, s);
    }
    void linear8(const uint8_t* q, const float* sc, int M, const Mat8& W, const Vec* bias, int act, void* out, int out_mode = OUT_BF16,
                 int ld_out = -1, int rows_per_b = 0, float* amax = nullptr) {
      if (rc || dry) return;
      Fp8GemmArgs g; std::memset(&g, 0, sizeof(g));
      g.A = q; g.sA = sc; g.W = u->arena8 + W.off; g.sW = (const float*)(u->arena8 + W.soff);
      g.M = M; g.N = W.N; g.K = W.K; g.bias = bias ? v32(*bias) : nullptr; g.act = act;
      g.out = out; g.out_mode = out_mode; g.ld_out = ld_out < 0 ? (act == ACT_GEGLU ? W.N / 2 : W.N) : ld_out;
      g.rows_per_b = rows_per_b; g.zero = (const uint8_t*)zero; g.amax = amax;
      rc = dfh::gemm_fp8_launch(g, s);
    }
    // ---- fp8 walk (round 4): every operand of these launches is e4m3
    Fp8GemmArgs args8(const uint8_t* A, int M, const Mat8& W, const Vec* bias, void* out) const {
      Fp8GemmArgs g; std::memset(&g, 0, sizeof(g));
      g.A = A; g.W = u->arena8 + W.off; g.sW = (const float*)(u->arena8 + W.soff);
      g.M = M; g.N = W.N; g.K = W.K; g.bias = bias ? v32(*bias) : nullptr;
      g.out = out; g.out_mode = OUT_BF16; g.ld_out = W.N; g.zero = (const uint8_t*)zero;
      return g;
    }
    void gemm8(const Fp8GemmArgs& g) {
      if (rc || dry) return;
      rc = dfh::gemm_fp8_launch(g, s);
    }
    // GroupNorm without its affine, as e4m3 under the static scale 448 / GN_Z (the consumer's weights carry gamma)
    void groupnorm8(const Tensor& x, float eps, uint8_t* q) {
      if (rc || dry) return;
      GnArgs a; std::memset(&a, 0, sizeof(a));
      a.src0 = x.p; a.C0 = x.C; a.B = B; a.HW = x.H * x.W; a.G = u->cfg.norm_num_groups; a.eps = eps; a.partial = gn_partial;
      a.out8 = q; a.q_mul = 448.0f / dfh_unet::GN_Z;
      if (x.gst && x.gst_cpg == x.C / a.G) { a.pre = x.gst; a.pre_chunks = x.gst_chunks; }
      rc = dfh::groupnorm_launch(a, s);
    }
    void attention8(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const bf16_t* Vt, int ldvt, uint8_t* O8, const float* amax, int C,
                    int heads, int Nq, int Nk, long vt_bstride = 0, const float* f8 = nullptr) {
      if (rc || dry) return;
      AttnArgs a; std::memset(&a, 0, sizeof(a));
      if (f8) { a.f8_rq = f8; a.f8_rk = f8 + C; a.f8_rv = f8 + 2 * C; a.f8_hs = f8 + 3 * C; }
      a.vt_bstride = vt_bstride;
      a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.Vt = Vt; a.ldvt = ldvt; a.O8 = O8; a.o_amax = amax; a.ldo = C;
      a.B = B; a.H = heads; a.D = C / heads; a.Nq = Nq; a.Nk = Nk;
      a.scale = 1.0f / sqrtf((float)a.D);
      rc = dfh::attention_launch(a, s);
    }
    void layernorm(const bf16_t* x, const Vec& w, const Vec& b, bf16_t* y, int M, int C) {
      if (rc || dry) return;
      rc = dfh::layernorm_launch(x, v32(w), v32(b), y, M, C, 1e-5f, s);
    }
    // f8: operand factors of the fp8 attention products (AttL::f8a_off) or null
    void attention(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const bf16_t* Vt, int ldvt, bf16_t* O, int C,
                   int heads, int Nq, int Nk, long vt_bstride = 0, const float* f8 = nullptr) {
      if (rc || dry) return;
      AttnArgs a; std::memset(&a, 0, sizeof(a));
      if (f8) { a.f8_rq = f8; a.f8_rk = f8 + C; a.f8_rv = f8 + 2 * C; a.f8_hs = f8 + 3 * C; }
      a.vt_bstride = vt_bstride;
      a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.Vt = Vt; a.ldvt = ldvt; a.O = O; a.ldo = C;
      a.B = B; a.H = heads; a.D = C / heads; a.Nq = Nq; a.Nk = Nk;
      a.scale = 1.0f / sqrtf((float)a.D);
      rc = dfh::attention_launch(a, s);
    }

    // 3x3 conv (pad 1) as implicit GEMM; optional stride-2 / fused nearest-2x upsample
    Tensor conv(const Tensor& x, const ConvL& c, int stride, int ups, bool to_persist) {
      const int Ho = ups ? x.H * 2 : (stride == 2 ? x.H / 2 : x.H);
      const int Wo = ups ? x.W * 2 : (stride == 2 ? x.W / 2 : x.W);
      Tensor o = to_persist ? palloc(Ho, Wo, c.cout) : talloc(Ho, Wo, c.cout);
      // nearest-2x upsample + conv: four 2x2 convs over the source image with the summed taps (4/9 of the multiply-adds), one launch
      // over the four phase planes.  DFH_UPS_PHASE=0 keeps the 3x3 conv over the virtual upsampled image (A/B).
      static const bool phase_off = [] { const char* e = getenv("DFH_UPS_PHASE"); return e && e[0] == '0'; }();
      if (ups == 1 && c.has_ph && u->fold_valid && !phase_off && !dry && x.C == c.cin) {
        GemmArgs g = base(B * x.H * x.W, c.cout);
        g.conv_src = x.p; g.conv_c = x.C; g.ntaps = 4; g.phase2x = 1; g.nbatch = 4; g.w_bs = (long)c.cout * 4 * x.C;
        g.Hin = x.H; g.Win = x.W; g.Hout = x.H; g.Wout = x.W; g.stride = 1; g.rows_per_b = x.H * x.W;
        g.W = u->fold_w() + c.ph; g.ldw = 4 * x.C; g.bias = v32(c.b);
        g.out = o.p;
        gemm(g);
        return o;
      }
      GemmArgs g = base(B * Ho * Wo, c.cout);
      g.conv_src = x.p; g.conv_c = x.C; g.ntaps = 9;
      g.Hin = x.H; g.Win = x.W; g.Hout = Ho; g.Wout = Wo; g.stride = stride; g.ups = ups;
      g.W = w16(c.w); g.ldw = c.w.K; g.bias = v32(c.b);
      g.out = o.p;
      gemm(g, &o, to_persist ? &persist : &temp);
      return o;
    }

    // stride-1 3x3 conv by Winograd F(2x2, 3x3) (winograd.hip): input transform, ONE batched GEMM over the sixteen transform-domain
    // planes, output transform with the epilogue (bias, time-embedding row, residual).  The scratch is planned by the dry run too.
    // Winograd F(2x2, 3x3) conv in three stages (winograd.hip); the scratch (V, M) is planned by the dry run too.
    //   wino_in   : V = B^T d B of the conv's input.  nw / nb: the GroupNorm(+SiLU) in front of the conv runs inside the transform (x, x1 = its
    //               raw, possibly concatenated input); Mprev: that input is the output transform of the PREVIOUS conv's planes (+ pbias + temb row),
    //               rebuilt inside the kernel (conv1 -> conv2 of a resnet); neither: x is the already normalised tensor
    //   wino_gemm : ONE batched GEMM over the sixteen transform-domain planes
    //   wino_out  : A^T m A + bias (+ time-embedding row) (+ residual)
    bf16_t* wino_in(const Tensor& x, const Tensor* x1, const Vec* nw, const Vec* nb, const bf16_t* Mprev = nullptr, const Vec* pbias = nullptr,
                    const float* prowvec = nullptr, int prv_off = 0) {
      const int C = x.C + (x1 ? x1->C : 0);
      const long mt = (long)B * (x.H / 2) * (x.W / 2);
      bf16_t* V = (bf16_t*)temp.alloc((size_t)16 * mt * C * 2);
      if (rc || dry) return V;
      if (nw) rc = dfh::gn_wino_input_launch(x.p, x.C, x1 ? x1->p : nullptr, x1 ? x1->C : 0, v32(*nw), v32(*nb), u->cfg.norm_eps,
                                             u->cfg.norm_num_groups, V, B, x.H, x.W, s, Mprev, pbias ? v32(*pbias) : nullptr, prowvec, temb_ld, prv_off);
      else rc = dfh::wino_input_launch(x.p, V, B, x.H, x.W, C, s);
      return V;
    }
    bf16_t* wino_gemm(const bf16_t* V, int H, int W, int C, size_t uoff, int cout) {
      const long mt = (long)B * (H / 2) * (W / 2);
      bf16_t* Mb = (bf16_t*)temp.alloc((size_t)16 * mt * cout * 2);
      if (rc || dry) return Mb;
      GemmArgs g = base((int)mt, cout);
      g.p_src[0] = V; g.p_c[0] = C; g.nplain = 1; g.W = u->fold_w() + uoff; g.ldw = C;
      g.nbatch = 16; g.a_bs = mt * C; g.w_bs = (long)cout * C; g.o_bs = mt * cout; g.w_blocked = dfh::wino_blocked(cout, C);
      g.out = Mb; g.zero = zero;
      g.prof_flops = 2.0 * B * H * W * (double)cout * 9.0 * C;
      rc = dfh::gemm_launch(g, s, dfh::wino_gemm_tile(g), 0, -1);
      dfh::census(dfh::CK_CONV_WINO);
      return Mb;
    }
    void wino_out(const bf16_t* Mb, const Vec& bias, const float* rowvec, int rv_off, const bf16_t* resid, Tensor& o) {
      if (rc || dry) return;
      rc = dfh::wino_output_launch(Mb, o.p, v32(bias), rowvec, temb_ld, rv_off, resid, B, o.H, o.W, o.C, s);
    }

    Tensor resnet(const Tensor& x0, const Tensor* x1, const ResL& r, const float* temb_all) {
      const int H = x0.H, W = x0.W;
      Tensor out = palloc(H, W, r.cout);
      const size_t mark = temp.off;
      Tensor g1 = talloc(H, W, r.cin);
      Tensor h1 = talloc(H, W, r.cout);
      // DFH_WINO: 0 = direct implicit GEMM everywhere, 1 = Winograd at H * W <= 64 (the 8x8 level), 2 = also at H * W <= 256 (A/B)
      static const int wino_mode = [] { const char* e = getenv("DFH_WINO"); return e ? atoi(e) : 2; }();
      const bool wino = r.has_u && (dry || u->fold_valid) && ((wino_mode >= 1 && H * W <= 64) || (wino_mode >= 2 && H * W <= wino_max_hw()));
      if (wino) {
        // the GroupNorms in front of the two convs run inside the input transforms where the (image, group) slab fits the kernel
        // (DFH_WINO_GN=0: separate GroupNorm launches, A/B)
        static const bool gn_off = [] { const char* e = getenv("DFH_WINO_GN"); return e && e[0] == '0'; }();
        const int G = u->cfg.norm_num_groups;
        // conv1 -> conv2: the tensor between them (conv1's output, GroupNorm 2's input) is rebuilt from conv1's transform-domain planes
        // inside conv2's input transform -- no output-transform launch, no round trip (DFH_WINO_CHAIN=0: materialise it, A/B)
        static const bool chain_off = [] { const char* e = getenv("DFH_WINO_CHAIN"); return e && e[0] == '0'; }();
        const bf16_t* V1;
        if (!gn_off && dfh::gn_wino_ok(x0.C, x1 ? x1->C : 0, G, H, W)) V1 = wino_in(x0, x1, &r.n1w, &r.n1b);
        else {
          groupnorm(x0, x1, r.n1w, r.n1b, u->cfg.norm_eps, 1, g1);
          V1 = wino_in(g1, nullptr, nullptr, nullptr);
        }
        const bf16_t* M1 = wino_gemm(V1, H, W, r.cin, r.u1, r.cout);
        const bool gn2 = !gn_off && dfh::gn_wino_ok(r.cout, 0, G, H, W);
        const bool chain = gn2 && !chain_off;
        Tensor g2 = talloc(H, W, r.cout);
        const bf16_t* V2;
        if (chain) V2 = wino_in(h1, nullptr, &r.n2w, &r.n2b, M1, &r.b1, temb_all, r.temb_off);     // h1 only names the shape: it is never written
        else {
          wino_out(M1, r.b1, temb_all, r.temb_off, nullptr, h1);
          if (gn2) V2 = wino_in(h1, nullptr, &r.n2w, &r.n2b);
          else {
            groupnorm(h1, nullptr, r.n2w, r.n2b, u->cfg.norm_eps, 1, g2);
            V2 = wino_in(g2, nullptr, nullptr, nullptr);
          }
        }
        const bf16_t* resid = x0.p;
        if (r.shortcut) {      // the 1x1 shortcut over the (possibly concatenated) block input: its own GEMM, added by the output transform
          Tensor sc = talloc(H, W, r.cout);
          GemmArgs g = base(B * H * W, r.cout);
          g.p_src[0] = x0.p; g.p_c[0] = x0.C; g.nplain = 1;
          if (x1) { g.p_src[1] = x1->p; g.p_c[1] = x1->C; g.nplain = 2; }
          g.W = w16(r.w2) + 9 * r.cout; g.ldw = r.w2.K;
          g.out = sc.p;
          gemm(g);
          resid = sc.p;
        }
        const bf16_t* M2 = wino_gemm(V2, H, W, r.cout, r.u2, r.cout);
        wino_out(M2, r.b2, nullptr, 0, resid, out);
        temp.off = mark;
        return out;
      }
      groupnorm(x0, x1, r.n1w, r.n1b, u->cfg.norm_eps, 1, g1);
      {
        GemmArgs g = base(B * H * W, r.cout);
        g.conv_src = g1.p; g.conv_c = r.cin; g.ntaps = 9;
        g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1;
        g.W = w16(r.w1); g.ldw = r.w1.K; g.bias = v32(r.b1);
        g.rowvec = temb_all; g.rv_ld = temb_ld; g.rv_off = r.temb_off; g.rows_per_b = H * W;
        g.out = h1.p;
        gemm(g, &h1, &temp);
      }
      Tensor g2 = talloc(H, W, r.cout);
      groupnorm(h1, nullptr, r.n2w, r.n2b, u->cfg.norm_eps, 1, g2);
      {
        GemmArgs g = base(B * H * W, r.cout);
        g.conv_src = g2.p; g.conv_c = r.cout; g.ntaps = 9;
        g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1;
        g.W = w16(r.w2); g.ldw = r.w2.K; g.bias = v32(r.b2);
        if (r.shortcut) {   // 1x1 shortcut over the (possibly concatenated) block input rides along as K segments
          g.p_src[0] = x0.p; g.p_c[0] = x0.C; g.nplain = 1;
          if (x1) { g.p_src[1] = x1->p; g.p_c[1] = x1->C; g.nplain = 2; }
        } else {
          g.resid = x0.p; g.ld_res = r.cout;
        }
        g.out = out.p;
        gemm(g, &out, &persist);
      }
      temp.off = mark;
      return out;
    }

    // pre_n > 0 (first transformer block of a guidance batch, dfh_unet::dup_tail): the last pre_n images have the same INPUT as the pre_n
    // before them and differ only in their text states, so everything up to the self-attention output is computed for B - pre_n images
    // and the repeated images' rows are copied; from the self-attention output projection on the block runs on the whole batch.
    Tensor transformer(const Tensor& x, const AttL& a, const bf16_t* kx, const bf16_t* vxt, int T, int pre_n = 0) {
      const int H = x.H, W = x.W, C = a.C, N = H * W;
      const int Bfull = B;
      int M = B * N;
      Tensor out = palloc(H, W, C);
      const size_t mark = temp.off;
      // LayerNorm folding (gemm.h, lnfold.hip): the GEMM that produces a LayerNorm's input leaves per-row statistics of its output, the
      // projections behind the LayerNorm run on the raw rows with gamma folded into their weights and fix the rows up in their
      // epilogue -- no layernorm_kernel launch, no normalised copy of the tensor.  Falls back to the LayerNorm kernel + plain weights
      // whenever the producer ran on a kernel that writes no statistics or a consumer would split K.  DFH_LN_FOLD=0 turns it off (A/B).
      static const bool fold_off = [] { const char* e = getenv("DFH_LN_FOLD"); return e && e[0] == '0'; }();
      const bool f8 = use8(a.qk8);                    // fp8 path: LayerNorm -> e4m3 + token scales -> block-scaled MFMA GEMM
      // round 4: proj_in, both to_out, ff.net.2 and proj_out in e4m3 as well, every operand quantised by the kernel that produces it
      const bool f8x = f8 && a.pin8.on;
      const bool fold = u->fold_valid && !fold_off && !f8 && !dry;
      const bool pre = pre_n > 0 && !dry && 2 * pre_n <= Bfull;
      if (pre) { B = Bfull - pre_n; M = B * N; }
      float* st = (float*)temp.alloc((size_t)Ba * N * ((C + 63) / 64) * 2 * sizeof(float));       // [C / bn][M][2], bn >= 64
      int bn = 0;
      auto try_folded = [&](std::initializer_list<GemmArgs> gs) {
        if (!fold || bn <= 0 || C % bn) return false;
        for (const GemmArgs& g : gs) if (!dfh::gemm_ln_consumer_ok(g)) return false;
        for (const GemmArgs& g : gs) gemm(g);
        dfh::census(dfh::CK_LN_FOLDED);
        return true;
      };
      // the K = N = C projections of a C = 320 block on the register-resident token-linear kernel (mlp_fused2.hip); DFH_TOKEN_LINEAR=0: dfh_gemm (A/B)
      // PROBE builds only (DFH_TOKEN_LINEAR=1 with the probe library): the K = N = C projections on the register-resident token-linear kernel
      // (scripts/probes/kernels/token_linear.hip).  Parity-tested, measured slower than the tile GEMM here -- 37 / 46 us against 27 / 34 us per
      // launch at M = 65536, sampling step 15.7 -> 15.9 ms (profiles/r05/token_linear_ab.txt): one workgroup per CU leaves its prologue (row
      // loads, first weight slice) and epilogue exposed twice per launch, and every 128-token tile re-streams the whole 200-KB matrix
#ifdef DFH_PROBES
      static const bool tl_on = [] { const char* e = getenv("DFH_TOKEN_LINEAR"); return e && e[0] == '1'; }();
      const bool tl = fold && tl_on && a.has_tl && dfh::token_linear_eligible(C, C, M);
      auto token_linear = [&](const bf16_t* xin, size_t img, const float* bias, const bf16_t* resid, const Fold* f, bf16_t* o, bool stats) {
        if (rc) return;
        TokLinArgs t; std::memset(&t, 0, sizeof(t));
        t.x = xin; t.img = (const unsigned char*)(u->fold_w() + img); t.bias = bias; t.resid = resid; t.out = o; t.M = M;
        if (f) { t.ln_stat = st; t.ln_parts = C / bn; t.ln_cnt = bn; t.ln_eps = 1e-5f; t.ln_s = u->fold_v() + f->s; t.bias = u->fold_v() + f->b; }
        if (stats) t.rowstat = st;
        rc = dfh::token_linear_launch(t, s);
        if (stats) bn = C;                            // one record per row over all C columns
      };
#else
      constexpr bool tl = false;
      auto token_linear = [](const bf16_t*, size_t, const float*, const bf16_t*, const Fold*, bf16_t*, bool) {};
#endif
      Tensor h0 = talloc(H, W, C);
      uint8_t* a8 = f8x ? (uint8_t*)temp.alloc((size_t)Ba * N * C) : nullptr;          // e4m3 operand of proj_in, then of the two to_out
      float* am_self = f8x ? amax_self + (size_t)a.idx * Ba : nullptr;
      const float* am_cross = f8x ? amax_cross + (size_t)a.idx * Ba : nullptr;
      if (f8x) {
        groupnorm8(x, 1e-6f, a8);
        Fp8GemmArgs g = args8(a8, M, a.pin8, nullptr, h0.p);
        g.bias = (const float*)(u->arena8 + a.pin8.boff); g.sa_mul = dfh_unet::GN_Z / 448.0f;
        gemm8(g);
      } else {
        // GroupNorm FOLDED into proj_in (norm.h GnFoldArgs): per-image weights W . gamma . rstd and a per-image row vector for the mean /
        // beta terms, so proj_in reads the block input itself and the normalised copy (one read + one write of the tensor) is never made.
        // Pays while the per-image weights (B x C x C) are small against the tensor: C <= DFH_GN_FOLD (default 320: the five 64x64-level
        // blocks; 0 = off).  Same box, sampling step: off 15.97 / 15.93, 320: 15.83 / 15.78, 640: 15.86 / 15.90, 1280: 16.01 / 16.00 ms
        // (profiles/r05/gn_fold_ab.txt).  The images must be whole 128-row tiles.
        static const int gn_fold_max = [] { const char* e = getenv("DFH_GN_FOLD"); return e ? atoi(e) : 320; }();
        const bool gfold = !tl && C <= gn_fold_max && N % 128 == 0 && a.pin.K == C && a.pin.N == C;
        if (gfold) {
          bf16_t* wimg = (bf16_t*)temp.alloc((size_t)Ba * C * C * 2);
          float* rv = (float*)temp.alloc((size_t)Ba * C * sizeof(float));
          if (!rc && !dry) {
            GnFoldArgs f; std::memset(&f, 0, sizeof(f));
            f.x = x.p; f.B = B; f.HW = N; f.C = C; f.G = u->cfg.norm_num_groups; f.eps = 1e-6f; f.gamma = v32(a.nw); f.beta = v32(a.nb);
            if (x.gst && x.gst_cpg == C / f.G) { f.pre = x.gst; f.pre_chunks = x.gst_chunks; }
            f.partial = gn_partial; f.W = w16(a.pin); f.ldw = a.pin.K; f.N = C; f.bias = v32(a.pinb); f.Wimg = wimg; f.rv = rv;
            rc = dfh::groupnorm_fold_launch(f, s);
          }
          GemmArgs g = base(M, C);
          g.p_src[0] = x.p; g.p_c[0] = C; g.nplain = 1;
          g.W = wimg; g.ldw = C; g.w_img_bs = (long)C * C;
          g.rowvec = rv; g.rv_ld = C; g.rv_off = 0; g.rows_per_b = N;
          g.out = h0.p; g.rowstat = fold ? st : nullptr;
          gemm(g, nullptr, nullptr, &bn);
        } else {
          Tensor gn = talloc(H, W, C);
          groupnorm(x, nullptr, a.nw, a.nb, 1e-6f, 0, gn);
          if (tl) token_linear(gn.p, a.tl_pin, v32(a.pinb), nullptr, nullptr, h0.p, true);
          else linear(gn.p, M, C, a.pin, &a.pinb, ACT_NONE, nullptr, h0.p, C, OUT_BF16, -1, 0, nullptr, nullptr, fold ? st : nullptr, &bn);
        }
      }
      // --- self attention
      Tensor n1 = talloc(H, W, C);
      uint8_t* n8 = f8 ? (uint8_t*)temp.alloc((size_t)Ba * N * C) : nullptr;
      float* s8 = f8 ? (float*)temp.alloc((size_t)Ba * N * sizeof(float)) : nullptr;
      Tensor qk = talloc(H, W, 2 * C);
      const int Np = (N + 7) & ~7;    // V^T rows padded to 8 keys (the 2x2 level of tiny configs has N = 4)
      bf16_t* vt = (bf16_t*)temp.alloc((size_t)Ba * C * Np * 2);   // [B][C][Np]
      // q | k and V^T from ONE launch (columns 2C .. 3C leave transposed into vt: GemmArgs::out2) wherever the column tile divides 2C;
      // DFH_QKV_MERGE=0 keeps the two launches (A/B)
      static const bool merge_off = [] { const char* e = getenv("DFH_QKV_MERGE"); return e && e[0] == '0'; }();
      auto with_v = [&](GemmArgs g) {               // q | k launch -> q | k | v: same rows, N = 3C, the v columns into vt
        g.N = 3 * C; g.out2 = vt; g.ld_out2 = Np; g.n_split = 2 * C; g.rows_per_b = N;
        return g;
      };
      const bool v_contig = a.v.off == a.qk.off + (size_t)2 * C * C && a.v.K == a.qk.K;   // packed back to back (build_attn)
      bool done = false;
      if (!f8 && !merge_off && !dry) {
        GemmArgs gq = with_v(folded(h0.p, M, a.fqk, st, bn, ACT_NONE, qk.p, OUT_BF16, 2 * C, 0));
        if (dfh::gemm_out2_ok(gq)) done = try_folded({gq});
      }
      if (!done) done = try_folded({folded(h0.p, M, a.fqk, st, bn, ACT_NONE, qk.p, OUT_BF16, -1, 0),
                                    folded(h0.p, M, a.fv, st, bn, ACT_NONE, vt, OUT_BF16_T, Np, N)});
      if (!done) {
        if (f8) layernorm8(h0.p, a.l1w, a.l1b, n8, s8, M, C);
        else layernorm(h0.p, a.l1w, a.l1b, n1.p, M, C);
        bool merged = false;
        if (!f8 && !merge_off && !dry && v_contig) {
          GemmArgs g = base(M, 2 * C);
          g.p_src[0] = n1.p; g.p_c[0] = C; g.nplain = 1; g.W = w16(a.qk); g.ldw = C; g.out = qk.p; g.ld_out = 2 * C;
          g = with_v(g);
          if (dfh::gemm_out2_ok(g)) { gemm(g); merged = true; }
        }
        if (!merged) {
          if (f8) linear8(n8, s8, M, a.qk8, nullptr, ACT_NONE, qk.p);
          else linear(n1.p, M, C, a.qk, nullptr, ACT_NONE, nullptr, qk.p, 2 * C);
          if (f8) linear8(n8, s8, M, a.v8, nullptr, ACT_NONE, vt, OUT_BF16_T, Np, N, am_self);
          else linear(n1.p, M, C, a.v, nullptr, ACT_NONE, nullptr, vt, C, OUT_BF16_T, Np, N);
        }
      }
      Tensor at = talloc(H, W, C);
      Tensor h1 = talloc(H, W, C);
      // self-attention products on the e4m3 MFMA where the layer has operand factors and the keys make whole 64-key tiles
      const float* f8attn = (f8 && a.f8a && N % 64 == 0) ? (const float*)(u->arena8 + a.f8a_off) : nullptr;
      if (f8x) {
        attention8(qk.p, 2 * C, qk.p + C, 2 * C, vt, Np, a8, am_self, C, a.heads, N, N, 0, f8attn);
        if (pre) {                                     // end of the shared prefix (e4m3 walk): rows, e4m3 attention output and its per-image max
          B = Bfull; M = B * N;
          dup_images(h0, pre_n); dup_bytes(a8, (size_t)N * C, pre_n); dup_bytes(am_self, sizeof(float), pre_n);
          dfh::census(dfh::CK_DUP_PREFIX);
        }
        Fp8GemmArgs g = args8(a8, M, a.o18, &a.o1b, h1.p);
        g.sA = am_self; g.sa_div = N; g.sa_mul = 1.0f / 448.0f; g.resid = h0.p; g.ld_res = C;
        gemm8(g);
      } else {
        attention(qk.p, 2 * C, qk.p + C, 2 * C, vt, Np, at.p, C, a.heads, N, N, 0, f8attn);
        if (pre) {                                     // end of the shared prefix: the whole batch from here on
          B = Bfull; M = B * N;
          dup_images(h0, pre_n); dup_images(at, pre_n);
          dfh::census(dfh::CK_DUP_PREFIX);
        }
        if (tl) token_linear(at.p, a.tl_o1, v32(a.o1b), h0.p, nullptr, h1.p, true);
        else linear(at.p, M, C, a.o1, &a.o1b, ACT_NONE, h0.p, h1.p, C, OUT_BF16, -1, 0, nullptr, nullptr, fold ? st : nullptr, &bn);
      }
      // --- cross attention over the T text tokens
      const int Tp = (T + 7) & ~7;
      if (tl && bn > 0 && C % bn == 0) {
        token_linear(h1.p, a.tl_q2, nullptr, nullptr, &a.fq2, qk.p, false);
        dfh::census(dfh::CK_LN_FOLDED);
      } else if (!try_folded({folded(h1.p, M, a.fq2, st, bn, ACT_NONE, qk.p, OUT_BF16, -1, 0)})) {
        if (f8) { layernorm8(h1.p, a.l2w, a.l2b, n8, s8, M, C); linear8(n8, s8, M, a.q28, nullptr, ACT_NONE, qk.p); }
        else {
          layernorm(h1.p, a.l2w, a.l2b, n1.p, M, C);
          linear(n1.p, M, C, a.q2, nullptr, ACT_NONE, nullptr, qk.p, C);
        }
      }
      // text K / V^T of this layer live inside the batched projections computed once per forward
      const int XT = u->x_total;
      Tensor h2 = talloc(H, W, C);
      if (f8x) {
        attention8(qk.p, C, kx + a.x_off, XT, vxt + (size_t)a.x_off * Tp, Tp, a8, am_cross, C, a.heads, N, T, (long)XT * Tp);
        Fp8GemmArgs g = args8(a8, M, a.o28, &a.o2b, h2.p);
        g.sA = am_cross; g.sa_div = N; g.sa_mul = 1.0f / 448.0f; g.resid = h1.p; g.ld_res = C;
        gemm8(g);
      } else {
        attention(qk.p, C, kx + a.x_off, XT, vxt + (size_t)a.x_off * Tp, Tp, at.p, C, a.heads, N, T, (long)XT * Tp);
        if (tl) token_linear(at.p, a.tl_o2, v32(a.o2b), h1.p, nullptr, h2.p, true);
        else linear(at.p, M, C, a.o2, &a.o2b, ACT_NONE, h1.p, h2.p, C, OUT_BF16, -1, 0, nullptr, nullptr, fold ? st : nullptr, &bn);
      }
      // --- GEGLU feed-forward
      if (f8x) {
        // hidden tensor in e4m3 with one E8M0 scale per token and 32 hidden units, written by the GEGLU epilogue and consumed by ff.net.2
        // through the MFMA's scale operand; ff.net.2's output (+ bias + h2) likewise, consumed by proj_out (+ the block's residual)
        uint8_t* ff8 = (uint8_t*)temp.alloc((size_t)M * 4 * C);
        uint8_t* ffsx = (uint8_t*)temp.alloc((size_t)(4 * C / 32) * M + 256);
        uint8_t* t8 = (uint8_t*)temp.alloc((size_t)M * C);
        uint8_t* tsx = (uint8_t*)temp.alloc((size_t)(C / 32) * M + 256);
        layernorm8(h2.p, a.l3w, a.l3b, n8, s8, M, C);
        Fp8GemmArgs g1 = args8(n8, M, a.ff18, &a.ff1b, ff8);
        g1.sA = s8; g1.act = ACT_GEGLU; g1.out_mode = OUT_FP8_MX; g1.ld_out = 4 * C; g1.out_sx = ffsx;
        gemm8(g1);
        Fp8GemmArgs g2 = args8(ff8, M, a.ff28, &a.ff2b, t8);
        g2.sx = ffsx; g2.resid = h2.p; g2.ld_res = C; g2.out_mode = OUT_FP8_MX; g2.ld_out = C; g2.out_sx = tsx;
        gemm8(g2);
        Fp8GemmArgs g3 = args8(t8, M, a.pout8, &a.poutb, out.p);
        g3.sx = tsx; g3.resid = x.p; g3.ld_res = C;
        gemm8(g3);
        temp.off = mark;
        return out;
      }
      // statistics of the block's output for the next GroupNorm, written by the fused kernel's epilogue in 128-token chunks (allocated in the
      // dry run as well: the unfused path plans 256-token chunks through gemm())
      const int G = u->cfg.norm_num_groups;
      const bool mlp_gst_ok = a.has_mlp && N % 128 == 0 && C % G == 0 && N / 128 <= (int)GN_MAX_CHUNKS;
      float* mlp_gst = mlp_gst_ok ? (float*)persist.alloc((size_t)B * G * (N / 128) * 2 * sizeof(float)) : nullptr;
      Tensor ff = talloc(H, W, 4 * C);
      // the whole feed-forward + proj_out in one kernel where the X tile fits the register file (C = 320: the 64x64 level); needs the row
      // statistics of h2 from its producer like every folded-LayerNorm consumer.  DFH_MLP_FUSED=0: the two-launch walk, 1 / 2: the two forms of the kernel (A/B)
      const bool mlp_off = dfh::mlp_fused_form() == 0;
      if (fold && !mlp_off && a.has_mlp && bn > 0 && C % bn == 0 && dfh::mlp_fused_eligible(C, M)) {
        MlpArgs ma; std::memset(&ma, 0, sizeof(ma));
        ma.x = h2.p; ma.resid = x.p; ma.img = (const unsigned char*)(u->fold_w() + a.mlp_img);
        ma.ln_stat = st; ma.ln_parts = C / bn; ma.ln_cnt = bn; ma.ln_eps = 1e-5f;
        ma.bias = u->fold_v() + a.fffp.b; ma.out = out.p; ma.M = M;
        static const bool pre_off = [] { const char* e = getenv("DFH_GN_PRE"); return e && e[0] == '0'; }();
        if (dfh::mlp_fused_form() == 2 && mlp_gst && !pre_off) {
          ma.gstat = mlp_gst; ma.gstat_cpg = C / G; ma.gstat_hw = N;
          out.gst = mlp_gst; out.gst_cpg = C / G; out.gst_chunks = N / 128;
          dfh::census(dfh::CK_GSTAT_WRITTEN);
        }
#ifdef DFH_PROBES
        if (!rc && dfh::mlp_fused_form() == 1) rc = dfh::mlp_fused_launch(ma, s); else
#endif
        if (!rc) rc = dfh::mlp2_fused_launch(ma, s);
        dfh::census(dfh::CK_LN_FOLDED);                  // LayerNorm 3 is consumed folded here too
        temp.off = mark;
        return out;
      }
      if (!try_folded({folded(h2.p, M, a.fff1, st, bn, ACT_GEGLU, ff.p, OUT_BF16, -1, 0)})) {
        if (f8) { layernorm8(h2.p, a.l3w, a.l3b, n8, s8, M, C); linear8(n8, s8, M, a.ff18, &a.ff1b, ACT_GEGLU, ff.p); }
        else {
          layernorm(h2.p, a.l3w, a.l3b, n1.p, M, C);
          linear(n1.p, M, C, a.ff1, &a.ff1b, ACT_GEGLU, nullptr, ff.p, 8 * C);
        }
      }
      static const bool ffp_off = [] { const char* e = getenv("DFH_FFP_FOLD"); return e && e[0] == '0'; }();      // A/B switch
      // ff.net.2 and proj_out as ONE linear over the two K segments [GEGLU output | h2] (AttL::fffp) + the block's residual
      GemmArgs gf = base(M, C);
      gf.p_src[0] = ff.p; gf.p_c[0] = 4 * C; gf.p_src[1] = h2.p; gf.p_c[1] = C; gf.nplain = 2;
      gf.ldw = 5 * C; gf.resid = x.p; gf.ld_res = C; gf.out = out.p;
      if (dry) gemm(gf);                                                   // planning: its split-K slabs, whichever path runs later
      if (u->fold_valid && !ffp_off && !dry) {
        gf.W = u->fold_w() + a.fffp.w; gf.bias = u->fold_v() + a.fffp.b;
        gemm(gf, &out, &persist);                                          // feeds the next block's GroupNorm
      } else {
        linear(ff.p, M, 4 * C, a.ff2, &a.ff2b, ACT_NONE, h2.p, h0.p, C);   // h0 is dead by now: reuse
        linear(h0.p, M, C, a.pout, &a.poutb, ACT_NONE, x.p, out.p, C, OUT_BF16, -1, 0, &out, &persist);   // feeds the next block's GroupNorm
      }
      temp.off = mark;
      return out;
    }
  };

  // Per-run constants of a sampling loop (reference DiFashion/models/difashion.py:340-357: the prompt states are fixed for the run;
  // :456: the timesteps are the schedule's): the cross-attention K / V^T of every transformer block and the time-embedding rows
  // (all time_emb_proj outputs) per schedule entry, computed once by run_cache() into a caller-owned buffer.
  struct RunCache { const bf16_t* kx = nullptr; const bf16_t* vxt = nullptr; const float* temb_row = nullptr; const float* xamax = nullptr; };
  // fp8 walk: the largest |V| of every layer's cross-attention per batch element, [n_att][B] floats behind the time-embedding table
  size_t cache_xamax_bytes(int B) const { return fp8 ? (((size_t)n_att * B * 4 + 255) & ~(size_t)255) : 0; }
  static size_t cache_kx_bytes(const dfh_unet& u, int B) { return ((size_t)B * u.cfg.text_len * u.x_total * 2 + 255) & ~(size_t)255; }
  static size_t cache_vxt_bytes(const dfh_unet& u, int B) { return ((size_t)B * u.x_total * ((u.cfg.text_len + 7) & ~7) * 2 + 255) & ~(size_t)255; }
  size_t run_cache_bytes(int B, int n_t) const {
    return cache_kx_bytes(*this, B) + cache_vxt_bytes(*this, B) + (((size_t)n_t * temb_total * 4 + 255) & ~(size_t)255) + cache_xamax_bytes(B);
  }
  // amax_cross[layer][b] = max |V^T| of the layer's slice of the batched text V^T
  int cross_amax(const bf16_t* vxt, int B, float* out, hipStream_t s) {
    const int Tp = (cfg.text_len + 7) & ~7;
    // the pad columns T .. Tp - 1 of V^T are never written: only the T real keys count
    return dfh::amax_slabs_launch(vxt, (long)x_total * Tp, Tp, cfg.text_len, slab_row0(), slab_rows(), out, n_att, B, s);
  }

  int run(const void* sample, int sample_bf16, const float* timestep, const void* ehs, int ehs_bf16, float* out, int B,
          hipStream_t s, bool dry, const RunCache* rcache = nullptr) {
    Run r; r.u = this; r.B = B; r.Ba = B; r.s = s; r.dry = dry;
    r.temb_ld = (rcache && rcache->temb_row) ? 0 : temb_total;
    // one-shot hint of the caller (dfh_unet_set_dup_tail): the last `dup` images repeat the sample / timestep of the `dup` before them
    const int dup = (!dry && dup_tail > 0 && 2 * dup_tail <= B) ? dup_tail : 0;
    if (!dry) dup_tail = 0;
    const int S = cfg.sample_size, T = cfg.text_len, X = cfg.cross_attention_dim;
    const int* boc = cfg.block_out_channels;
    const int nb = cfg.num_blocks, temb = boc[0] * 4;
    // fixed regions at the head of the workspace
    char* const wsb = dry ? nullptr : ws + fold_bytes();      // the fold region comes first (fold_layernorms)
    Bump head; head.base = wsb;
    r.zero = (bf16_t*)head.alloc(256);
    r.gn_partial = (float*)head.alloc((size_t)B * GN_MAX_CHUNKS * 64 * 2 * sizeof(float));
    r.partial = (float*)head.alloc(dry ? 0 : plan_partial);
    r.partial_cap = dry ? 0 : plan_partial;
    const size_t head_bytes = (head.off + 255) & ~(size_t)255;
    if (!dry) {
      if (B != plan_batch) { dfh::set_error("forward batch differs from the planned batch"); return -1; }
      r.persist.base = wsb + head_bytes;
      r.temp.base = wsb + head_bytes + plan_persist;
      if (fold_bytes() + head_bytes + plan_persist + plan_temp > ws_bytes) { dfh::set_error("workspace too small"); return -1; }
      (void)hipMemsetAsync(r.zero, 0, 256, s);
    }
    if (!dry && fold_dirty) { if (int rc = fold_layernorms(s)) return rc; }      // derived weights (LayerNorm / ff2 . proj_out folds): lazily
    taps.clear();

    // ---- time embedding: sinusoid -> MLP (SiLU folded into both epilogues: only silu(emb) is ever
    //      consumed) -> all 22 time_emb_proj rows in one GEMM (fp32 [B][temb_total])
    bf16_t* tsin = (bf16_t*)r.persist.alloc((size_t)B * boc[0] * 2);
    bf16_t* e1 = (bf16_t*)r.persist.alloc((size_t)B * temb * 2);
    bf16_t* e2 = (bf16_t*)r.persist.alloc((size_t)B * temb * 2);
    float* temb_all = (float*)r.persist.alloc((size_t)B * temb_total * 4);
    const bool t_cached = rcache && rcache->temb_row, x_cached = rcache && rcache->kx && rcache->vxt;
    if (!dry && !t_cached) r.rc = dfh::timestep_embed_launch(timestep, tsin, B, boc[0], s);
    if (dry || !t_cached) {          // the dry run plans for the uncached walk (same workspace either way)
      r.linear(tsin, B, boc[0], te1, &te1b, ACT_SILU, nullptr, e1, temb);
      r.linear(e1, B, temb, te2, &te2b, ACT_SILU, nullptr, e2, temb);
      r.linear(e2, B, temb, tproj, &tprojb, ACT_NONE, nullptr, temb_all, temb_total, OUT_F32);
    } else {
      temb_all = const_cast<float*>(rcache->temb_row);
    }

    // ---- inputs to kernel layout
    bf16_t* ehs16 = (bf16_t*)r.persist.alloc((size_t)B * T * X * 2);
    if (!dry && !r.rc && !x_cached) {
      if (ehs_bf16) (void)hipMemcpyAsync(ehs16, ehs, (size_t)B * T * X * 2, hipMemcpyDeviceToDevice, s);
      else r.rc = dfh::cast_f32_to_bf16_launch((const float*)ehs, ehs16, (long)B * T * X, s);
    }
    // text K for every transformer layer in one GEMM ([B*T][x_total]) and V^T in another ([B][x_total][Tp])
    const int Tp = (T + 7) & ~7;
    bf16_t* kx = (bf16_t*)r.persist.alloc((size_t)B * T * x_total * 2);
    bf16_t* vxt = (bf16_t*)r.persist.alloc((size_t)B * x_total * Tp * 2);
    if (dry || !x_cached) {
      r.linear(ehs16, B * T, X, kx_all, nullptr, ACT_NONE, nullptr, kx, x_total);
      r.linear(ehs16, B * T, X, vx_all, nullptr, ACT_NONE, nullptr, vxt, x_total, OUT_BF16_T, Tp, T);
    } else {
      kx = const_cast<bf16_t*>(rcache->kx); vxt = const_cast<bf16_t*>(rcache->vxt);
      dfh::census(dfh::CK_TEXT_CACHED);
    }
    if (fp8) {
      r.amax_self = (float*)r.persist.alloc((size_t)n_att * B * 4);
      float* xam = (float*)r.persist.alloc((size_t)n_att * B * 4);
      r.amax_cross = xam;
      if (!dry && !r.rc) {
        if (hipMemsetAsync(r.amax_self, 0, (size_t)n_att * B * 4, s) != hipSuccess) { dfh::set_error("hipMemsetAsync failed"); return -2; }
        if (x_cached && rcache->xamax) r.amax_cross = rcache->xamax;
        else r.rc = cross_amax(vxt, B, xam, s);
      }
    }
    Tensor x = r.palloc(S, S, conv_in.cin);   // in_channels padded to a multiple of 8
    if (!dry && !r.rc) r.rc = dfh::nchw_to_nhwc_launch(sample, sample_bf16, x.p, B, cfg.in_channels, S * S, s);
    if (dup && !r.rc) {
      // DFH_CHECK_DUP=1 (debugging a caller): verify what the hint claims -- the repeated images' inputs equal the ones they repeat --
      // with a synchronous compare of the converted input rows; a wrong hint is an error, not a silently different result
      static const bool check = [] { const char* e = getenv("DFH_CHECK_DUP"); return e && e[0] == '1'; }();
      if (check) {
        const size_t per = (size_t)S * S * conv_in.cin * 2, n = (size_t)dup * per;
        std::vector<char> a(n), b(n);
        if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(a.data(), (char*)x.p + (size_t)(B - 2 * dup) * per, n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(b.data(), (char*)x.p + (size_t)(B - dup) * per, n, hipMemcpyDeviceToHost) != hipSuccess) { dfh::set_error("DFH_CHECK_DUP: copy failed"); return -2; }
        if (std::memcmp(a.data(), b.data(), n) != 0) { dfh::set_error("dfh_unet_set_dup_tail: the last images do NOT repeat the inputs of the ones before them"); return -1; }
        if (timestep && !(rcache && rcache->temb_row)) {
          std::vector<float> t(B);
          if (hipMemcpy(t.data(), timestep, (size_t)B * 4, hipMemcpyDeviceToHost) != hipSuccess) { dfh::set_error("DFH_CHECK_DUP: copy failed"); return -2; }
          for (int i = 0; i < dup; ++i)
            if (t[B - dup + i] != t[B - 2 * dup + i]) { dfh::set_error("dfh_unet_set_dup_tail: the repeated images sit at other timesteps"); return -1; }
        }
      }
    }

    // Shared prefix of a guidance batch (reference difashion.py:388-427, 494-512: the branches of classifier-free guidance that differ only
    // in their PROMPT get the same latent / mutual / history input): conv_in, the first resnet and the first transformer block up to its
    // self-attention see no text state, so they run once for the repeated images.
    if (dup) r.B = B - dup;
    Tensor h = r.conv(x, conv_in, 1, 0, true);
    taps["conv_in"] = h;
    std::vector<Tensor> skips{h};
    for (int i = 0; i < nb; ++i) {
      for (int j = 0; j < cfg.layers_per_block; ++j) {
        const bool first = dup && i == 0 && j == 0;
        h = r.resnet(h, nullptr, down_res[i][j], temb_all);
        if (first) {
          r.B = B;
          r.dup_images(skips[0], dup); taps["conv_in"] = skips[0];
          if (!cfg.down_attn[i]) r.dup_images(h, dup);
          else { const float* g = h.gst; r.dup_images(h, dup); h.gst = g; }     // the block's entry GroupNorm still runs on the prefix (transformer(): pre_n), which the producer statistics cover
        }
        if (cfg.down_attn[i]) h = r.transformer(h, down_att[i][j], kx, vxt, T, first ? dup : 0);
        skips.push_back(h);
      }
      if (i != nb - 1) { h = r.conv(h, down_samp[i], 2, 0, true); skips.push_back(h); }
      taps["down" + std::to_string(i)] = h;
    }
    h = r.resnet(h, nullptr, mid_res[0], temb_all);
    h = r.transformer(h, mid_att, kx, vxt, T);
    h = r.resnet(h, nullptr, mid_res[1], temb_all);
    taps["mid"] = h;
    for (int i = 0; i < nb; ++i) {
      for (int j = 0; j < (int)up_res[i].size(); ++j) {
        Tensor sk = skips.back(); skips.pop_back();
        h = r.resnet(h, &sk, up_res[i][j], temb_all);
        if (!up_att[i].empty()) h = r.transformer(h, up_att[i][j], kx, vxt, T);
      }
      if (i != nb - 1) h = r.conv(h, up_samp[i], 1, 1, true);
      taps["up" + std::to_string(i)] = h;
    }
    Tensor g = r.palloc(h.H, h.W, h.C);
    r.groupnorm(h, nullptr, cnw, cnb, cfg.norm_eps, 1, g);
    {
      GemmArgs ga = Run::base(B * S * S, cfg.out_channels);
      ga.conv_src = g.p; ga.conv_c = g.C; ga.ntaps = 9;
      ga.Hin = S; ga.Win = S; ga.Hout = S; ga.Wout = S; ga.stride = 1;
      ga.W = r.w16(conv_out.w); ga.ldw = conv_out.w.K; ga.bias = r.v32(conv_out.b);
      ga.out = out; ga.out_mode = OUT_F32_T; ga.ld_out = S * S; ga.rows_per_b = S * S;
      r.gemm(ga);
    }
    if (dry) {
      plan_persist = (r.persist.peak + 255) & ~(size_t)255;
      plan_temp = (r.temp.peak + 255) & ~(size_t)255;
      plan_partial = (r.partial_need + 255) & ~(size_t)255;
      plan_batch = B;
      // head is re-derived with the real partial size
      Bump hd; hd.alloc(256); hd.alloc((size_t)B * GN_MAX_CHUNKS * 64 * 2 * sizeof(float)); hd.alloc(plan_partial);
      plan_total = fold_bytes() + ((hd.off + 255) & ~(size_t)255) + plan_persist + plan_temp;
      taps.clear();
    } else {
      last_batch = B;
    }
    return r.rc;
  }

  // Fills a run cache (layout: kx | vxt | temb table [n_t][temb_total] fp32).  Uses the bound workspace as scratch, in chunks of the
  // planned batch so that every launch has a shape the workspace plan covered (split-K slabs included).
  int run_cache(const void* ehs, int ehs_bf16, int B, const float* timesteps, int n_t, void* cache, hipStream_t s) {
    if (B != plan_batch) run(nullptr, 0, nullptr, nullptr, 0, nullptr, B, nullptr, true);
    DFH_REQUIRE(plan_total <= ws_bytes, "workspace too small for this batch");
    Run r; r.u = this; r.B = B; r.Ba = B; r.s = s; r.dry = false;
    const int T = cfg.text_len, X = cfg.cross_attention_dim, Tp = (T + 7) & ~7;
    const int temb = cfg.block_out_channels[0] * 4, c0 = cfg.block_out_channels[0];
    Bump head; head.base = ws + fold_bytes();
    r.zero = (bf16_t*)head.alloc(256);
    r.gn_partial = (float*)head.alloc((size_t)B * GN_MAX_CHUNKS * 64 * 2 * sizeof(float));
    r.partial = (float*)head.alloc(plan_partial); r.partial_cap = plan_partial;
    Bump tmp; tmp.base = head.base + ((head.off + 255) & ~(size_t)255);
    (void)hipMemsetAsync(r.zero, 0, 256, s);
    bf16_t* kx = (bf16_t*)cache;
    bf16_t* vxt = (bf16_t*)((char*)cache + cache_kx_bytes(*this, B));
    float* table = (float*)((char*)cache + cache_kx_bytes(*this, B) + cache_vxt_bytes(*this, B));
    float* xamax = (float*)((char*)table + (((size_t)n_t * temb_total * 4 + 255) & ~(size_t)255));
    bf16_t* ehs16 = (bf16_t*)tmp.alloc((size_t)B * T * X * 2);
    if (ehs_bf16) (void)hipMemcpyAsync(ehs16, ehs, (size_t)B * T * X * 2, hipMemcpyDeviceToDevice, s);
    else r.rc = dfh::cast_f32_to_bf16_launch((const float*)ehs, ehs16, (long)B * T * X, s);
    r.linear(ehs16, B * T, X, kx_all, nullptr, ACT_NONE, nullptr, kx, x_total);
    r.linear(ehs16, B * T, X, vx_all, nullptr, ACT_NONE, nullptr, vxt, x_total, OUT_BF16_T, Tp, T);
    if (fp8 && !r.rc) r.rc = cross_amax(vxt, B, xamax, s);
    bf16_t* tsin = (bf16_t*)tmp.alloc((size_t)B * c0 * 2);
    bf16_t* e1 = (bf16_t*)tmp.alloc((size_t)B * temb * 2);
    bf16_t* e2 = (bf16_t*)tmp.alloc((size_t)B * temb * 2);
    float* trow = (float*)tmp.alloc((size_t)B * temb_total * 4);
    DFH_REQUIRE(fold_bytes() + ((head.off + 255) & ~(size_t)255) + tmp.off <= ws_bytes, "workspace too small for the run cache scratch");
    for (int t0 = 0; t0 < n_t && !r.rc; t0 += B) {
      // always B rows (the planned GEMM shapes); rows past n_t repeat the last timestep and are not copied out
      const int n = std::min(B, n_t - t0);
      r.rc = dfh::timestep_embed_launch(timesteps + t0, tsin, n, c0, s);
      if (r.rc) break;
      if (n < B) (void)hipMemsetAsync(tsin + (size_t)n * c0, 0, (size_t)(B - n) * c0 * 2, s);
      r.linear(tsin, B, c0, te1, &te1b, ACT_SILU, nullptr, e1, temb);
      r.linear(e1, B, temb, te2, &te2b, ACT_SILU, nullptr, e2, temb);
      r.linear(e2, B, temb, tproj, &tprojb, ACT_NONE, nullptr, trow, temb_total, OUT_F32);
      if (!r.rc) (void)hipMemcpyAsync(table + (size_t)t0 * temb_total, trow, (size_t)n * temb_total * 4, hipMemcpyDeviceToDevice, s);
    }
    return r.rc;
  }

  OpTable tab_pack, tab_pack_acc, tab_packt, tab_unpack, tab_pack2;
  int pack_all(const float* const* master, int count, hipStream_t s);      // training: pack() + pack_train() with one read of the weights (unet_train.hip)
  // every PackOp in one launch (plus one for the few biases that ADD onto an already packed vector)
  int pack(const float* const* master, int count, hipStream_t s) {
    DFH_REQUIRE(count == (int)params.size(), "parameter count mismatch");
    DFH_REQUIRE(arena16 && arena32, "arenas not bound");
    tab_pack.clear(); tab_pack_acc.clear();
    for (const PackOp& op : packs) {
      void* src = (void*)master[op.param];
      DFH_REQUIRE(src != nullptr, "null master parameter: " + params[op.param].name);
      if (op.kind == PK_VEC) (op.accumulate ? tab_pack_acc : tab_pack).add(src, TAB_PACK_VEC, (long)op.dst, op.N, 0, 0, op.geglu, op.accumulate, 0, 0, op.N);
      else if (op.kind == PK_MAT) tab_pack.add(src, TAB_PACK_MAT, (long)op.dst, op.N, op.K, op.ldw, op.row_off, op.col_off, op.geglu, 0, (long)op.N * op.K);
      else tab_pack.add(src, TAB_PACK_CONV, (long)op.dst, op.N, op.K, op.ldw, 0, op.col_off, 0, op.cin_pad, (long)op.N * op.K * 9);
    }
    if (int rc = tab_pack.launch(arena32, arena16, s)) return rc;
    if (int rc = tab_pack_acc.launch(arena32, arena16, s)) return rc;
    if (int rc = quantize_fp8(s)) return rc;   // e4m3 copies of the LayerNorm-fed projections from the freshly packed bf16 matrices
    fold_valid = false; fold_dirty = true;     // the folded copies are re-derived by the next INFERENCE walk (a training step never pays)
    return 0;
  }
};
