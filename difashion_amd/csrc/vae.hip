// AutoencoderKL (Stable-Diffusion VAE) on the same gfx950 kernels as the U-Net: SURVEY.md 8(f)-1, the step on either side of
// the denoising path.
//
// Replaces (arithmetic): diffusers 0.18.2 AutoencoderKL as the reference calls it -- vae.encode(images).latent_dist
// (DiFashion/models/difashion.py:129, :144, :376, :435-437) and vae.decode(latents / scaling_factor) (:580).  Topology and
// parameter names restated in oracle/vae_ref.py.
//
// Everything is the implicit GEMM of gemm.hip / gemm_wide.hip on NHWC bf16 activations plus GroupNorm(+SiLU):
//   * resnets without time embedding; 1x1 shortcut fused into conv2 as a K segment;
//   * Downsample2D = conv3x3 stride 2 over the input zero-extended right / bottom (GemmArgs.pad0), Upsample2D = the
//     fused nearest-2x conv;
//   * the mid-block attention has ONE head of width C (512): too wide for the register-resident flash kernel, and tiny
//     next to the convs (4 x 34 GFLOP per image against 1.2 TFLOP), so it runs as three GEMMs per image -- S = Q K^T
//     (fp32 out), row softmax (softmax_rows_kernel), O = P V against V^T from the transposed epilogue;
//   * 3-channel images / 4-channel latents are padded to one 16-byte NHWC pixel (8 channels), conv_out to 4 outputs.
#include "unet_model.h"

namespace {

struct VRes { int cin = 0, cout = 0; bool shortcut = false; Vec n1w, n1b, b1, n2w, n2b, b2; Mat w1, w2; };
struct VAtt { int C = 0; Vec gw, gb, qb, kb, vb, ob; Mat q, k, v, o; };
struct VConv { Mat w; Vec b; int cin = 0, cout = 0, npad = 0; };

// P[r][c] = softmax_c(scale * S[r][c]) in bf16; one block per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, bf16_t* __restrict__ P, int cols, int ld, float scale) {
  const long row = blockIdx.x;
  const float* s = S + row * ld;
  __shared__ float red[4];
  float mx = -3.0e38f;
  for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, s[c]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) sum += __expf((s[c] - mx) * scale);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  bf16_t* p = P + row * ld;
  for (int c = threadIdx.x; c < cols; c += 256) p[c] = f2bf(__expf((s[c] - mx) * scale) * inv);
}

}  // namespace

struct dfh_vae {
  dfh_vae_config cfg{};
  std::vector<ParamDesc> params;
  std::vector<PackOp> packs;
  size_t a16 = 0, a32 = 0;
  // layers
  VConv e_in, e_out, d_in, d_out, quant, post_quant;
  std::vector<std::vector<VRes>> e_res, d_res;
  std::vector<VConv> e_down, d_up;
  VRes e_mid[2], d_mid[2]; VAtt e_att, d_att;
  Vec e_nw, e_nb, d_nw, d_nb;
  // bound memory
  bf16_t* arena16 = nullptr; float* arena32 = nullptr;
  char* ws = nullptr; size_t ws_bytes = 0;
  OpTable tab_pack, tab_pack_acc;

  // ---------------------------------------------------------------- build (same packing conventions as unet_model.h)
  int add_param(const std::string& name, std::vector<int> shape) { params.push_back({name, std::move(shape)}); return (int)params.size() - 1; }
  size_t alloc16(size_t n) { size_t o = a16; a16 += (n + 127) & ~(size_t)127; return o; }
  size_t alloc32(size_t n) { size_t o = a32; a32 += (n + 63) & ~(size_t)63; return o; }
  Vec vec(const std::string& name, int N, int npad = 0) {
    Vec v; v.N = npad ? npad : N; v.off = alloc32(v.N);
    packs.push_back({add_param(name, {N}), PK_VEC, v.off, N, 0, 0, 0, 0, 0, 0});
    return v;
  }
  Mat mat_alloc(int N, int K) { Mat m; m.N = N; m.K = K; m.off = alloc16((size_t)N * K); return m; }
  void conv_into(const std::string& name, int cout, int cin, const Mat& dst, int cin_pad) {
    PackOp op{add_param(name, {cout, cin, 3, 3}), PK_CONV3, dst.off, cout, cin, dst.K, 0, 0, 0, 0};
    op.cin_pad = cin_pad;
    packs.push_back(op);
  }
  void mat_into(const std::string& name, int N, int K, bool conv1x1, const Mat& dst, int col_off) {
    std::vector<int> shape = conv1x1 ? std::vector<int>{N, K, 1, 1} : std::vector<int>{N, K};
    packs.push_back({add_param(name, shape), PK_MAT, dst.off, N, K, dst.K, 0, col_off, 0, 0});
  }
  void build_conv(const std::string& pre, int cout, int cin, VConv& c) {
    const int cp = (cin + 7) & ~7;
    c.cin = cp; c.cout = cout; c.npad = (cout + 3) & ~3;            // GEMM N is a multiple of 4: padded rows stay zero
    c.w = mat_alloc(c.npad, 9 * cp);
    conv_into(pre + ".weight", cout, cin, c.w, cp);
    c.b = vec(pre + ".bias", cout, c.npad);
  }
  void build_conv1x1(const std::string& pre, int cout, int cin, VConv& c) {
    const int cp = (cin + 7) & ~7;
    c.cin = cp; c.cout = cout; c.npad = (cout + 3) & ~3;
    c.w = mat_alloc(c.npad, cp);
    mat_into(pre + ".weight", cout, cin, true, c.w, 0);
    c.b = vec(pre + ".bias", cout, c.npad);
  }
  void build_resnet(const std::string& pre, int cin, int cout, VRes& r) {
    r.cin = cin; r.cout = cout; r.shortcut = cin != cout;
    r.n1w = vec(pre + ".norm1.weight", cin); r.n1b = vec(pre + ".norm1.bias", cin);
    r.w1 = mat_alloc(cout, 9 * cin);
    conv_into(pre + ".conv1.weight", cout, cin, r.w1, cin);
    r.b1 = vec(pre + ".conv1.bias", cout);
    r.n2w = vec(pre + ".norm2.weight", cout); r.n2b = vec(pre + ".norm2.bias", cout);
    r.w2 = mat_alloc(cout, 9 * cout + (r.shortcut ? cin : 0));
    conv_into(pre + ".conv2.weight", cout, cout, r.w2, cout);
    r.b2 = vec(pre + ".conv2.bias", cout);
    if (r.shortcut) {
      mat_into(pre + ".conv_shortcut.weight", cout, cin, true, r.w2, 9 * cout);
      packs.push_back({add_param(pre + ".conv_shortcut.bias", {cout}), PK_VEC, r.b2.off, cout, 0, 0, 0, 0, 0, /*accumulate=*/1});
    }
  }
  void build_mid(const std::string& pre, int C, VRes* res, VAtt& a) {
    build_resnet(pre + ".resnets.0", C, C, res[0]);
    const std::string ap = pre + ".attentions.0";
    a.C = C;
    a.gw = vec(ap + ".group_norm.weight", C); a.gb = vec(ap + ".group_norm.bias", C);
    auto lin = [&](const std::string& n, Mat& m, Vec& b) {
      m = mat_alloc(C, C);
      mat_into(ap + "." + n + ".weight", C, C, false, m, 0);
      b = vec(ap + "." + n + ".bias", C);
    };
    lin("to_q", a.q, a.qb); lin("to_k", a.k, a.kb); lin("to_v", a.v, a.vb); lin("to_out.0", a.o, a.ob);
    build_resnet(pre + ".resnets.1", C, C, res[1]);
  }
  int build() {
    const int nb = cfg.num_blocks, L = cfg.layers_per_block;
    const int* boc = cfg.block_out_channels;
    build_conv("encoder.conv_in", boc[0], cfg.in_channels, e_in);
    e_res.resize(nb); e_down.resize(nb);
    int ch = boc[0];
    for (int i = 0; i < nb; ++i) {
      e_res[i].resize(L);
      for (int j = 0; j < L; ++j)
        build_resnet("encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? ch : boc[i], boc[i], e_res[i][j]);
      ch = boc[i];
      if (i != nb - 1) build_conv("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", ch, ch, e_down[i]);
    }
    build_mid("encoder.mid_block", boc[nb - 1], e_mid, e_att);
    e_nw = vec("encoder.conv_norm_out.weight", boc[nb - 1]); e_nb = vec("encoder.conv_norm_out.bias", boc[nb - 1]);
    build_conv("encoder.conv_out", 2 * cfg.latent_channels, boc[nb - 1], e_out);
    build_conv1x1("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, quant);
    build_conv1x1("post_quant_conv", cfg.latent_channels, cfg.latent_channels, post_quant);
    build_conv("decoder.conv_in", boc[nb - 1], cfg.latent_channels, d_in);
    build_mid("decoder.mid_block", boc[nb - 1], d_mid, d_att);
    d_res.resize(nb); d_up.resize(nb);
    ch = boc[nb - 1];
    for (int i = 0; i < nb; ++i) {
      const int oc = boc[nb - 1 - i];
      d_res[i].resize(L + 1);
      for (int j = 0; j <= L; ++j)
        build_resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? ch : oc, oc, d_res[i][j]);
      ch = oc;
      if (i != nb - 1) build_conv("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", ch, ch, d_up[i]);
    }
    d_nw = vec("decoder.conv_norm_out.weight", boc[0]); d_nb = vec("decoder.conv_norm_out.bias", boc[0]);
    build_conv("decoder.conv_out", cfg.out_channels, boc[0], d_out);
    return 0;
  }

  int pack(const float* const* master, int count, hipStream_t s) {
    DFH_REQUIRE(count == (int)params.size(), "parameter count mismatch");
    DFH_REQUIRE(arena16 && arena32, "arenas not bound");
    tab_pack.clear(); tab_pack_acc.clear();
    for (const PackOp& op : packs) {
      void* src = (void*)master[op.param];
      DFH_REQUIRE(src != nullptr, "null master parameter: " + params[op.param].name);
      if (op.kind == PK_VEC) (op.accumulate ? tab_pack_acc : tab_pack).add(src, TAB_PACK_VEC, (long)op.dst, op.N, 0, 0, 0, op.accumulate, 0, 0, op.N);
      else if (op.kind == PK_MAT) tab_pack.add(src, TAB_PACK_MAT, (long)op.dst, op.N, op.K, op.ldw, 0, op.col_off, 0, 0, (long)op.N * op.K);
      else tab_pack.add(src, TAB_PACK_CONV, (long)op.dst, op.N, op.K, op.ldw, 0, op.col_off, 0, op.cin_pad, (long)op.N * op.K * 9);
    }
    if (int rc = tab_pack.launch(arena32, arena16, s)) return rc;
    return tab_pack_acc.launch(arena32, arena16, s);
  }

  // ---------------------------------------------------------------- run
  struct Run {
    dfh_vae* u; int B; hipStream_t s; bool dry;
    Bump persist, temp; size_t partial_need = 0;
    float* partial = nullptr; size_t partial_cap = 0; float* gn_partial = nullptr; bf16_t* zero = nullptr;
    int rc = 0;
    bf16_t* w16(const Mat& m) const { return u->arena16 + m.off; }
    float* v32(const Vec& v) const { return u->arena32 + v.off; }
    Tensor talloc(int H, int W, int C) { return Tensor{(bf16_t*)temp.alloc((size_t)B * H * W * C * 2), H, W, C}; }
    Tensor palloc(int H, int W, int C) { return Tensor{(bf16_t*)persist.alloc((size_t)B * H * W * C * 2), H, W, C}; }
    void gemm(GemmArgs g) {
      if (rc) return;
      g.zero = zero; g.partial = partial;
      if (dry) { partial_need = std::max(partial_need, dfh::gemm_partial_floats(g) * sizeof(float)); return; }
      if (dfh::gemm_partial_floats(g) * sizeof(float) > partial_cap) { dfh::set_error("split-K partial buffer too small"); rc = -1; return; }
      rc = dfh::gemm_launch(g, s);
    }
    static GemmArgs base(int M, int N) {
      GemmArgs g; std::memset(&g, 0, sizeof(g));
      g.M = M; g.N = N; g.rows_per_b = M; g.out_mode = OUT_BF16; g.ld_out = N;
      return g;
    }
    void groupnorm(const Tensor& x, const Vec& w, const Vec& b, int silu, Tensor& out) {
      if (rc || dry) return;
      GnArgs a; std::memset(&a, 0, sizeof(a));
      a.src0 = x.p; a.C0 = x.C; a.B = B; a.HW = x.H * x.W; a.G = u->cfg.norm_num_groups;
      a.gamma = v32(w); a.beta = v32(b); a.eps = 1e-6f; a.silu = silu; a.out = out.p; a.partial = gn_partial;
      rc = dfh::groupnorm_launch(a, s);
    }
    // 3x3 conv; mode 0 same size, 1 stride 2 over the right/bottom zero-extended input, 2 fused nearest-2x upsample
    Tensor conv(const Tensor& x, const VConv& c, int mode, bool to_persist, void* f32_nchw_out = nullptr) {
      const int Ho = mode == 2 ? x.H * 2 : (mode == 1 ? x.H / 2 : x.H), Wo = mode == 2 ? x.W * 2 : (mode == 1 ? x.W / 2 : x.W);
      const int ldo = (c.npad + 7) & ~7;      // narrow outputs (4 latent channels) land in a zero-filled 8-channel pixel
      Tensor o{nullptr, Ho, Wo, ldo};
      GemmArgs g = base(B * Ho * Wo, c.npad);
      g.conv_src = x.p; g.conv_c = x.C; g.ntaps = 9;
      g.Hin = x.H; g.Win = x.W; g.Hout = Ho; g.Wout = Wo; g.stride = mode == 1 ? 2 : 1; g.ups = mode == 2 ? 1 : 0; g.pad0 = mode == 1;
      g.W = w16(c.w); g.ldw = c.w.K; g.bias = v32(c.b);
      if (f32_nchw_out) { g.out = f32_nchw_out; g.out_mode = OUT_F32_T; g.ld_out = Ho * Wo; g.rows_per_b = Ho * Wo; }
      else {
        o.p = (bf16_t*)(to_persist ? persist : temp).alloc((size_t)B * Ho * Wo * ldo * 2);
        if (ldo != c.npad && !dry && !rc) (void)hipMemsetAsync(o.p, 0, (size_t)B * Ho * Wo * ldo * 2, s);
        g.out = o.p; g.ld_out = ldo;
      }
      gemm(g);
      return o;
    }
    Tensor resnet(const Tensor& x, const VRes& r) {
      const int H = x.H, W = x.W;
      Tensor out = palloc(H, W, r.cout);
      const size_t mark = temp.off;
      Tensor g1 = talloc(H, W, r.cin);
      groupnorm(x, r.n1w, r.n1b, 1, g1);
      Tensor h1 = talloc(H, W, r.cout);
      {
        GemmArgs g = base(B * H * W, r.cout);
        g.conv_src = g1.p; g.conv_c = r.cin; g.ntaps = 9; g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1;
        g.W = w16(r.w1); g.ldw = r.w1.K; g.bias = v32(r.b1); g.out = h1.p;
        gemm(g);
      }
      Tensor g2 = talloc(H, W, r.cout);
      groupnorm(h1, r.n2w, r.n2b, 1, g2);
      {
        GemmArgs g = base(B * H * W, r.cout);
        g.conv_src = g2.p; g.conv_c = r.cout; g.ntaps = 9; g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1;
        g.W = w16(r.w2); g.ldw = r.w2.K; g.bias = v32(r.b2);
        if (r.shortcut) { g.p_src[0] = x.p; g.p_c[0] = x.C; g.nplain = 1; }
        else { g.resid = x.p; g.ld_res = r.cout; }
        g.out = out.p;
        gemm(g);
      }
      temp.off = mark;
      return out;
    }
    void linear(const bf16_t* x, int M, int K, const Mat& W, const Vec* bias, const bf16_t* resid, void* out, int N, int out_mode = OUT_BF16,
                int ld_out = -1, int rows_per_b = 0) {
      GemmArgs g = base(M, N);
      g.p_src[0] = x; g.p_c[0] = K; g.nplain = 1;
      g.W = w16(W); g.ldw = W.K; g.bias = bias ? v32(*bias) : nullptr; g.resid = resid; g.ld_res = N;
      g.out = out; g.out_mode = out_mode; if (ld_out >= 0) g.ld_out = ld_out; if (rows_per_b) g.rows_per_b = rows_per_b;
      gemm(g);
    }
    Tensor attention(const Tensor& x, const VAtt& a) {
      const int H = x.H, W = x.W, C = a.C, N = H * W, M = B * N, Np = (N + 7) & ~7;
      Tensor out = palloc(H, W, C);
      const size_t mark = temp.off;
      Tensor t = talloc(H, W, C);
      groupnorm(x, a.gw, a.gb, 0, t);
      Tensor q = talloc(H, W, C), k = talloc(H, W, C), o = talloc(H, W, C);
      bf16_t* vt = (bf16_t*)temp.alloc((size_t)B * C * Np * 2);
      float* S = (float*)temp.alloc((size_t)N * Np * 4);
      bf16_t* P = (bf16_t*)temp.alloc((size_t)N * Np * 2);
      linear(t.p, M, C, a.q, &a.qb, nullptr, q.p, C);
      linear(t.p, M, C, a.k, &a.kb, nullptr, k.p, C);
      if (!dry && !rc && Np != N) (void)hipMemsetAsync(vt, 0, (size_t)B * C * Np * 2, s);
      linear(t.p, M, C, a.v, &a.vb, nullptr, vt, C, OUT_BF16_T, Np, N);
      if (!dry && !rc && Np != N) (void)hipMemsetAsync(P, 0, (size_t)N * Np * 2, s);
      for (int b = 0; b < B; ++b) {
        {   // S = Q_b K_b^T (fp32)
          GemmArgs g = base(N, N);
          g.p_src[0] = q.p + (size_t)b * N * C; g.p_c[0] = C; g.nplain = 1;
          g.W = k.p + (size_t)b * N * C; g.ldw = C;
          g.out = S; g.out_mode = OUT_F32; g.ld_out = Np;
          if (N % 4 == 0) gemm(g); else { dfh::set_error("VAE attention needs a multiple of 4 tokens"); rc = -1; }
        }
        if (!dry && !rc) {
          hipLaunchKernelGGL(softmax_rows_kernel, dim3(N), dim3(256), 0, s, S, P, N, Np, 1.0f / sqrtf((float)C));
          rc = dfh::check_launch("softmax_rows_kernel");
        }
        {   // O_b = P V_b   (W operand = V_b^T [C][Np], contraction over the keys)
          GemmArgs g = base(N, C);
          g.p_src[0] = P; g.p_c[0] = Np; g.nplain = 1;
          g.W = vt + (size_t)b * C * Np; g.ldw = Np;
          g.out = o.p + (size_t)b * N * C;
          gemm(g);
        }
      }
      linear(o.p, M, C, a.o, &a.ob, x.p, out.p, C);
      temp.off = mark;
      return out;
    }
  };

  // plan + layout shared by encode / decode
  int run(bool encode, const float* in, float* out, int B, int size, hipStream_t s, bool dry, size_t* need) {
    Run r; r.u = this; r.B = B; r.s = s; r.dry = dry;
    Bump head; head.base = dry ? nullptr : ws;
    r.zero = (bf16_t*)head.alloc(256);
    r.gn_partial = (float*)head.alloc((size_t)B * GN_MAX_CHUNKS * 64 * 2 * sizeof(float));
    // region sizes come from a dry pass of the same walk (plan_* below)
    r.partial = (float*)head.alloc(dry ? 0 : plan_partial); r.partial_cap = dry ? 0 : plan_partial;
    const size_t head_bytes = (head.off + 255) & ~(size_t)255;
    if (!dry) {
      r.persist.base = ws + head_bytes; r.temp.base = ws + head_bytes + plan_persist;
      if (head_bytes + plan_persist + plan_temp > ws_bytes) { dfh::set_error("VAE workspace too small"); return -1; }
      (void)hipMemsetAsync(r.zero, 0, 256, s);
    }
    const int nb = cfg.num_blocks, L = cfg.layers_per_block;
    if (encode) {
      Tensor x = r.palloc(size, size, e_in.cin);
      if (!dry) r.rc = dfh::nchw_to_nhwc_launch(in, 0, x.p, B, cfg.in_channels, size * size, s);
      Tensor h = r.conv(x, e_in, 0, true);
      for (int i = 0; i < nb; ++i) {
        for (int j = 0; j < L; ++j) h = r.resnet(h, e_res[i][j]);
        if (i != nb - 1) h = r.conv(h, e_down[i], 1, true);
      }
      h = r.resnet(h, e_mid[0]); h = r.attention(h, e_att); h = r.resnet(h, e_mid[1]);
      Tensor g = r.palloc(h.H, h.W, h.C);
      r.groupnorm(h, e_nw, e_nb, 1, g);
      Tensor m = r.conv(g, e_out, 0, true);                         // [M][2L] bf16 (2L = 8)
      // quant_conv 1x1 -> fp32 NCHW moments
      GemmArgs q = Run::base(B * m.H * m.W, quant.npad);
      q.p_src[0] = m.p; q.p_c[0] = m.C; q.nplain = 1;
      q.W = r.w16(quant.w); q.ldw = quant.w.K; q.bias = r.v32(quant.b);
      q.out = out; q.out_mode = OUT_F32_T; q.ld_out = m.H * m.W; q.rows_per_b = m.H * m.W;
      r.gemm(q);
    } else {
      Tensor z = r.palloc(size, size, post_quant.cin);
      if (!dry) r.rc = dfh::nchw_to_nhwc_launch(in, 0, z.p, B, cfg.latent_channels, size * size, s);
      // post_quant_conv 1x1 into a zero-filled 8-channel pixel
      Tensor pq = r.palloc(size, size, d_in.cin);
      if (!dry && !r.rc) (void)hipMemsetAsync(pq.p, 0, (size_t)B * size * size * pq.C * 2, s);
      {
        GemmArgs g = Run::base(B * size * size, post_quant.npad);
        g.p_src[0] = z.p; g.p_c[0] = z.C; g.nplain = 1;
        g.W = r.w16(post_quant.w); g.ldw = post_quant.w.K; g.bias = r.v32(post_quant.b);
        g.out = pq.p; g.ld_out = pq.C;
        r.gemm(g);
      }
      Tensor h = r.conv(pq, d_in, 0, true);
      h = r.resnet(h, d_mid[0]); h = r.attention(h, d_att); h = r.resnet(h, d_mid[1]);
      for (int i = 0; i < nb; ++i) {
        for (int j = 0; j <= L; ++j) h = r.resnet(h, d_res[i][j]);
        if (i != nb - 1) h = r.conv(h, d_up[i], 2, true);
      }
      Tensor g = r.palloc(h.H, h.W, h.C);
      r.groupnorm(h, d_nw, d_nb, 1, g);
      r.conv(g, d_out, 0, true, out);                               // fp32 NCHW, out_channels padded to 4
    }
    if (dry) {
      plan_persist = (r.persist.peak + 255) & ~(size_t)255;
      plan_temp = (r.temp.peak + 255) & ~(size_t)255;
      plan_partial = (r.partial_need + 255) & ~(size_t)255;
      Bump hd; hd.alloc(256); hd.alloc((size_t)B * GN_MAX_CHUNKS * 64 * 2 * sizeof(float)); hd.alloc(plan_partial);
      if (need) *need = ((hd.off + 255) & ~(size_t)255) + plan_persist + plan_temp;
    }
    return r.rc;
  }
  size_t plan_persist = 0, plan_temp = 0, plan_partial = 0;
};

// ------------------------------------------------------------------------------------------- C ABI
extern "C" {

int dfh_vae_create(const dfh_vae_config* cfg, dfh_vae** out) {
  DFH_REQUIRE(cfg && out, "null argument");
  DFH_REQUIRE(cfg->num_blocks >= 2 && cfg->num_blocks <= DFH_MAX_BLOCKS, "num_blocks out of range");
  DFH_REQUIRE(cfg->in_channels > 0 && cfg->in_channels <= 8 && cfg->latent_channels > 0 && cfg->latent_channels <= 4, "channel counts out of range");
  for (int i = 0; i < cfg->num_blocks; ++i)
    DFH_REQUIRE(cfg->block_out_channels[i] % 8 == 0 && cfg->block_out_channels[i] % cfg->norm_num_groups == 0,
                "block_out_channels must be multiples of 8 and of norm_num_groups");
  dfh_vae* u = new dfh_vae();
  u->cfg = *cfg;
  u->build();
  *out = u;
  return 0;
}
void dfh_vae_destroy(dfh_vae* u) { delete u; }
int dfh_vae_num_params(const dfh_vae* u) { return (int)u->params.size(); }
const char* dfh_vae_param_name(const dfh_vae* u, int i) { return u->params[i].name.c_str(); }
int dfh_vae_param_ndim(const dfh_vae* u, int i) { return (int)u->params[i].shape.size(); }
int dfh_vae_param_dim(const dfh_vae* u, int i, int d) { return u->params[i].shape[d]; }
size_t dfh_vae_arena16_bytes(const dfh_vae* u) { return u->a16 * 2 + 256; }
size_t dfh_vae_arena32_bytes(const dfh_vae* u) { return u->a32 * 4 + 256; }

size_t dfh_vae_workspace_bytes(dfh_vae* u, int encode, int batch, int size) {
  size_t need = 0;
  if (batch <= 0 || size <= 0) return 0;
  u->run(encode != 0, nullptr, nullptr, batch, size, nullptr, true, &need);
  return need;
}
int dfh_vae_bind(dfh_vae* u, void* arena16, void* arena32, void* workspace, size_t workspace_bytes) {
  DFH_REQUIRE(u && arena16 && arena32 && workspace, "null argument");
  DFH_REQUIRE(((uintptr_t)arena16 | (uintptr_t)arena32 | (uintptr_t)workspace) % 256 == 0, "buffers must be 256-byte aligned");
  u->arena16 = (bf16_t*)arena16; u->arena32 = (float*)arena32; u->ws = (char*)workspace; u->ws_bytes = workspace_bytes;
  return 0;
}
int dfh_vae_pack(dfh_vae* u, const float* const* master_params, int count, void* stream) {
  DFH_REQUIRE(u && master_params, "null argument");
  return u->pack(master_params, count, (hipStream_t)stream);
}
static int vae_run(dfh_vae* u, bool encode, const float* in, float* out, int batch, int size, void* stream) {
  DFH_REQUIRE(u && in && out, "null argument");
  DFH_REQUIRE(u->ws != nullptr, "dfh_vae_bind not called");
  DFH_REQUIRE(batch > 0 && size > 0, "empty batch");
  const int f = 1 << (u->cfg.num_blocks - 1);
  DFH_REQUIRE(!encode || size % f == 0, "image size must be divisible by the down-sampling factor");
  size_t need = 0;
  u->run(encode, nullptr, nullptr, batch, size, nullptr, true, &need);
  DFH_REQUIRE(need <= u->ws_bytes, "workspace smaller than dfh_vae_workspace_bytes for this batch / size");
  return u->run(encode, in, out, batch, size, (hipStream_t)stream, false, nullptr);
}
int dfh_vae_encode(dfh_vae* u, const float* images, float* moments, int batch, int image_size, void* stream) {
  return vae_run(u, true, images, moments, batch, image_size, stream);
}
int dfh_vae_decode(dfh_vae* u, const float* latents, float* images, int batch, int latent_size, void* stream) {
  return vae_run(u, false, latents, images, batch, latent_size, stream);
}

}  // extern "C"
