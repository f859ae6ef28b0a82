// Training step of the conditional U-Net: forward that keeps what the backward needs, and the backward walk.
//
// Replaces (arithmetic): torch autograd through diffusers UNet2DConditionModel as driven by the reference at
// DiFashion/models/difashion.py:249-253 (forward in training_step) and DiFashion/train.py:699
// (accelerator.backward(loss)); SURVEY.md 8a rows a2 / a12.
//
// MI355X-first choices (DESIGN.md "Training path"):
//   * no autograd graph: the forward walk appends one closure per layer to a tape, the backward runs the tape in
//     reverse.  Every closure is a handful of launches of the SAME gfx950 kernels as the forward pass:
//       data gradients  = the implicit-GEMM kernel against transposed/flipped weight packs (arena16t), with
//                         stride-2 convs as zero-insertion, upsample convs as dgrad + 2x2 sum pool, and gradient
//                         accumulation into an already-written buffer through the residual epilogue;
//       weight gradients = gemm_wgrad_kernel over the forward K-segment description, fp32 atomics straight into a
//                         gradient arena laid out like the packed weights (no per-layer temporaries);
//       norm / attention = their dedicated backward kernels (GroupNorm from saved mean/rstd, attention from LSE).
//   * 288 GB HBM: every activation of the step is kept (no recomputation / gradient checkpointing); layer-internal
//     gradient buffers come from a stack allocator and overlap across layers.
//   * nothing is allocated per step: a dry run of the same forward + tape sizes the workspace.
#include <array>
#include <functional>
#include <memory>

#include "unet_model.h"
#include "bwd_elementwise.h"
#include "wgrad.h"

namespace {
struct GT { bf16_t* p = nullptr; bf16_t* g = nullptr; int H = 0, W = 0, C = 0; int gid = -1; };
}

struct dfh_unet::TrainRun {
  dfh_unet* u = nullptr; int B = 0; hipStream_t s = nullptr; bool dry = true;
  Bump persist, gtemp;
  size_t partial_need = 0; float* partial = nullptr; size_t partial_cap = 0;
  float* partial2 = nullptr;                                   // slab region of the weight-gradient launches on the side stream
  hipStream_t s2 = nullptr; bool forked = false;               // see wgrad() / join()
  float* gn_partial = nullptr; bf16_t* zero = nullptr;
  int rc = 0;
  std::vector<std::function<void()>> tape;
  std::vector<char> gstate;
  const float* d_out = nullptr; float* d_sample = nullptr;   // set by backward()
  float* dtemb_all = nullptr; size_t dtemb_bytes = 0;
  // weight-gradient write ranges per tape entry (recorded by the dry run of the tape): a bucket of grad16 is final once the
  // LOWEST tape index that writes into it has run (the tape runs from the back)
  int cur_entry = -1;
  std::vector<std::array<size_t, 3>> writes;                  // {tape index, first float, end float}
  // segmented backward
  size_t bucket = 0; int next_entry = -1;
  std::vector<int> bucket_last;                                // per bucket: lowest tape index writing into it (-1: never written)
  std::vector<std::pair<size_t, size_t>> ready;                // finished ranges not handed out yet

  bf16_t* w16(const Mat& m) const { return u->arena16 + m.off; }
  bf16_t* w16t(const Mat& m) const { return u->arena16t + m.off; }
  float* v32(const Vec& v) const { return u->arena32 + v.off; }
  float* g32(const Vec& v) const { return u->grad32 + v.off; }

  bf16_t* buf(size_t elems) { return (bf16_t*)persist.alloc(elems * 2); }
  float* fbuf(size_t elems) { return (float*)persist.alloc(elems * 4); }
  bf16_t* gbuf(size_t elems) { return (bf16_t*)gtemp.alloc(elems * 2); }
  // activation + gradient buffer; io tensors (layer inputs / outputs) keep their gradient across layers
  GT act(int H, int W, int C, bool io = false) {
    GT t; t.H = H; t.W = W; t.C = C;
    const size_t n = (size_t)B * H * W * C;
    t.p = buf(n);
    t.g = io ? buf(n) : gbuf(n);
    t.gid = (int)gstate.size();
    gstate.push_back(0);
    return t;
  }
  // true when the gradient buffer already holds a contribution (the next writer must accumulate)
  bool acc(const GT& t) { const bool a = gstate[t.gid] != 0; gstate[t.gid] = 1; return a; }

  // ------------------------------------------------------------------ kernel wrappers
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
  // out = x . W^T + bias (+resid); returns the descriptor (the weight-gradient GEMM re-reads its K segments)
  GemmArgs linear(const bf16_t* x, int M, int K, const Mat& W, const Vec* bias, const bf16_t* resid, void* out, int N,
                  int out_mode = OUT_BF16) {
    GemmArgs g = base(M, N);
    g.p_src[0] = x; g.p_c[0] = K; g.nplain = 1;
    g.W = w16(W); g.ldw = W.K;
    g.bias = bias ? v32(*bias) : nullptr;
    g.resid = resid; g.ld_res = N;
    g.out = out; g.out_mode = out_mode;
    gemm(g);
    return g;
  }
  GemmArgs conv_desc(const bf16_t* src, int C, int Hin, int Win, int Hout, int Wout, int stride, int ups, int N) {
    GemmArgs g = base(B * Hout * Wout, N);
    g.conv_src = src; g.conv_c = C; g.ntaps = 9;
    g.Hin = Hin; g.Win = Win; g.Hout = Hout; g.Wout = Wout; g.stride = stride; g.ups = ups;
    return g;
  }
  // dW (packed fp32 gradient of the matrix at arena16 + w_off) += dY^T . A, A described by the forward descriptor
  void wgrad(const GemmArgs& f, const bf16_t* dY, int ldy, size_t w_off, const Vec* bias = nullptr) {
    if (rc) return;
    WgradArgs w; std::memset(&w, 0, sizeof(w));
    w.conv_src = f.conv_src; w.conv_c = f.conv_c; w.ntaps = f.ntaps;
    w.Hin = f.Hin; w.Win = f.Win; w.Hout = f.Hout; w.Wout = f.Wout; w.stride = f.stride; w.ups = f.ups;
    w.p_src[0] = f.p_src[0]; w.p_src[1] = f.p_src[1]; w.p_c[0] = f.p_c[0]; w.p_c[1] = f.p_c[1]; w.nplain = f.nplain;
    w.dY = dY; w.ldy = ldy; w.zero = zero; w.M = f.M; w.N = f.N;
    w.dW = u->grad16 + w_off; w.ldw = f.ldw; w.msplit = 0;
    w.dbias = bias ? g32(*bias) : nullptr;
    w.overwrite = 1;          // every packed matrix has exactly one weight-gradient launch per backward: no memset, no RMW
    w.partial = partial; w.partial_cap = partial_cap / sizeof(float);     // shares the split-K slab region of the GEMMs
    if (dry) {
      partial_need = std::max(partial_need, dfh::wgrad_partial_floats(w) * sizeof(float));
      if (cur_entry >= 0) writes.push_back({(size_t)cur_entry, w_off, w_off + (size_t)f.N * f.ldw});
      return;
    }
    // A large weight-gradient GEMM goes to the side stream: it only READS dY and the saved forward operand, as does the data-gradient
    // GEMM that follows on the main stream, so the two run side by side and each fills the other's tail round (576 tiles on 512
    // resident slots leave 64 blocks alone on the chip).  join() orders the main stream behind it: before anything writes a buffer the
    // launch reads, and at the end of every tape entry (gradient ranges are handed out / temporaries are reused per entry).
    double kreal = (double)f.ntaps * f.conv_c + f.p_c[0] + (f.nplain > 1 ? f.p_c[1] : 0);
    // DFH_TRAIN_SIDE_MIN_FLOP: smaller launches stay on the main stream (two event operations cost more than their tail); tests set it to 0
    static const double side_min = [] { const char* e = getenv("DFH_TRAIN_SIDE_MIN_FLOP"); return e ? atof(e) : 2e10; }();
    if (s2 && 2.0 * f.M * f.N * kreal >= side_min) {
      (void)hipEventRecord(u->ev_fork, s);                     // everything dY depends on
      (void)hipStreamWaitEvent(s2, u->ev_fork, 0);
      w.partial = partial2;
      rc = dfh::wgrad_launch(w, s2);
      forked = true;
      return;
    }
    rc = dfh::wgrad_launch(w, s);
  }
  void join() {
    if (!forked || dry) return;
    (void)hipEventRecord(u->ev_join, s2);
    (void)hipStreamWaitEvent(s, u->ev_join, 0);
    forked = false;
  }
  void colsum(const bf16_t* Y, int ldy, int N, int groups, int rows_per_group, float* out, int ld_out) {
    if (rc || dry) return;
    rc = dfh::colsum_launch(Y, ldy, N, groups, rows_per_group, out, ld_out, s);
  }
  // dX[M][Nout] (=|+=) dY[M][Kin] . Wt^T   with Wt = [Nout][ldw] transposed pack (up to two dY segments)
  void dgrad_linear(const bf16_t* dY, int M, int Kin, const bf16_t* Wt, int ldw, int Nout, bf16_t* out, bool accumulate,
                    const bf16_t* dY1 = nullptr, int Kin1 = 0) {
    GemmArgs g = base(M, Nout);
    g.p_src[0] = dY; g.p_c[0] = Kin; g.nplain = 1;
    if (dY1) { g.p_src[1] = dY1; g.p_c[1] = Kin1; g.nplain = 2; }
    g.W = Wt; g.ldw = ldw; g.out = out;
    if (accumulate) { g.resid = out; g.ld_res = Nout; }
    gemm(g);
  }
  // dX = conv3x3(dY, flipped/transposed W); mode 0: same resolution, 2: dY zero-inserted 2x (backward of stride 2)
  void dgrad_conv(const bf16_t* dY, int Cd, int Hd, int Wd, int mode, const Mat& Wt, int Nout, bf16_t* out, bool accumulate) {
    const int Ho = mode == 2 ? Hd * 2 : Hd, Wo = mode == 2 ? Wd * 2 : Wd;
    GemmArgs g = conv_desc(dY, Cd, Hd, Wd, Ho, Wo, 1, mode, Nout);
    g.W = w16t(Wt); g.ldw = Wt.K; g.out = out;
    if (accumulate) { g.resid = out; g.ld_res = Nout; }
    gemm(g);
  }
  void groupnorm(const GT& x0, const GT* x1, const Vec& w, const Vec& b, float eps, int silu, bf16_t* out, float* stats) {
    if (rc || dry) return;
    GnArgs a; std::memset(&a, 0, sizeof(a));
    a.src0 = x0.p; a.C0 = x0.C; a.src1 = x1 ? x1->p : nullptr; a.C1 = x1 ? x1->C : 0;
    a.B = B; a.HW = x0.H * x0.W; a.G = u->cfg.norm_num_groups;
    a.gamma = v32(w); a.beta = v32(b); a.eps = eps; a.silu = silu; a.out = out; a.partial = gn_partial; a.stats_out = stats;
    rc = dfh::groupnorm_launch(a, s);
  }
  void groupnorm_bwd(const GT& x0, const GT* x1, const bf16_t* dy, const Vec& w, const Vec& b, const float* stats, int silu) {
    const bool a0 = acc(x0), a1 = x1 ? acc(*x1) : false;
    if (rc || dry) return;
    GnBwdArgs a; std::memset(&a, 0, sizeof(a));
    a.src0 = x0.p; a.C0 = x0.C; a.src1 = x1 ? x1->p : nullptr; a.C1 = x1 ? x1->C : 0;
    a.dy = dy; a.B = B; a.HW = x0.H * x0.W; a.G = u->cfg.norm_num_groups;
    a.gamma = v32(w); a.beta = v32(b); a.stats = stats; a.silu = silu;
    a.dx0 = x0.g; a.dx1 = x1 ? x1->g : nullptr; a.acc0 = a0; a.acc1 = a1;
    a.dgamma = g32(w); a.dbeta = g32(b); a.partial = gn_partial;
    rc = dfh::groupnorm_bwd_launch(a, s);
  }
  void layernorm(const bf16_t* x, const Vec& w, const Vec& b, bf16_t* y, int M, int C) {
    if (rc || dry) return;
    rc = dfh::layernorm_launch(x, v32(w), v32(b), y, M, C, 1e-5f, s);
  }
  void layernorm_bwd(const bf16_t* x, const bf16_t* dy, const Vec& w, const Vec& b, bf16_t* dx, int accumulate, int M, int C) {
    if (rc || dry) return;
    rc = dfh::layernorm_bwd_launch(x, dy, v32(w), dx, accumulate, g32(w), g32(b), M, C, 1e-5f, s);
  }
  void attention(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const bf16_t* Vt, int ldvt, bf16_t* O, int C, int heads,
                 int Nq, int Nk, long vt_bstride, float* lse) {
    if (rc || dry) return;
    AttnArgs a; std::memset(&a, 0, sizeof(a));
    a.vt_bstride = vt_bstride;
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.Vt = Vt; a.ldvt = ldvt; a.O = O; a.ldo = C;
    a.B = B; a.H = heads; a.D = C / heads; a.Nq = Nq; a.Nk = Nk;
    a.scale = 1.0f / sqrtf((float)a.D); a.lse = lse;
    rc = dfh::attention_launch(a, s);
  }
  void attention_bwd(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const bf16_t* V, int ldv, const bf16_t* O,
                     const bf16_t* dO, int C, const float* lse, float* delta, bf16_t* dQ, int lddq, bf16_t* dK, int lddk,
                     bf16_t* dV, int lddv, int heads, int Nq, int Nk) {
    if (rc || dry) return;
    const int D = C / heads;
    rc = dfh::attention_delta_launch(O, dO, C, delta, B, heads, D, Nq, s);
    if (rc) return;
    AttnBwdArgs a; std::memset(&a, 0, sizeof(a));
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.dO = dO; a.ldo = C; a.lse = lse; a.delta = delta;
    a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
    a.B = B; a.H = heads; a.D = D; a.Nq = Nq; a.Nk = Nk; a.scale = 1.0f / sqrtf((float)D);
    rc = dfh::attention_bwd_launch(a, s);
  }
  void transpose(const bf16_t* in, bf16_t* out, int R, int C, int ld_in, int ld_out, long in_bs, long out_bs) {
    if (rc || dry) return;
    if (ld_out != R) (void)hipMemsetAsync(out, 0, (size_t)B * out_bs * 2, s);   // padded key columns must read as 0
    rc = dfh::transpose_bf16_launch(in, out, B, R, C, ld_in, ld_out, in_bs, out_bs, s);
  }
#define TR_OP(call) do { if (!rc && !dry) rc = (call); } while (0)

  // ------------------------------------------------------------------ layers
  // Upsample2D (nearest 2x, then conv3x3) through its four PHASE PLANES, forward AND backward (gemm.h GemmArgs::phase2x; the inference
  // walk has run the forward this way since round 3): output pixel (2y + py, 2x + px) sees a 2x2 neighbourhood of the SOURCE image with
  // the summed taps, so forward, data gradient and weight gradient each execute 4/9 of the multiply-adds of the conv over the upsampled
  // image, and neither the upsampled tensor nor its gradient exists.
  //   forward : one launch over the four planes with the summed weights WP [4][Co][4 Ci] (re-derived every step: the weights move)
  //   dW      : the output gradient gathered phase-major, one weight-gradient launch per plane over the source image (WgradArgs::tap2)
  //             into dWP [4][Co][4 Ci], un-folded onto the packed 3x3 gradient (each 3x3 tap sums the four slots it was folded into)
  //   dX      : one launch over the four planes (phase2x = 2: mirrored taps, transposed weights WPT [4][Ci][4 Co]), the four planes summed
  GT conv_up_phase(const GT& x, const ConvL& c) {
    const int H = x.H, W = x.W, Ci = x.C, Co = c.cout, M = B * H * W;
    GT o = act(2 * H, 2 * W, Co, true);
    bf16_t* WP = buf((size_t)16 * Co * Ci);
    bf16_t* WPT = buf((size_t)16 * Co * Ci);
    TR_OP(dfh::ups_phase_fold_launch(w16(c.w), c.w.K, WP, Co, Ci, s));
    for (int p = 0; p < 4; ++p)      // the four slots of a plane are [Co][Ci] blocks side by side: transposed block by block
      TR_OP(dfh::transpose_bf16_launch(WP + (size_t)p * Co * 4 * Ci, WPT + (size_t)p * Ci * 4 * Co, 4, Co, Ci, 4 * Ci, 4 * Co, Ci, Co, s));
    GemmArgs f = base(M, Co);
    f.conv_src = x.p; f.conv_c = Ci; f.ntaps = 4; f.phase2x = 1; f.nbatch = 4; f.w_bs = (long)Co * 4 * Ci;
    f.Hin = H; f.Win = W; f.Hout = H; f.Wout = W; f.stride = 1; f.rows_per_b = H * W;
    f.W = WP; f.ldw = 4 * Ci; f.bias = v32(c.b); f.out = o.p;
    gemm(f);
    const size_t mark = gtemp.off;
    bf16_t* dyp = gbuf((size_t)4 * M * Co);
    bf16_t* dxp = gbuf((size_t)4 * M * Ci);
    float* dwp = (float*)gtemp.alloc((size_t)16 * Co * Ci * sizeof(float));
    gtemp.off = mark;
    const ConvL* cp = &c;
    tape.push_back([=] {
      TR_OP(dfh::phase_gather_launch(o.g, dyp, B, H, W, Co, s));
      for (int p = 0; p < 4 && !rc; ++p) {
        WgradArgs w; std::memset(&w, 0, sizeof(w));
        w.conv_src = x.p; w.conv_c = Ci; w.ntaps = 4; w.tap2 = 1; w.tap_py = p >> 1; w.tap_px = p & 1;
        w.Hin = H; w.Win = W; w.Hout = H; w.Wout = W; w.stride = 1;
        w.dY = dyp + (size_t)p * M * Co; w.ldy = Co; w.zero = zero; w.M = M; w.N = Co;
        w.dW = dwp + (size_t)p * Co * 4 * Ci; w.ldw = 4 * Ci; w.overwrite = 1; w.dbias = g32(cp->b);
        w.partial = partial; w.partial_cap = partial_cap / sizeof(float);
        if (dry) { partial_need = std::max(partial_need, dfh::wgrad_partial_floats(w) * sizeof(float)); continue; }
        rc = dfh::wgrad_launch(w, s);
      }
      if (dry && cur_entry >= 0) writes.push_back({(size_t)cur_entry, cp->w.off, cp->w.off + (size_t)Co * cp->w.K});
      TR_OP(dfh::ups_phase_unfold_launch(dwp, u->grad16 + cp->w.off, Co, Ci, cp->w.K, 1, s));
      GemmArgs g = base(M, Ci);
      g.conv_src = dyp; g.conv_c = Co; g.ntaps = 4; g.phase2x = 2; g.nbatch = 4;
      g.a_bs = (long)M * Co; g.w_bs = (long)Ci * 4 * Co; g.o_bs = (long)M * Ci;
      g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1; g.rows_per_b = H * W;
      g.W = WPT; g.ldw = 4 * Co; g.out = dxp;
      gemm(g);
      const bool a = acc(x);
      TR_OP(dfh::phase_sum4_launch(dxp, x.g, (long)M * Ci, a ? 1 : 0, s));
    });
    return o;
  }

  GT conv(const GT& x, const ConvL& c, int stride, int ups) {
    // DFH_TRAIN_UPS_PHASE=0: the upsample convs as one 3x3 conv over the virtual upsampled image + 2x2 sum pool in the backward (A/B)
    static const bool ph_off = [] { const char* e = getenv("DFH_TRAIN_UPS_PHASE"); return e && e[0] == '0'; }();
    if (ups == 1 && !ph_off && x.C % 8 == 0 && c.cout % 8 == 0 && x.C == c.cin) return conv_up_phase(x, c);
    const int Ho = ups ? x.H * 2 : (stride == 2 ? x.H / 2 : x.H);
    const int Wo = ups ? x.W * 2 : (stride == 2 ? x.W / 2 : x.W);
    GT o = act(Ho, Wo, c.cout, true);
    GemmArgs f = conv_desc(x.p, x.C, x.H, x.W, Ho, Wo, stride, ups, c.cout);
    f.W = w16(c.w); f.ldw = c.w.K; f.bias = v32(c.b); f.out = o.p;
    gemm(f);
    const size_t mark = gtemp.off;
    bf16_t* full = ups ? gbuf((size_t)B * Ho * Wo * x.C) : nullptr;      // dgrad at 2H before the 2x2 sum pool
    bf16_t* pooled = ups ? gbuf((size_t)B * x.H * x.W * x.C) : nullptr;
    gtemp.off = mark;
    const ConvL* cp = &c;
    tape.push_back([=] {
      const int M = B * Ho * Wo;
      wgrad(f, o.g, cp->cout, cp->w.off, &cp->b);
      if (ups) {
        dgrad_conv(o.g, cp->cout, Ho, Wo, 0, cp->wt, x.C, full, false);
        const bool a = acc(x);
        TR_OP(dfh::pool2x2_sum_launch(full, a ? pooled : x.g, B, x.H, x.W, x.C, s));
        if (a) TR_OP(dfh::add_bf16_launch(x.g, pooled, (long)B * x.H * x.W * x.C, 1, s));
      } else {
        dgrad_conv(o.g, cp->cout, Ho, Wo, stride == 2 ? 2 : 0, cp->wt, x.C, x.g, acc(x));
      }
    });
    return o;
  }

  GT resnet(const GT& x0, const GT* x1p, const ResL& r, const float* temb_all) {
    const int H = x0.H, W = x0.W, M = B * H * W;
    const int G = u->cfg.norm_num_groups;
    const bool has1 = x1p != nullptr;
    const GT x1 = has1 ? *x1p : GT{};
    GT out = act(H, W, r.cout, true);
    const size_t mark = gtemp.off;
    GT g1 = act(H, W, r.cin);
    float* st1 = fbuf((size_t)B * G * 2);
    groupnorm(x0, x1p, r.n1w, r.n1b, u->cfg.norm_eps, 1, g1.p, st1);
    GT h1 = act(H, W, r.cout);
    GemmArgs f1 = conv_desc(g1.p, r.cin, H, W, H, W, 1, 0, r.cout);
    f1.W = w16(r.w1); f1.ldw = r.w1.K; f1.bias = v32(r.b1);
    f1.rowvec = temb_all; f1.rv_ld = u->temb_total; f1.rv_off = r.temb_off; f1.rows_per_b = H * W;
    f1.out = h1.p;
    gemm(f1);
    GT g2 = act(H, W, r.cout);
    float* st2 = fbuf((size_t)B * G * 2);
    groupnorm(h1, nullptr, r.n2w, r.n2b, u->cfg.norm_eps, 1, g2.p, st2);
    GemmArgs f2 = conv_desc(g2.p, r.cout, H, W, H, W, 1, 0, r.cout);
    f2.W = w16(r.w2); f2.ldw = r.w2.K; f2.bias = v32(r.b2);
    if (r.shortcut) {
      f2.p_src[0] = x0.p; f2.p_c[0] = x0.C; f2.nplain = 1;
      if (has1) { f2.p_src[1] = x1.p; f2.p_c[1] = x1.C; f2.nplain = 2; }
    } else {
      f2.resid = x0.p; f2.ld_res = r.cout;
    }
    f2.out = out.p;
    gemm(f2);
    gtemp.off = mark;
    const ResL* rp = &r;
    tape.push_back([=] {
      const ResL& r = *rp;
      // out = conv2(g2) + shortcut(x)  |  conv2(g2) + x
      wgrad(f2, out.g, r.cout, r.w2.off, &r.b2);
      dgrad_conv(out.g, r.cout, H, W, 0, r.w2t, r.cout, g2.g, false);
      if (r.shortcut) {
        dgrad_linear(out.g, M, r.cout, w16t(r.wst), r.cout, x0.C, x0.g, acc(x0));
        if (has1) dgrad_linear(out.g, M, r.cout, w16t(r.wst) + (size_t)x0.C * r.cout, r.cout, x1.C, x1.g, acc(x1));
      } else {
        const bool a = acc(x0);
        TR_OP(dfh::add_bf16_launch(x0.g, out.g, (long)M * r.cout, a, s));
      }
      groupnorm_bwd(h1, nullptr, g2.g, r.n2w, r.n2b, st2, 1);
      // h1 = conv1(g1) + b1 + temb[b]
      wgrad(f1, h1.g, r.cout, r.w1.off, &r.b1);
      colsum(h1.g, r.cout, r.cout, B, H * W, dtemb_all + r.temb_off, u->temb_total);
      dgrad_conv(h1.g, r.cout, H, W, 0, r.w1t, r.cin, g1.g, false);
      groupnorm_bwd(x0, has1 ? &x1 : nullptr, g1.g, r.n1w, r.n1b, st1, 1);
    });
    return out;
  }

  GT transformer(const GT& x, const AttL& a, const bf16_t* kx, const bf16_t* vx, const bf16_t* vxt, bf16_t* dkx, bf16_t* dvx, int T) {
    const int H = x.H, W = x.W, C = a.C, N = H * W, M = B * N, heads = a.heads;
    const int G = u->cfg.norm_num_groups, XT = u->x_total;
    GT out = act(H, W, C, true);
    const size_t mark = gtemp.off;
    const size_t MC = (size_t)M * C;
    GT gn = act(H, W, C);
    float* st = fbuf((size_t)B * G * 2);
    groupnorm(x, nullptr, a.nw, a.nb, 1e-6f, 0, gn.p, st);
    bf16_t *h0 = buf(MC), *h1 = buf(MC), *h2 = buf(MC), *h3 = buf(MC);
    bf16_t* gh = gbuf(MC);                       // one gradient buffer walks the residual stream h3 -> h0
    GemmArgs f_pin = linear(gn.p, M, C, a.pin, &a.pinb, nullptr, h0, C);
    // --- self attention
    GT n1 = act(H, W, C);
    layernorm(h0, a.l1w, a.l1b, n1.p, M, C);
    // q | k | v in one [M][3C] buffer (the three matrices are packed back to back: build_attn): one projection, one weight-gradient
    // launch over N = 3C, one data-gradient launch over K = 3C
    if (a.v.off != a.qk.off + (size_t)2 * C * C || a.v.K != a.qk.K) { dfh::set_error("to_q | to_k | to_v are not packed back to back"); rc = -1; }
    GT qkv = act(H, W, 3 * C);
    GemmArgs f_qkv = linear(n1.p, M, C, a.qk, nullptr, nullptr, qkv.p, 3 * C);
    const int Np = (N + 7) & ~7;
    bf16_t* vt = buf((size_t)B * C * Np);
    transpose(qkv.p + 2 * C, vt, N, C, 3 * C, Np, (long)N * 3 * C, (long)C * Np);
    GT at = act(H, W, C);
    float* lse1 = fbuf((size_t)B * heads * N);
    float* delta = fbuf((size_t)B * heads * N);
    attention(qkv.p, 3 * C, qkv.p + C, 3 * C, vt, Np, at.p, C, heads, N, N, 0, lse1);
    GemmArgs f_o1 = linear(at.p, M, C, a.o1, &a.o1b, h0, h1, C);
    // --- cross attention over the T text tokens
    const int Tp = (T + 7) & ~7;
    GT n2 = act(H, W, C);
    layernorm(h1, a.l2w, a.l2b, n2.p, M, C);
    GT q2 = act(H, W, C);
    GemmArgs f_q2 = linear(n2.p, M, C, a.q2, nullptr, nullptr, q2.p, C);
    GT at2 = act(H, W, C);
    float* lse2 = fbuf((size_t)B * heads * N);
    attention(q2.p, C, kx + a.x_off, XT, vxt + (size_t)a.x_off * Tp, Tp, at2.p, C, heads, N, T, (long)XT * Tp, lse2);
    GemmArgs f_o2 = linear(at2.p, M, C, a.o2, &a.o2b, h1, h2, C);
    // --- GEGLU feed-forward (pre-activation kept for the backward)
    GT n3 = act(H, W, C);
    layernorm(h2, a.l3w, a.l3b, n3.p, M, C);
    GT ffpre = act(H, W, 8 * C);
    GT ff = act(H, W, 4 * C);
    // ff.net.0: the GEMM epilogue gates (value x exact-erf GELU(gate)) into ff AND writes the bias-added pre-activations the backward
    // needs as a second output -- a separate gating pass read the 8C-wide tensor back (DFH_TRAIN_GEGLU_FUSED=0: the two-launch form, A/B)
    static const bool geglu_split = [] { const char* e = getenv("DFH_TRAIN_GEGLU_FUSED"); return e && e[0] == '0'; }();
    GemmArgs f_ff1;
    if (!geglu_split && (8 * C) % 128 == 0) {
      f_ff1 = base(M, 8 * C);
      f_ff1.p_src[0] = n3.p; f_ff1.p_c[0] = C; f_ff1.nplain = 1;
      f_ff1.W = w16(a.ff1); f_ff1.ldw = a.ff1.K; f_ff1.bias = v32(a.ff1b);
      f_ff1.act = ACT_GEGLU; f_ff1.out = ff.p; f_ff1.ld_out = 4 * C; f_ff1.pre_out = ffpre.p; f_ff1.ld_pre = 8 * C;
      gemm(f_ff1);
    } else {
      f_ff1 = linear(n3.p, M, C, a.ff1, &a.ff1b, nullptr, ffpre.p, 8 * C);
      TR_OP(dfh::geglu_fwd_launch(ffpre.p, ff.p, M, 8 * C, s));
    }
    GemmArgs f_ff2 = linear(ff.p, M, 4 * C, a.ff2, &a.ff2b, h2, h3, C);
    GemmArgs f_pout = linear(h3, M, C, a.pout, &a.poutb, x.p, out.p, C);
    gtemp.off = mark;
    const AttL* ap = &a;
    tape.push_back([=] {
      const AttL& a = *ap;
      // out = proj_out(h3) + x
      wgrad(f_pout, out.g, C, a.pout.off, &a.poutb);
      dgrad_linear(out.g, M, C, w16t(a.poutt), C, C, gh, false);                     // gh = d h3
      { const bool ax = acc(x); TR_OP(dfh::add_bf16_launch(x.g, out.g, (long)MC, ax, s)); }
      // h3 = ff2(ff) + h2
      wgrad(f_ff2, gh, C, a.ff2.off, &a.ff2b);
      dgrad_linear(gh, M, C, w16t(a.ff2t), C, 4 * C, ff.g, false);
      TR_OP(dfh::geglu_bwd_launch(ffpre.p, ff.g, ffpre.g, M, 8 * C, s));
      wgrad(f_ff1, ffpre.g, 8 * C, a.ff1.off, &a.ff1b);
      dgrad_linear(ffpre.g, M, 8 * C, w16t(a.ff1t), 8 * C, C, n3.g, false);
      join();                                                                         // the ff.net.2 weight gradient reads gh
      layernorm_bwd(h2, n3.g, a.l3w, a.l3b, gh, 1, M, C);                             // gh = d h2
      // h2 = o2(at2) + h1
      wgrad(f_o2, gh, C, a.o2.off, &a.o2b);
      dgrad_linear(gh, M, C, w16t(a.o2t), C, C, at2.g, false);
      attention_bwd(q2.p, C, kx + a.x_off, XT, vx + a.x_off, XT, at2.p, at2.g, C, lse2, delta, q2.g, C, dkx + a.x_off, XT,
                    dvx + a.x_off, XT, heads, N, T);
      wgrad(f_q2, q2.g, C, a.q2.off);
      dgrad_linear(q2.g, M, C, w16t(a.q2t), C, C, n2.g, false);
      join();
      layernorm_bwd(h1, n2.g, a.l2w, a.l2b, gh, 1, M, C);                             // gh = d h1
      // h1 = o1(at) + h0
      wgrad(f_o1, gh, C, a.o1.off, &a.o1b);
      dgrad_linear(gh, M, C, w16t(a.o1t), C, C, at.g, false);
      attention_bwd(qkv.p, 3 * C, qkv.p + C, 3 * C, qkv.p + 2 * C, 3 * C, at.p, at.g, C, lse1, delta, qkv.g, 3 * C, qkv.g + C, 3 * C,
                    qkv.g + 2 * C, 3 * C, heads, N, N);
      wgrad(f_qkv, qkv.g, 3 * C, a.qk.off);
      dgrad_linear(qkv.g, M, 3 * C, w16t(a.qkvt), 3 * C, C, n1.g, false);             // [dQ dK dV] . [Wq; Wk; Wv]
      join();
      layernorm_bwd(h0, n1.g, a.l1w, a.l1b, gh, 1, M, C);                             // gh = d h0
      // h0 = proj_in(gn)
      wgrad(f_pin, gh, C, a.pin.off, &a.pinb);
      dgrad_linear(gh, M, C, w16t(a.pint), C, C, gn.g, false);
      groupnorm_bwd(x, nullptr, gn.g, a.nw, a.nb, st, 0);
    });
    return out;
  }

  // ------------------------------------------------------------------ the walk
  void walk(const void* sample, int sample_bf16, const float* timestep, const void* ehs, int ehs_bf16, float* out) {
    dfh_unet& U = *u;
    const dfh_unet_config& cfg = U.cfg;
    const int S = cfg.sample_size, T = cfg.text_len, X = cfg.cross_attention_dim;
    const int* boc = cfg.block_out_channels;
    const int nb = cfg.num_blocks, temb = boc[0] * 4, TT = U.temb_total, XT = U.x_total;
    // ---- time embedding MLP, pre-activations kept
    bf16_t* tsin = buf((size_t)B * boc[0]);
    bf16_t *pre1 = buf((size_t)B * temb), *e1 = buf((size_t)B * temb), *pre2 = buf((size_t)B * temb), *e2 = buf((size_t)B * temb);
    bf16_t *dpre1 = buf((size_t)B * temb), *de1 = buf((size_t)B * temb), *dpre2 = buf((size_t)B * temb), *de2 = buf((size_t)B * temb);
    float* temb_all = fbuf((size_t)B * TT);
    dtemb_all = fbuf((size_t)B * TT); dtemb_bytes = (size_t)B * TT * 4;
    bf16_t* dtemb16 = buf((size_t)B * TT);
    TR_OP(dfh::timestep_embed_launch(timestep, tsin, B, boc[0], s));
    GemmArgs f_te1 = linear(tsin, B, boc[0], U.te1, &U.te1b, nullptr, pre1, temb);
    TR_OP(dfh::act_fwd_launch(pre1, e1, (long)B * temb, 1, s));
    GemmArgs f_te2 = linear(e1, B, temb, U.te2, &U.te2b, nullptr, pre2, temb);
    TR_OP(dfh::act_fwd_launch(pre2, e2, (long)B * temb, 1, s));
    GemmArgs f_tp = linear(e2, B, temb, U.tproj, &U.tprojb, nullptr, temb_all, TT, OUT_F32);
    tape.push_back([=] {      // runs LAST: every resnet has added its slice of d temb by then
      dfh_unet& U = *u;
      TR_OP(dfh::cast_f32_to_bf16_launch(dtemb_all, dtemb16, (long)B * TT, s));
      wgrad(f_tp, dtemb16, TT, U.tproj.off, &U.tprojb);
      dgrad_linear(dtemb16, B, TT, w16t(U.tprojt), TT, temb, de2, false);
      TR_OP(dfh::act_bwd_launch(pre2, nullptr, de2, nullptr, dpre2, (long)B * temb, 1, 1.0f, s));
      wgrad(f_te2, dpre2, temb, U.te2.off, &U.te2b);
      dgrad_linear(dpre2, B, temb, w16t(U.te2t), temb, temb, de1, false);
      TR_OP(dfh::act_bwd_launch(pre1, nullptr, de1, nullptr, dpre1, (long)B * temb, 1, 1.0f, s));
      wgrad(f_te1, dpre1, temb, U.te1.off, &U.te1b);
    });

    // ---- text K / V of every transformer layer (row-major kept for the backward, V^T for the forward kernel)
    bf16_t* ehs16 = buf((size_t)B * T * X);
    if (!dry && !rc) {
      if (ehs_bf16) (void)hipMemcpyAsync(ehs16, ehs, (size_t)B * T * X * 2, hipMemcpyDeviceToDevice, s);
      else rc = dfh::cast_f32_to_bf16_launch((const float*)ehs, ehs16, (long)B * T * X, s);
    }
    const int Tp = (T + 7) & ~7;
    bf16_t *kx = buf((size_t)B * T * XT), *vx = buf((size_t)B * T * XT), *vxt = buf((size_t)B * XT * Tp);
    bf16_t *dkx = buf((size_t)B * T * XT), *dvx = buf((size_t)B * T * XT);
    GemmArgs f_kx = linear(ehs16, B * T, X, U.kx_all, nullptr, nullptr, kx, XT);
    GemmArgs f_vx = linear(ehs16, B * T, X, U.vx_all, nullptr, nullptr, vx, XT);
    transpose(vx, vxt, T, XT, XT, Tp, (long)T * XT, (long)XT * Tp);
    tape.push_back([=] {      // after every cross-attention has written its slice of dK / dV
      wgrad(f_kx, dkx, XT, u->kx_all.off);
      wgrad(f_vx, dvx, XT, u->vx_all.off);
    });

    // ---- conv_in
    GT x; x.H = S; x.W = S; x.C = U.conv_in.cin; x.p = buf((size_t)B * S * S * x.C); x.g = buf((size_t)B * S * S * x.C);
    x.gid = (int)gstate.size(); gstate.push_back(0);
    TR_OP(dfh::nchw_to_nhwc_launch(sample, sample_bf16, x.p, B, cfg.in_channels, S * S, s));
    GT h = act(S, S, boc[0], true);
    {
      GemmArgs f = conv_desc(x.p, x.C, S, S, S, S, 1, 0, boc[0]);
      f.W = w16(U.conv_in.w); f.ldw = U.conv_in.w.K; f.bias = v32(U.conv_in.b); f.out = h.p;
      gemm(f);
      const GT h0 = h;
      tape.push_back([=] {
        dfh_unet& U = *u;
        const int M = B * S * S, C0 = U.conv_in.cout;
        wgrad(f, h0.g, C0, U.conv_in.w.off, &U.conv_in.b);
        if (dry || d_sample) {
          dgrad_conv(h0.g, C0, S, S, 0, U.conv_in.wt, x.C, x.g, false);
          if (d_sample) TR_OP(dfh::nhwc_to_nchw_f32_launch(x.g, d_sample, B, S * S, x.C, U.cfg.in_channels, 1.0f, 0, s));
        }
      });
    }
    std::vector<GT> skips{h};
    for (int i = 0; i < nb; ++i) {
      for (int j = 0; j < cfg.layers_per_block; ++j) {
        h = resnet(h, nullptr, U.down_res[i][j], temb_all);
        if (cfg.down_attn[i]) h = transformer(h, U.down_att[i][j], kx, vx, vxt, dkx, dvx, T);
        skips.push_back(h);
      }
      if (i != nb - 1) { h = conv(h, U.down_samp[i], 2, 0); skips.push_back(h); }
    }
    h = resnet(h, nullptr, U.mid_res[0], temb_all);
    h = transformer(h, U.mid_att, kx, vx, vxt, dkx, dvx, T);
    h = resnet(h, nullptr, U.mid_res[1], temb_all);
    for (int i = 0; i < nb; ++i) {
      for (int j = 0; j < (int)U.up_res[i].size(); ++j) {
        GT sk = skips.back(); skips.pop_back();
        h = resnet(h, &sk, U.up_res[i][j], temb_all);
        if (!U.up_att[i].empty()) h = transformer(h, U.up_att[i][j], kx, vx, vxt, dkx, dvx, T);
      }
      if (i != nb - 1) h = conv(h, U.up_samp[i], 1, 1);
    }
    // ---- conv_norm_out + SiLU + conv_out (fp32 NCHW noise prediction)
    {
      const int G = cfg.norm_num_groups, Co = cfg.out_channels, Cop = (Co + 7) & ~7, M = B * S * S;
      GT g = act(S, S, h.C, true);
      float* st = fbuf((size_t)B * G * 2);
      groupnorm(h, nullptr, U.cnw, U.cnb, cfg.norm_eps, 1, g.p, st);
      GemmArgs f = conv_desc(g.p, g.C, S, S, S, S, 1, 0, Co);
      f.W = w16(U.conv_out.w); f.ldw = U.conv_out.w.K; f.bias = v32(U.conv_out.b);
      f.out = out; f.out_mode = OUT_F32_T; f.ld_out = S * S; f.rows_per_b = S * S;
      gemm(f);
      bf16_t* dy = buf((size_t)M * Cop);
      const GT hl = h;
      tape.push_back([=] {     // runs FIRST
        dfh_unet& U = *u;
        TR_OP(dfh::nchw_to_nhwc_launch(d_out, 0, dy, B, Co, S * S, s));             // pads the channels to Cop with zeros
        wgrad(f, dy, Cop, U.conv_out.w.off, &U.conv_out.b);
        dgrad_conv(dy, Cop, S, S, 0, U.conv_out.wt, g.C, g.g, false);
        groupnorm_bwd(hl, nullptr, g.g, U.cnw, U.cnb, st, 1);
      });
    }
  }
};

dfh_unet::~dfh_unet() {
  delete tr;
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (ev_join) (void)hipEventDestroy(ev_join);
  if (side_stream) (void)hipStreamDestroy(side_stream);
  if (sq_partials) (void)hipFree(sq_partials);
  if (sq_scratch) (void)hipFree(sq_scratch);
}

// ------------------------------------------------------------------------------------------- build / plan
int dfh_unet::build_train() {
  if (train_built) return 0;
  std::map<std::string, int> idx;
  for (int i = 0; i < (int)params.size(); ++i) idx[params[i].name] = i;
  auto talloc = [&](int N, int K) { Mat m; m.N = N; m.K = K; m.off = a16t; a16t += ((size_t)N * K + 127) & ~(size_t)127; return m; };
  auto tmat = [&](const std::string& name, int N, int K, const Mat& dst, int t_row_off, int t_col_off, int geglu) {
    tpacks.push_back({idx.at(name), 0, dst.off, N, K, dst.K, t_row_off, t_col_off, geglu, 0});
  };
  auto tconv = [&](const std::string& name, int cout, int cin, const Mat& dst, int o_pad) {
    tpacks.push_back({idx.at(name), 1, dst.off, cout, cin, dst.K, 0, 0, 0, o_pad});
  };
  const int temb = cfg.block_out_channels[0] * 4;
  tprojt = talloc(temb, temb_total);
  te2t = talloc(temb, temb);
  tmat("time_embedding.linear_2.weight", temb, temb, te2t, 0, 0, 0);
  auto res = [&](ResL& r) {
    r.w1t = talloc(r.cin, 9 * r.cout);
    tconv(r.pre + ".conv1.weight", r.cout, r.cin, r.w1t, r.cout);
    r.w2t = talloc(r.cout, 9 * r.cout);
    tconv(r.pre + ".conv2.weight", r.cout, r.cout, r.w2t, r.cout);
    if (r.shortcut) {
      r.wst = talloc(r.cin, r.cout);
      tmat(r.pre + ".conv_shortcut.weight", r.cout, r.cin, r.wst, 0, 0, 0);
    }
    tmat(r.pre + ".time_emb_proj.weight", r.cout, temb, tprojt, 0, r.temb_off, 0);
  };
  auto att = [&](AttL& a) {
    const int C = a.C;
    const std::string tb = a.pre + ".transformer_blocks.0";
    a.pint = talloc(C, C); tmat(a.pre + ".proj_in.weight", C, C, a.pint, 0, 0, 0);
    a.qkvt = talloc(C, 3 * C);
    tmat(tb + ".attn1.to_q.weight", C, C, a.qkvt, 0, 0, 0);
    tmat(tb + ".attn1.to_k.weight", C, C, a.qkvt, 0, C, 0);
    tmat(tb + ".attn1.to_v.weight", C, C, a.qkvt, 0, 2 * C, 0);
    a.o1t = talloc(C, C); tmat(tb + ".attn1.to_out.0.weight", C, C, a.o1t, 0, 0, 0);
    a.q2t = talloc(C, C); tmat(tb + ".attn2.to_q.weight", C, C, a.q2t, 0, 0, 0);
    a.o2t = talloc(C, C); tmat(tb + ".attn2.to_out.0.weight", C, C, a.o2t, 0, 0, 0);
    a.ff1t = talloc(C, 8 * C); tmat(tb + ".ff.net.0.proj.weight", 8 * C, C, a.ff1t, 0, 0, 1);
    a.ff2t = talloc(4 * C, C); tmat(tb + ".ff.net.2.weight", C, 4 * C, a.ff2t, 0, 0, 0);
    a.poutt = talloc(C, C); tmat(a.pre + ".proj_out.weight", C, C, a.poutt, 0, 0, 0);
  };
  auto cv = [&](ConvL& c, int real_cin) {
    const int op = (c.cout + 7) & ~7;
    c.wt = talloc(c.cin, 9 * op);                  // rows beyond real_cin / columns beyond cout stay zero (zero-filled arena)
    tconv(c.pre + ".weight", c.cout, real_cin, c.wt, op);
  };
  cv(conv_in, cfg.in_channels);
  cv(conv_out, conv_out.cin);
  for (auto& v : down_res) for (auto& r : v) res(r);
  for (auto& v : up_res) for (auto& r : v) res(r);
  res(mid_res[0]); res(mid_res[1]);
  for (auto& v : down_att) for (auto& a : v) att(a);
  for (auto& v : up_att) for (auto& a : v) att(a);
  att(mid_att);
  for (int i = 0; i + 1 < cfg.num_blocks; ++i) { cv(down_samp[i], down_samp[i].cin); cv(up_samp[i], up_samp[i].cin); }
  train_built = true;
  return 0;
}

namespace {
size_t head_bytes_for(int B, size_t partial) {
  Bump hd; hd.alloc(256); hd.alloc((size_t)B * GN_MAX_CHUNKS * 64 * 2 * sizeof(float)); hd.alloc(partial); hd.alloc(partial);
  return (hd.off + 255) & ~(size_t)255;
}
}

size_t dfh_unet::plan_train(int B) {
  build_train();
  TrainRun r; r.u = this; r.B = B; r.dry = true;
  r.walk(nullptr, 0, nullptr, nullptr, 0, nullptr);
  for (auto it = r.tape.rbegin(); it != r.tape.rend(); ++it) (*it)();
  const size_t partial = (r.partial_need + 255) & ~(size_t)255;
  tplan_total = head_bytes_for(B, partial) + ((r.persist.peak + 255) & ~(size_t)255) + ((r.gtemp.peak + 255) & ~(size_t)255) + 256;
  tplan_batch = B;
  return tplan_total;
}

int dfh_unet::forward_train(const void* sample, int sample_bf16, const float* timestep, const void* ehs, int ehs_bf16, float* out,
                            int B, hipStream_t s) {
  // size the regions for this batch with a dry walk, then lay them out in the bound workspace
  TrainRun plan; plan.u = this; plan.B = B; plan.dry = true;
  plan.walk(nullptr, 0, nullptr, nullptr, 0, nullptr);
  for (int i = (int)plan.tape.size() - 1; i >= 0; --i) { plan.cur_entry = i; plan.tape[i](); }
  const size_t partial = (plan.partial_need + 255) & ~(size_t)255;
  const size_t persist_bytes = (plan.persist.peak + 255) & ~(size_t)255, gtemp_bytes = (plan.gtemp.peak + 255) & ~(size_t)255;
  const size_t head = head_bytes_for(B, partial);
  DFH_REQUIRE(head + persist_bytes + gtemp_bytes <= tws_bytes, "training workspace too small for this batch");
  delete tr;
  tr = new TrainRun();
  TrainRun& r = *tr;
  r.u = this; r.B = B; r.s = s; r.dry = false;
  Bump hd; hd.base = tws;
  r.zero = (bf16_t*)hd.alloc(256);
  r.gn_partial = (float*)hd.alloc((size_t)B * GN_MAX_CHUNKS * 64 * 2 * sizeof(float));
  r.partial = (float*)hd.alloc(partial); r.partial_cap = partial;
  r.partial2 = (float*)hd.alloc(partial);
  r.persist.base = tws + head;
  r.gtemp.base = tws + head + persist_bytes;
  (void)hipMemsetAsync(r.zero, 0, 256, s);
  r.walk(sample, sample_bf16, timestep, ehs, ehs_bf16, out);
  r.writes = std::move(plan.writes);          // same walk, same tape: entry i of the dry tape is entry i of this one
  DFH_REQUIRE(r.rc || r.tape.size() == plan.tape.size(), "dry and real tape differ");
  return r.rc;
}

int dfh_unet::backward(const float* d_out, float* d_sample, float* const* master_grads, int count, hipStream_t s, int overwrite) {
  DFH_REQUIRE(count == (int)params.size(), "parameter count mismatch");
  if (int rc = backward_begin(d_out, d_sample, a16 ? a16 : 1, s); rc < 0) return rc;
  size_t lo, hi;
  for (;;) {
    const int rc = backward_next(&lo, &hi, s);
    if (rc < 0) { tr->tape.clear(); tr->bucket = 0; return rc; }      // one backward per forward, also after an error
    if (rc == 0) break;
  }
  return backward_finish(master_grads, count, s, overwrite);
}

int dfh_unet::backward_begin(const float* d_out, float* d_sample, size_t bucket_floats, hipStream_t s) {
  DFH_REQUIRE(tr != nullptr && !tr->tape.empty(), "dfh_unet_backward needs a preceding dfh_unet_forward_train");
  DFH_REQUIRE(bucket_floats > 0, "bucket size must be positive");
  TrainRun& r = *tr;
  r.s = s; r.d_out = d_out; r.d_sample = d_sample;
  static const bool side_off = [] { const char* e = getenv("DFH_TRAIN_SIDE"); return e && e[0] == '0'; }();     // A/B
  if (!side_off && !side_stream) {
    if (hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) { dfh::set_error("side stream / events"); return -2; }
  }
  r.s2 = side_off ? nullptr : side_stream; r.forked = false;
  std::fill(r.gstate.begin(), r.gstate.end(), 0);
  (void)hipMemsetAsync(grad32, 0, a32 * sizeof(float), s);
  (void)hipMemsetAsync(r.dtemb_all, 0, r.dtemb_bytes, s);
  r.bucket = bucket_floats;
  const size_t nb = (a16 + bucket_floats - 1) / bucket_floats;
  r.bucket_last.assign(nb, -1);
  for (const auto& w : r.writes)
    for (size_t b = w[1] / bucket_floats; b <= (w[2] - 1) / bucket_floats && b < nb; ++b)
      if (r.bucket_last[b] < 0 || (int)w[0] < r.bucket_last[b]) r.bucket_last[b] = (int)w[0];
  r.ready.clear();
  r.next_entry = (int)r.tape.size() - 1;
  return (int)r.tape.size();
}

int dfh_unet::backward_next(size_t* lo, size_t* hi, hipStream_t s) {
  DFH_REQUIRE(tr != nullptr && tr->bucket > 0 && lo && hi, "dfh_unet_backward_begin must come first");
  TrainRun& r = *tr;
  r.s = s;
  while (r.ready.empty() && r.next_entry >= 0 && !r.rc) {
    const int i = r.next_entry--;
    r.tape[i]();
    r.join();                                   // side-stream weight gradients of this entry: done before its ranges are handed out
    // buckets whose last writer just ran; neighbours that finish together are handed out as one range
    for (size_t b = 0; b < r.bucket_last.size(); ++b) {
      if (r.bucket_last[b] != i) continue;
      const size_t first = b * r.bucket, end = std::min(a16, (b + 1) * r.bucket);
      if (!r.ready.empty() && r.ready.back().second == first) r.ready.back().second = end;
      else r.ready.push_back({first, end});
    }
  }
  if (r.rc) return r.rc;
  if (r.ready.empty()) return 0;
  *lo = r.ready.back().first; *hi = r.ready.back().second;
  r.ready.pop_back();
  return 1;
}

int dfh_unet::backward_finish(float* const* master_grads, int count, hipStream_t s, int overwrite) {
  DFH_REQUIRE(tr != nullptr && tr->bucket > 0, "dfh_unet_backward_begin must come first");
  DFH_REQUIRE(tr->next_entry < 0 && tr->ready.empty(), "dfh_unet_backward_next has not reached the end of the tape");
  DFH_REQUIRE(count == (int)params.size(), "parameter count mismatch");
  TrainRun& r = *tr;
  r.tape.clear(); r.bucket = 0;
  if (r.rc) return r.rc;
  tab_unpack.clear();
  for (const PackOp& op : packs) {
    float* g = master_grads ? master_grads[op.param] : nullptr;
    if (!g) continue;
    if (op.kind == PK_VEC) tab_unpack.add(g, TAB_UNPACK_VEC, (long)op.dst, op.N, 0, 0, op.geglu, overwrite, 0, 0, op.N);
    else if (op.kind == PK_MAT) tab_unpack.add(g, TAB_UNPACK_MAT, (long)op.dst, op.N, op.K, op.ldw, op.row_off, op.col_off, op.geglu, overwrite, (long)op.N * op.K);
    else tab_unpack.add(g, TAB_UNPACK_CONV, (long)op.dst, op.N, op.K, op.ldw, overwrite, op.col_off, 0, op.cin_pad, (long)op.N * op.K * 9);
  }
  if (!grad_sumsq_out) return tab_unpack.launch(grad32, grad16, s);
  // the partials / ticket scratch are sized and zeroed when the output is registered (dfh_unet_grad_sumsq); the partial buffer only
  // grows here if the un-pack table does (more gradients requested than at registration) -- behind a sync of the stream that may still
  // read the old one, never on the steady-state path
  if (tab_unpack.blocks > sq_cap) {
    if (sq_partials) { (void)hipStreamSynchronize(s); (void)hipFree(sq_partials); }
    sq_partials = nullptr; sq_cap = 0;
    const size_t want = std::max<size_t>(tab_unpack.blocks, 4096);
    if (hipMalloc((void**)&sq_partials, want * sizeof(float)) != hipSuccess) { dfh::set_error("hipMalloc of the norm partials failed"); return -1; }
    sq_cap = want;
  }
  if (!sq_scratch) {
    // stream-ordered zeroing: the ticket counter must be zero before table_sq_reduce_kernel on s touches it (a synchronous hipMemset on the
    // NULL stream is not ordered against a non-blocking stream)
    if (hipMalloc((void**)&sq_scratch, 257 * sizeof(float)) != hipSuccess || hipMemsetAsync(sq_scratch, 0, 257 * sizeof(float), s) != hipSuccess) {
      dfh::set_error("hipMalloc of the norm scratch failed"); return -1;
    }
  }
  if (int rc = tab_unpack.launch(grad32, grad16, s, nullptr, sq_partials)) return rc;
  return dfh::table_sq_reduce_launch(sq_partials, (long)tab_unpack.blocks, sq_scratch, grad_sumsq_out, s);
}

int dfh_unet::pack_train(const float* const* master, int count, hipStream_t s) {
  DFH_REQUIRE(count == (int)params.size(), "parameter count mismatch");
  DFH_REQUIRE(arena16t != nullptr, "training arenas not bound");
  tab_packt.clear();
  for (const TPackOp& op : tpacks) {
    void* src = (void*)master[op.param];
    DFH_REQUIRE(src != nullptr, "null master parameter: " + params[op.param].name);
    if (op.conv) tab_packt.add(src, TAB_PACKT_CONV, (long)op.dst, op.N, op.K, op.ldt, 0, op.t_col_off, 0, op.o_pad, (long)op.N * op.K * 9);
    else tab_packt.add(src, TAB_PACKT_MAT, (long)op.dst, op.N, op.K, op.ldt, op.t_row_off, op.t_col_off, op.geglu, 0, (long)op.N * op.K);
  }
  return tab_packt.launch(nullptr, arena16t, s);
}

// pack() + pack_train() of a training step in one pass over the masters: every weight that has exactly one plain and one transposed pack
// goes through a PACK2 op (read once, written to arena16 AND arena16t); vectors, the few weights packed more than once and the
// accumulating biases keep their own ops.  Same bytes in both arenas as the two separate calls (tests/test_gpu_train.py).
int dfh_unet::pack_all(const float* const* master, int count, hipStream_t s) {
  DFH_REQUIRE(count == (int)params.size(), "parameter count mismatch");
  DFH_REQUIRE(arena16 && arena32 && arena16t, "arenas not bound");
  std::vector<int> n_plain(params.size(), 0), n_tr(params.size(), 0), tr_at(params.size(), -1);
  for (const PackOp& op : packs) if (op.kind != PK_VEC) ++n_plain[op.param];
  for (size_t i = 0; i < tpacks.size(); ++i) { ++n_tr[tpacks[i].param]; tr_at[tpacks[i].param] = (int)i; }
  tab_pack2.clear(); tab_pack_acc.clear(); tab_packt.clear();
  for (const PackOp& op : packs) {
    void* src = (void*)master[op.param];
    DFH_REQUIRE(src != nullptr, "null master parameter: " + params[op.param].name);
    if (op.kind == PK_VEC) {
      (op.accumulate ? tab_pack_acc : tab_pack2).add(src, TAB_PACK_VEC, (long)op.dst, op.N, 0, 0, op.geglu, op.accumulate, 0, 0, op.N);
      continue;
    }
    const bool twin = n_plain[op.param] == 1 && n_tr[op.param] == 1;
    const TPackOp* t = twin ? &tpacks[tr_at[op.param]] : nullptr;
    if (t && t->N == op.N && t->K == op.K && (t->conv != 0) == (op.kind != PK_MAT) && t->geglu == op.geglu) {
      if (op.kind == PK_MAT)
        tab_pack2.add2(src, TAB_PACK2_MAT, (long)op.dst, op.N, op.K, op.ldw, op.row_off, op.col_off, op.geglu, 0, (long)t->dst, t->ldt, t->t_row_off, t->t_col_off, 0);
      else
        tab_pack2.add2(src, TAB_PACK2_CONV, (long)op.dst, op.N, op.K, op.ldw, 0, op.col_off, 0, op.cin_pad, (long)t->dst, t->ldt, 0, t->t_col_off, t->o_pad);
      n_tr[op.param] = -1;                      // its transposed pack is done
    } else if (op.kind == PK_MAT) {
      tab_pack2.add(src, TAB_PACK_MAT, (long)op.dst, op.N, op.K, op.ldw, op.row_off, op.col_off, op.geglu, 0, (long)op.N * op.K);
    } else {
      tab_pack2.add(src, TAB_PACK_CONV, (long)op.dst, op.N, op.K, op.ldw, 0, op.col_off, 0, op.cin_pad, (long)op.N * op.K * 9);
    }
  }
  for (const TPackOp& op : tpacks) {            // transposed packs without a twin
    if (n_tr[op.param] < 0) continue;
    void* src = (void*)master[op.param];
    if (op.conv) tab_packt.add(src, TAB_PACKT_CONV, (long)op.dst, op.N, op.K, op.ldt, 0, op.t_col_off, 0, op.o_pad, (long)op.N * op.K * 9);
    else tab_packt.add(src, TAB_PACKT_MAT, (long)op.dst, op.N, op.K, op.ldt, op.t_row_off, op.t_col_off, op.geglu, 0, (long)op.N * op.K);
  }
  if (int rc = tab_pack2.launch(arena32, arena16, s, arena16t)) return rc;
  if (int rc = tab_pack_acc.launch(arena32, arena16, s)) return rc;
  if (int rc = tab_packt.launch(nullptr, arena16t, s)) return rc;
  if (int rc = quantize_fp8(s)) return rc;
  fold_valid = false; fold_dirty = true;
  return 0;
}

// ------------------------------------------------------------------------------------------- C ABI
extern "C" {

size_t dfh_unet_arena16t_bytes(dfh_unet* u) { u->build_train(); return u->a16t * 2 + 256; }
size_t dfh_unet_grad16_bytes(const dfh_unet* u) { return u->a16 * 4 + 256; }
size_t dfh_unet_grad32_bytes(const dfh_unet* u) { return u->a32 * 4 + 256; }

size_t dfh_unet_train_workspace_bytes(dfh_unet* u, int batch) {
  if (batch <= 0) return 0;
  return u->plan_train(batch);
}

int dfh_unet_bind_train(dfh_unet* u, void* arena16t, void* grad16, void* grad32, void* workspace, size_t workspace_bytes,
                        int max_batch) {
  DFH_REQUIRE(u && arena16t && grad16 && grad32 && workspace, "null argument");
  DFH_REQUIRE(((uintptr_t)arena16t | (uintptr_t)grad16 | (uintptr_t)grad32 | (uintptr_t)workspace) % 256 == 0,
              "buffers must be 256-byte aligned");
  DFH_REQUIRE(u->arena16 && u->arena32, "dfh_unet_bind must come first (weights are shared with the inference path)");
  DFH_REQUIRE(workspace_bytes >= u->plan_train(max_batch), "workspace smaller than dfh_unet_train_workspace_bytes(max_batch)");
  u->arena16t = (bf16_t*)arena16t; u->grad16 = (float*)grad16; u->grad32 = (float*)grad32;
  u->tws = (char*)workspace; u->tws_bytes = workspace_bytes; u->train_max_batch = max_batch;
  return 0;
}

int dfh_unet_pack_train(dfh_unet* u, const float* const* master_params, int count, void* stream) {
  DFH_REQUIRE(u && master_params, "null argument");
  return u->pack_train(master_params, count, (hipStream_t)stream);
}

int dfh_unet_grad_sumsq(dfh_unet* u, float* out) {
  DFH_REQUIRE(u, "null argument");
  u->grad_sumsq_out = out;          // null un-registers: the un-pack then writes no norm (and keeps no pointer into caller memory)
  if (out && !u->sq_scratch) {      // ticket counter + block partials: allocated and zeroed HERE, outside any step
    if (hipMalloc((void**)&u->sq_scratch, 257 * sizeof(float)) != hipSuccess || hipMemset(u->sq_scratch, 0, 257 * sizeof(float)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
      dfh::set_error("hipMalloc of the norm scratch failed"); return -1;
    }
  }
  if (out && !u->sq_partials) {
    const size_t want = 1 << 16;    // blocks of the un-pack table of the SD-1.5 walk: ~21 k; the step re-checks
    if (hipMalloc((void**)&u->sq_partials, want * sizeof(float)) != hipSuccess) { dfh::set_error("hipMalloc of the norm partials failed"); return -1; }
    u->sq_cap = want;
  }
  return 0;
}

int dfh_unet_pack_all(dfh_unet* u, const float* const* master_params, int count, void* stream) {
  DFH_REQUIRE(u && master_params, "null argument");
  u->build_train();
  return u->pack_all(master_params, count, (hipStream_t)stream);
}

int dfh_unet_forward_train(dfh_unet* u, const void* sample, int sample_bf16, const float* timestep, const void* ehs, int ehs_bf16,
                           float* out, int batch, void* stream) {
  if (u) u->dup_tail = 0;          // the guidance-batch hint is an inference-walk hint: a training forward never consumes it and never leaves it behind
  DFH_REQUIRE(u && sample && timestep && ehs && out, "null argument");
  DFH_REQUIRE(u->tws != nullptr, "dfh_unet_bind_train not called");
  DFH_REQUIRE(batch > 0 && batch <= u->train_max_batch, "batch exceeds the bound max_batch");
  return u->forward_train(sample, sample_bf16, timestep, ehs, ehs_bf16, out, batch, (hipStream_t)stream);
}

int dfh_unet_backward(dfh_unet* u, const float* d_out, float* d_sample, float* const* master_grads, int count, int overwrite,
                      void* stream) {
  DFH_REQUIRE(u && d_out, "null argument");
  return u->backward(d_out, d_sample, master_grads, count, (hipStream_t)stream, overwrite ? 1 : 0);
}

int dfh_unet_backward_begin(dfh_unet* u, const float* d_out, float* d_sample, size_t bucket_floats, void* stream) {
  DFH_REQUIRE(u && d_out, "null argument");
  return u->backward_begin(d_out, d_sample, bucket_floats, (hipStream_t)stream);
}
int dfh_unet_backward_next(dfh_unet* u, size_t* lo, size_t* hi, void* stream) {
  DFH_REQUIRE(u, "null argument");
  return u->backward_next(lo, hi, (hipStream_t)stream);
}
int dfh_unet_backward_finish(dfh_unet* u, float* const* master_grads, int count, int overwrite, void* stream) {
  DFH_REQUIRE(u, "null argument");
  return u->backward_finish(master_grads, count, (hipStream_t)stream, overwrite ? 1 : 0);
}

}  // extern "C"
