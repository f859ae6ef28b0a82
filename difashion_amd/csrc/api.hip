// extern "C" surface of libdifashion_hip.so (declared in include/difashion_hip.h): error plumbing and
// the op-level entry points.  The U-Net context entry points live in unet.hip.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/difashion_hip.h"
#include "attention.h"
#include "dfh_common.h"
#include "elementwise.h"
#include "gemm.h"
#include "mlp_fused.h"
#include "norm.h"
#include "wgrad.h"
#include "bwd_elementwise.h"

namespace dfh {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return -2;
  }
  return 0;
}

static long g_census[CK_COUNT] = {0};
void census(int id) { if (id >= 0 && id < CK_COUNT) ++g_census[id]; }
static const char* const kCensusNames[CK_COUNT] = {"gemm_wide", "gemm_8wave", "gemm_lean", "gemm_other", "gemm_row", "splitk_reduce",
    "splitk_fused", "gstat_written", "gn_pre", "gn_stats", "gn_small", "gn_mid", "layernorm", "ln_folded", "attention_x32", "attention_16",
    "gemm_fp8", "text_cached", "conv_phase", "conv_wino", "gemm_rows_geglu", "attention_fp8", "mlp_fused", "gn_folded", "token_linear", "dup_prefix", "gemm_persist"};

struct ProfRec { hipEvent_t e0, e1; int cls; double flops, bytes; };
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_next = 0;
static bool g_prof = false;
bool prof_enabled() { return g_prof; }
static hipEvent_t pool_get() {
  if (g_pool_next == g_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); g_pool.push_back(e); }
  return g_pool[g_pool_next++];
}
void prof_open(int cls, double flops, double bytes, hipStream_t s) {
  ProfRec r; r.e0 = pool_get(); r.e1 = pool_get(); r.cls = cls; r.flops = flops; r.bytes = bytes;
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
}
void prof_close(hipStream_t s) { (void)hipEventRecord(g_recs.back().e1, s); }
// multiply-adds of the reference algorithm that a rewrite (phase planes, Winograd) did NOT execute, summed over a profiling window
static double g_prof_saved = 0.0;
void prof_note_saved(double flops) { if (g_prof) g_prof_saved += flops; }
}  // namespace dfh

static int fill_gemm(const dfh_gemm_desc* d, GemmArgs* g) {
  DFH_REQUIRE(d != nullptr, "null descriptor");
  std::memset(g, 0, sizeof(*g));
  if (d->conv) {
    DFH_REQUIRE(d->stride == 1 || d->stride == 2, "stride must be 1 or 2");
    DFH_REQUIRE(!(d->upsample && d->stride != 1), "upsample only with stride 1");
    g->conv_src = (const bf16_t*)d->conv_src; g->conv_c = d->conv_c; g->ntaps = 9;
    g->Hin = d->Hin; g->Win = d->Win; g->stride = d->stride; g->ups = d->upsample;
    g->Hout = d->upsample ? d->Hin * 2 : (d->stride == 2 ? d->Hin / 2 : d->Hin);
    g->Wout = d->upsample ? d->Win * 2 : (d->stride == 2 ? d->Win / 2 : d->Win);
    DFH_REQUIRE(d->M == d->batch * g->Hout * g->Wout, "M must equal batch * Hout * Wout");
  }
  if (d->a0) { g->p_src[g->nplain] = (const bf16_t*)d->a0; g->p_c[g->nplain] = d->a0_c; ++g->nplain; }
  if (d->a1) { DFH_REQUIRE(d->a0 != nullptr, "a1 without a0"); g->p_src[g->nplain] = (const bf16_t*)d->a1; g->p_c[g->nplain] = d->a1_c; ++g->nplain; }
  g->W = (const bf16_t*)d->W; g->ldw = d->ldw; g->zero = (const bf16_t*)d->zero_page;
  g->M = d->M; g->N = d->N;
  g->bias = d->bias; g->rowvec = d->rowvec; g->rv_ld = d->rv_ld; g->rv_off = d->rv_off;
  g->rows_per_b = d->rows_per_b > 0 ? d->rows_per_b : d->M;
  g->w_img_bs = (long)d->w_img_stride;
  g->resid = (const bf16_t*)d->resid; g->ld_res = d->ld_res;
  g->act = d->act; g->out = d->out; g->ld_out = d->ld_out; g->out_mode = d->out_mode;
  g->partial = d->partial;
  DFH_REQUIRE(g->W && g->out, "null W / out");
  return 0;
}

extern "C" {

int dfh_abi_version(void) { return DFH_ABI_VERSION; }

int dfh_prof_begin(void) {
  dfh::g_recs.clear(); dfh::g_pool_next = 0; dfh::g_prof = true; dfh::g_prof_saved = 0.0;
  return 0;
}
double dfh_prof_saved_flops(void) { return dfh::g_prof_saved; }
int dfh_prof_end(dfh_prof_class* out, int max_classes) {
  dfh::g_prof = false;
  DFH_REQUIRE(out != nullptr && max_classes >= dfh::PC_COUNT, "need room for every kernel class");
  static const char* names[dfh::PC_COUNT] = {"gemm_conv3x3", "gemm_linear", "attention", "groupnorm", "layernorm", "splitk_reduce", "other",
                                                  "gemm_wgrad", "attention_bwd", "norm_bwd", "optimizer", "gemm_linear_fp8"};
  for (int i = 0; i < dfh::PC_COUNT; ++i) {
    std::memset(&out[i], 0, sizeof(out[i]));
    std::strncpy(out[i].name, names[i], sizeof(out[i].name) - 1);
  }
  if (hipDeviceSynchronize() != hipSuccess) { dfh::set_error("hipDeviceSynchronize failed in dfh_prof_end"); return -2; }
  // DFH_PROF_DUMP=<file>: one line per launch (class, algorithmic flops, algorithmic bytes, ms) for per-shape tables
  const char* dump_path = std::getenv("DFH_PROF_DUMP");
  FILE* dump = dump_path ? std::fopen(dump_path, "w") : nullptr;
  for (const auto& r : dfh::g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    out[r.cls].launches += 1; out[r.cls].ms += ms; out[r.cls].flops += r.flops; out[r.cls].bytes += r.bytes;
    if (dump) std::fprintf(dump, "%s %.0f %.0f %.6f\n", names[r.cls], r.flops, r.bytes, ms);
  }
  if (dump) std::fclose(dump);
  dfh::g_recs.clear();
  return dfh::PC_COUNT;
}
void dfh_census_reset(void) { std::memset(dfh::g_census, 0, sizeof(dfh::g_census)); }
int dfh_census_count(void) { return dfh::CK_COUNT; }
const char* dfh_census_name(int i) { return (i >= 0 && i < dfh::CK_COUNT) ? dfh::kCensusNames[i] : ""; }
long dfh_census_get(int i) { return (i >= 0 && i < dfh::CK_COUNT) ? dfh::g_census[i] : -1; }
const char* dfh_last_error(void) { return dfh::g_err.c_str(); }
#define DFH_STR2(x) #x
#define DFH_STR(x) DFH_STR2(x)
const char* dfh_build_info(void) { return "libdifashion_hip gfx950 (CDNA4) bf16/fp8-MFMA abi=" DFH_STR(DFH_ABI_VERSION); }

size_t dfh_gemm_partial_floats(const dfh_gemm_desc* d) {
  GemmArgs g;
  if (fill_gemm(d, &g)) return 0;
  if (d->force_split > 1) return (size_t)d->force_split * g.M * g.N;
  return dfh::gemm_partial_floats(g);
}

int dfh_gemm(const dfh_gemm_desc* d, void* stream) {
  GemmArgs g;
  if (int rc = fill_gemm(d, &g)) return rc;
  const size_t need = d->force_split > 1 ? (size_t)d->force_split * g.M * g.N : dfh::gemm_partial_floats(g);
  DFH_REQUIRE(need == 0 || (d->partial && d->partial_floats >= need), "partial buffer too small for split-K");
  return dfh::gemm_launch(g, (hipStream_t)stream, d->force_tile, d->force_split, d->force_order);
}

int dfh_gemm_gstat(const dfh_gemm_desc* d, void* stream, int* written) {
  GemmArgs g;
  if (int rc = fill_gemm(d, &g)) return rc;
  DFH_REQUIRE(written != nullptr, "null argument");
  const size_t need = d->force_split > 1 ? (size_t)d->force_split * g.M * g.N : dfh::gemm_partial_floats(g);
  DFH_REQUIRE(need == 0 || (d->partial && d->partial_floats >= need), "partial buffer too small for split-K");
  g.gstat = d->gstat; g.gstat_cpg = d->gstat_cpg; g.gstat_hw = d->gstat_hw;
  int rows = 0;
  const int rc = dfh::gemm_launch(g, (hipStream_t)stream, d->force_tile, d->force_split, d->force_order, &rows);
  *written = rows;            // pixel rows per statistics chunk (256 or 128): d->gstat is [image][group][gstat_hw / rows][2]; 0 = not written
  return rc;
}

static int fill_wgrad(const dfh_gemm_desc* d, const void* dY, int ldy, float* dW, int ldw, int msplit, WgradArgs& w) {
  GemmArgs g;
  DFH_REQUIRE(d, "null argument");
  // only the A-operand description (K segments), M, N and the zero page of the descriptor are used
  dfh_gemm_desc tmp = *d;
  int dummy = 0;
  if (!tmp.W) tmp.W = &dummy;
  if (!tmp.out) tmp.out = &dummy;
  if (int rc = fill_gemm(&tmp, &g)) return rc;
  std::memset(&w, 0, sizeof(w));
  w.conv_src = g.conv_src; w.conv_c = g.conv_c; w.ntaps = g.ntaps;
  w.Hin = g.Hin; w.Win = g.Win; w.Hout = g.Hout; w.Wout = g.Wout; w.stride = g.stride; w.ups = g.ups;
  w.p_src[0] = g.p_src[0]; w.p_src[1] = g.p_src[1]; w.p_c[0] = g.p_c[0]; w.p_c[1] = g.p_c[1]; w.nplain = g.nplain;
  w.dY = (const bf16_t*)dY; w.ldy = ldy; w.zero = g.zero; w.M = g.M; w.N = g.N; w.dW = dW; w.ldw = ldw; w.msplit = msplit;
  w.partial = d->partial; w.partial_cap = d->partial_floats;
  return 0;
}

int dfh_gemm_wgrad(const dfh_gemm_desc* d, const void* dY, int ldy, float* dW, int ldw, int msplit, void* stream) {
  DFH_REQUIRE(d && dY && dW, "null argument");
  WgradArgs w;
  if (int rc = fill_wgrad(d, dY, ldy, dW, ldw, msplit, w)) return rc;
  return dfh::wgrad_launch(w, (hipStream_t)stream);
}

size_t dfh_gemm_wgrad_partial_floats(const dfh_gemm_desc* d, int msplit) {
  WgradArgs w;
  if (!d || fill_wgrad(d, nullptr, 8, nullptr, 8, msplit, w)) return 0;
  return dfh::wgrad_partial_floats(w);
}

int dfh_gemm_wgrad_plan(const dfh_gemm_desc* d, int msplit, int* tiles, int* whole_tiles, int* slices) {
  WgradArgs w;
  DFH_REQUIRE(d && tiles && whole_tiles && slices, "null argument");
  if (int rc = fill_wgrad(d, nullptr, 8, nullptr, 8, msplit, w)) return rc;
  dfh::wgrad_plan_only(w);
  *tiles = w.xblocks; *whole_tiles = w.whole; *slices = w.msplit;
  return 0;
}

int dfh_colsum(const void* Y, int ldy, int N, int groups, int rows_per_group, float* out, int ld_out, void* stream) {
  DFH_REQUIRE(Y && out, "null argument");
  return dfh::colsum_launch((const bf16_t*)Y, ldy, N, groups, rows_per_group, out, ld_out, (hipStream_t)stream);
}

int dfh_groupnorm(const void* src0, int c0, const void* src1, int c1, int batch, int hw, int groups, const float* gamma,
                  const float* beta, float eps, int silu, void* out, float* partial, void* stream) {
  GnArgs a; std::memset(&a, 0, sizeof(a));
  a.src0 = (const bf16_t*)src0; a.C0 = c0; a.src1 = (const bf16_t*)src1; a.C1 = src1 ? c1 : 0;
  a.B = batch; a.HW = hw; a.G = groups; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu;
  a.out = (bf16_t*)out; a.partial = partial;
  return dfh::groupnorm_launch(a, (hipStream_t)stream);
}

int dfh_groupnorm_pre(const void* src, int c, int batch, int hw, int groups, const float* gamma, const float* beta, float eps, int silu,
                      void* out, const float* gstat, int chunks, float* stats_out, void* stream) {
  GnArgs a; std::memset(&a, 0, sizeof(a));
  a.src0 = (const bf16_t*)src; a.C0 = c; a.B = batch; a.HW = hw; a.G = groups; a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.silu = silu; a.out = (bf16_t*)out; a.pre = gstat; a.pre_chunks = chunks; a.stats_out = stats_out;
  return dfh::groupnorm_launch(a, (hipStream_t)stream);
}

int dfh_groupnorm_stats(const void* src0, int c0, const void* src1, int c1, int batch, int hw, int groups, const float* gamma,
                        const float* beta, float eps, int silu, void* out, float* partial, float* stats_out, void* stream) {
  GnArgs a; std::memset(&a, 0, sizeof(a));
  a.src0 = (const bf16_t*)src0; a.C0 = c0; a.src1 = (const bf16_t*)src1; a.C1 = src1 ? c1 : 0;
  a.B = batch; a.HW = hw; a.G = groups; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu;
  a.out = (bf16_t*)out; a.partial = partial; a.stats_out = stats_out;
  return dfh::groupnorm_launch(a, (hipStream_t)stream);
}
int dfh_groupnorm_bwd(const void* src0, int c0, const void* src1, int c1, const void* dy, int batch, int hw, int groups,
                      const float* gamma, const float* beta, const float* stats, int silu, void* dx0, int acc0, void* dx1,
                      int acc1, float* dgamma, float* dbeta, float* partial, void* stream) {
  GnBwdArgs a; std::memset(&a, 0, sizeof(a));
  a.src0 = (const bf16_t*)src0; a.C0 = c0; a.src1 = (const bf16_t*)src1; a.C1 = src1 ? c1 : 0; a.dy = (const bf16_t*)dy;
  a.B = batch; a.HW = hw; a.G = groups; a.gamma = gamma; a.beta = beta; a.stats = stats; a.silu = silu;
  a.dx0 = (bf16_t*)dx0; a.dx1 = (bf16_t*)dx1; a.acc0 = acc0; a.acc1 = acc1; a.dgamma = dgamma; a.dbeta = dbeta; a.partial = partial;
  return dfh::groupnorm_bwd_launch(a, (hipStream_t)stream);
}
int dfh_attention_lse(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo, int batch,
                      int heads, int head_dim, int Nq, int Nk, float scale, float* lse, void* stream) {
  AttnArgs a; std::memset(&a, 0, sizeof(a));
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.Vt = (const bf16_t*)Vt; a.ldvt = ldvt;
  a.O = (bf16_t*)O; a.ldo = ldo; a.B = batch; a.H = heads; a.D = head_dim; a.Nq = Nq; a.Nk = Nk; a.scale = scale; a.lse = lse;
  return dfh::attention_launch(a, (hipStream_t)stream);
}
int dfh_attention_delta(const void* O, const void* dO, int ld, float* delta, int batch, int heads, int head_dim, int Nq, void* stream) {
  return dfh::attention_delta_launch((const bf16_t*)O, (const bf16_t*)dO, ld, delta, batch, heads, head_dim, Nq, (hipStream_t)stream);
}
int dfh_attention_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* dO, int ldo,
                      const float* lse, const float* delta, void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, int batch,
                      int heads, int head_dim, int Nq, int Nk, float scale, void* stream) {
  AttnBwdArgs a; std::memset(&a, 0, sizeof(a));
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.dO = (const bf16_t*)dO; a.ldo = ldo; a.lse = lse; a.delta = delta;
  a.dQ = (bf16_t*)dQ; a.lddq = lddq; a.dK = (bf16_t*)dK; a.lddk = lddk; a.dV = (bf16_t*)dV; a.lddv = lddv;
  a.B = batch; a.H = heads; a.D = head_dim; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
  return dfh::attention_bwd_launch(a, (hipStream_t)stream);
}
int dfh_layernorm_bwd(const void* x, const void* dy, const float* gamma, void* dx, int accumulate, float* dgamma, float* dbeta,
                      int M, int C, float eps, void* stream) {
  return dfh::layernorm_bwd_launch((const bf16_t*)x, (const bf16_t*)dy, gamma, (bf16_t*)dx, accumulate, dgamma, dbeta, M, C, eps,
                                   (hipStream_t)stream);
}
int dfh_pack_matrix_t(const float* w, void* out, int N, int K, int ldt, int t_row_off, int t_col_off, int geglu, void* stream) {
  return dfh::pack_matrix_t_launch(w, (bf16_t*)out, N, K, ldt, t_row_off, t_col_off, geglu, (hipStream_t)stream);
}
int dfh_pack_conv3x3_t(const float* w, void* out, int Cout, int Cin, int ldt, int t_col_off, int o_pad, void* stream) {
  return dfh::pack_conv3x3_t_launch(w, (bf16_t*)out, Cout, Cin, ldt, t_col_off, o_pad, (hipStream_t)stream);
}
int dfh_unpack_matrix(const float* g, float* grad, int N, int K, int ldw, int row_off, int col_off, int geglu, void* stream) {
  return dfh::unpack_matrix_launch(g, grad, N, K, ldw, row_off, col_off, geglu, (hipStream_t)stream);
}
int dfh_unpack_conv3x3(const float* g, float* grad, int Cout, int Cin, int ldw, int col_off, int cin_pad, void* stream) {
  return dfh::unpack_conv3x3_launch(g, grad, Cout, Cin, ldw, col_off, cin_pad, (hipStream_t)stream);
}
int dfh_unpack_vector(const float* g, float* grad, int N, int off, int geglu, void* stream) {
  return dfh::unpack_vector_launch(g, grad, N, off, geglu, (hipStream_t)stream);
}
int dfh_pool2x2_sum(const void* in, void* out, int batch, int H, int W, int C, void* stream) {
  return dfh::pool2x2_sum_launch((const bf16_t*)in, (bf16_t*)out, batch, H, W, C, (hipStream_t)stream);
}
int dfh_add_bf16(void* dst, const void* src, size_t n, int accumulate, void* stream) {
  return dfh::add_bf16_launch((bf16_t*)dst, (const bf16_t*)src, (long)n, accumulate, (hipStream_t)stream);
}
int dfh_geglu_fwd(const void* pre, void* y, size_t M, int N2, void* stream) {
  return dfh::geglu_fwd_launch((const bf16_t*)pre, (bf16_t*)y, (long)M, N2, (hipStream_t)stream);
}
int dfh_geglu_bwd(const void* pre, const void* dy, void* dpre, size_t M, int N2, void* stream) {
  return dfh::geglu_bwd_launch((const bf16_t*)pre, (const bf16_t*)dy, (bf16_t*)dpre, (long)M, N2, (hipStream_t)stream);
}
int dfh_act_fwd(const void* pre, void* y, size_t n, int kind, void* stream) {
  return dfh::act_fwd_launch((const bf16_t*)pre, (bf16_t*)y, (long)n, kind, (hipStream_t)stream);
}
int dfh_act_bwd(const void* ref_bf16, const float* ref_f32, const void* dy_bf16, const float* dy_f32, void* dpre, size_t n, int kind,
                float scale, void* stream) {
  return dfh::act_bwd_launch((const bf16_t*)ref_bf16, ref_f32, (const bf16_t*)dy_bf16, dy_f32, (bf16_t*)dpre, (long)n, kind, scale,
                             (hipStream_t)stream);
}
int dfh_nhwc_to_nchw_f32(const void* src, float* dst, int batch, int HW, int Cp, int C, float scale, int accumulate, void* stream) {
  return dfh::nhwc_to_nchw_f32_launch((const bf16_t*)src, dst, batch, HW, Cp, C, scale, accumulate, (hipStream_t)stream);
}
int dfh_transpose_bf16(const void* in, void* out, int batch, int R, int C, int ld_in, int ld_out, size_t in_bstride,
                       size_t out_bstride, void* stream) {
  return dfh::transpose_bf16_launch((const bf16_t*)in, (bf16_t*)out, batch, R, C, ld_in, ld_out, (long)in_bstride, (long)out_bstride,
                                    (hipStream_t)stream);
}
int dfh_mse_bwd(const float* pred, const float* target, const float* w, float* dpred, int rows, int L, float loss_scale,
                const float* scale_dev, void* stream) {
  return dfh::mse_bwd_launch(pred, target, w, dpred, rows, L, loss_scale, scale_dev, (hipStream_t)stream);
}
int dfh_assemble_bwd(const float* dx, const uint8_t* mutual_real, float* dmutual, int rows, int CL, float eta, void* stream) {
  return dfh::assemble_bwd_launch(dx, mutual_real, dmutual, rows, CL, eta, (hipStream_t)stream);
}
int dfh_sumsq(const float* g, size_t n, float* out, void* stream) { return dfh::sumsq_launch(g, (long)n, out, (hipStream_t)stream); }
int dfh_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, const float* sumsq, float max_norm, void* stream) {
  return dfh::adamw_launch(p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, step, sumsq, max_norm, (hipStream_t)stream);
}
int dfh_adamw_ema(float* p, const float* g, float* m, float* v, float* shadow, size_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, const float* sumsq, float max_norm, float ema_decay, void* stream) {
  DFH_REQUIRE(shadow != nullptr, "null shadow");
  return dfh::adamw_launch(p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, step, sumsq, max_norm, (hipStream_t)stream, shadow,
                           ema_decay);
}
int dfh_ema(float* shadow, const float* p, size_t n, float decay, void* stream) {
  return dfh::ema_launch(shadow, p, (long)n, decay, (hipStream_t)stream);
}

int dfh_wire_pack(const float* g, void* wire, size_t n, size_t n_pad, void* stream) {
  return dfh::wire_pack_launch(g, (bf16_t*)wire, (long)n, (long)n_pad, (hipStream_t)stream);
}
int dfh_wire_shard_mean(const void* recv, void* shard, int world, size_t per, void* stream) {
  return dfh::wire_shard_mean_launch((const bf16_t*)recv, (bf16_t*)shard, world, (long)per, (hipStream_t)stream);
}
int dfh_wire_unpack(const void* wire, float* g, size_t n, void* stream) {
  return dfh::wire_unpack_launch((const bf16_t*)wire, g, (long)n, (hipStream_t)stream);
}

int dfh_layernorm(const void* x, const float* gamma, const float* beta, void* y, int M, int C, float eps, void* stream) {
  return dfh::layernorm_launch((const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps, (hipStream_t)stream);
}

// LayerNorm folded into the GEMMs around it (gemm.h, lnfold.hip)
int dfh_groupnorm_fold(const void* x, int B, int HW, int C, int G, const float* gamma, const float* beta, float eps, const float* pre,
                       int pre_chunks, float* partial, const void* W, int ldw, int N, const float* bias, void* Wimg, float* rv, void* stream) {
  GnFoldArgs f; std::memset(&f, 0, sizeof(f));
  f.x = (const bf16_t*)x; f.B = B; f.HW = HW; f.C = C; f.G = G; f.eps = eps; f.gamma = gamma; f.beta = beta; f.pre = pre; f.pre_chunks = pre_chunks;
  f.partial = partial; f.W = (const bf16_t*)W; f.ldw = ldw; f.N = N; f.bias = bias; f.Wimg = (bf16_t*)Wimg; f.rv = rv;
  return dfh::groupnorm_fold_launch(f, (hipStream_t)stream);
}

int dfh_ln_fold(const void* W, int ldw, const float* gamma, const float* beta, const float* bias, void* WF, float* s, float* b, int N, int K,
                void* stream) {
  return dfh::ln_fold_launch((const bf16_t*)W, ldw, gamma, beta, bias, (bf16_t*)WF, s, b, N, K, (hipStream_t)stream);
}

int dfh_gemm_ln(const dfh_gemm_desc* d, float* rowstat, int* rowstat_bn, const float* ln_stat, int ln_parts, int ln_cnt, float ln_eps,
                const float* ln_s, void* stream) {
  GemmArgs g;
  if (int rc = fill_gemm(d, &g)) return rc;
  const size_t need = d->force_split > 1 ? (size_t)d->force_split * g.M * g.N : dfh::gemm_partial_floats(g);
  DFH_REQUIRE(need == 0 || (d->partial && d->partial_floats >= need), "partial buffer too small for split-K");
  g.rowstat = rowstat;
  g.ln_stat = ln_stat; g.ln_parts = ln_parts; g.ln_cnt = ln_cnt; g.ln_eps = ln_eps; g.ln_s = ln_s;
  if (ln_stat) DFH_REQUIRE(dfh::gemm_ln_consumer_ok(g), "this launch shape cannot consume folded-LayerNorm statistics (split-K / unstaged GEGLU)");
  int bn = 0;
  const int rc = dfh::gemm_launch(g, (hipStream_t)stream, d->force_tile, d->force_split, d->force_order, nullptr, &bn);
  if (rowstat_bn) *rowstat_bn = bn;
  return rc;
}

int dfh_gemm_out2(const dfh_gemm_desc* d, void* out2, int ld_out2, int n_split, void* stream) {
  GemmArgs g;
  if (int rc = fill_gemm(d, &g)) return rc;
  DFH_REQUIRE(out2 != nullptr, "null argument");
  g.out2 = out2; g.ld_out2 = ld_out2; g.n_split = n_split;
  DFH_REQUIRE(dfh::gemm_out2_ok(g), "this launch cannot carry a second destination (split-K, or n_split is no multiple of its column tile)");
  // tile ids 10 / 24 pin the one-tile-per-workgroup / the persistent 128 x 160 kernel (tests, same-box A/B); anything else = heuristics
  const int ft = (d->force_tile == 10 || d->force_tile == 24) ? d->force_tile : 0;
  return dfh::gemm_launch(g, (hipStream_t)stream, ft, 0, d->force_order);
}

// nearest-2x upsample + 3x3 conv as four phase planes over the source image (gemm.h GemmArgs::phase2x)
int dfh_ups_phase_fold(const void* W, int ldw, void* WP, int N, int C, void* stream) {
  return dfh::ups_phase_fold_launch((const bf16_t*)W, ldw, (bf16_t*)WP, N, C, (hipStream_t)stream);
}
int dfh_conv_up2x(const void* src, int batch, int H, int W, int C, const void* WP, int N, const float* bias, void* out,
                  const void* zero_page, void* stream) {
  DFH_REQUIRE(src && WP && out && zero_page, "null argument");
  DFH_REQUIRE(batch > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && N > 0 && N % 8 == 0, "bad shape (C and N multiples of 8)");
  GemmArgs g; std::memset(&g, 0, sizeof(g));
  g.M = batch * H * W; g.N = N; g.rows_per_b = H * W; g.out_mode = OUT_BF16; g.ld_out = N;
  g.conv_src = (const bf16_t*)src; g.conv_c = C; g.ntaps = 4; g.phase2x = 1; g.nbatch = 4; g.w_bs = (long)N * 4 * C;
  g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1;
  g.W = (const bf16_t*)WP; g.ldw = 4 * C; g.bias = bias; g.out = out; g.zero = (const bf16_t*)zero_page;
  return dfh::gemm_launch(g, (hipStream_t)stream, 0, 0, -1);
}
// Winograd F(2x2, 3x3) conv (winograd.hip): weight transform, and the three-launch conv over caller-provided scratch
int dfh_wino_weights(const void* W, int ldw, void* U, int N, int C, int blocked, void* stream) {
  return dfh::wino_weight_launch((const bf16_t*)W, ldw, (bf16_t*)U, N, C, blocked, (hipStream_t)stream);
}
int dfh_wino_blocked(int N, int C) { return dfh::wino_blocked(N, C) ? 1 : 0; }
int dfh_wino_input(const void* src, void* V, int batch, int H, int W, int C, void* stream) {
  return dfh::wino_input_launch((const bf16_t*)src, (bf16_t*)V, batch, H, W, C, (hipStream_t)stream);
}
int dfh_gn_wino_input_ok(int c0, int c1, int groups, int H, int W) { return dfh::gn_wino_ok(c0, c1, groups, H, W) ? 1 : 0; }
int dfh_gn_wino_input(const void* src0, int c0, const void* src1, int c1, const float* gamma, const float* beta, float eps, int groups,
                      void* V, int batch, int H, int W, void* stream) {
  return dfh::gn_wino_input_launch((const bf16_t*)src0, c0, (const bf16_t*)src1, c1, gamma, beta, eps, groups, (bf16_t*)V, batch, H, W,
                                   (hipStream_t)stream);
}
int dfh_gn_wino_input_chain(const void* Mprev, const float* pbias, const float* prowvec, int prv_ld, int prv_off, int C, const float* gamma,
                            const float* beta, float eps, int groups, void* V, int batch, int H, int W, void* stream) {
  DFH_REQUIRE(Mprev != nullptr, "null argument");
  return dfh::gn_wino_input_launch((const bf16_t*)Mprev, C, nullptr, 0, gamma, beta, eps, groups, (bf16_t*)V, batch, H, W, (hipStream_t)stream,
                                   (const bf16_t*)Mprev, pbias, prowvec, prv_ld, prv_off);
}
size_t dfh_conv3x3_wino_scratch_bytes(int batch, int H, int W, int C, int N) {
  if (batch <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return 0;
  const size_t mt = (size_t)batch * (H / 2) * (W / 2);
  return ((16 * mt * C * 2 + 255) & ~(size_t)255) + ((16 * mt * N * 2 + 255) & ~(size_t)255);
}
int dfh_conv3x3_wino(const void* src, int batch, int H, int W, int C, const void* U, int u_blocked, int N, const float* bias, const float* rowvec,
                     int rv_ld, int rv_off, const void* resid, void* out, void* scratch, size_t scratch_bytes, const void* zero_page,
                     void* stream) {
  DFH_REQUIRE(src && U && bias && out && scratch && zero_page, "null argument");
  DFH_REQUIRE(batch > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0 && C > 0 && C % 8 == 0 && N > 0 && N % 8 == 0,
              "even image sides, channel counts multiples of 8");
  DFH_REQUIRE(scratch_bytes >= dfh_conv3x3_wino_scratch_bytes(batch, H, W, C, N) && (uintptr_t)scratch % 256 == 0, "scratch too small or misaligned");
  const long mt = (long)batch * (H / 2) * (W / 2);
  bf16_t* V = (bf16_t*)scratch;
  bf16_t* Mb = (bf16_t*)((char*)scratch + ((16 * (size_t)mt * C * 2 + 255) & ~(size_t)255));
  hipStream_t s = (hipStream_t)stream;
  if (int rc = dfh::wino_input_launch((const bf16_t*)src, V, batch, H, W, C, s)) return rc;
  GemmArgs g; std::memset(&g, 0, sizeof(g));
  g.M = (int)mt; g.N = N; g.rows_per_b = (int)mt; g.out_mode = OUT_BF16; g.ld_out = N;
  g.p_src[0] = V; g.p_c[0] = C; g.nplain = 1; g.W = (const bf16_t*)U; g.ldw = C;
  g.nbatch = 16; g.a_bs = mt * C; g.w_bs = (long)N * C; g.o_bs = mt * N; g.w_blocked = u_blocked;
  g.out = Mb; g.zero = (const bf16_t*)zero_page;
  g.prof_flops = 2.0 * batch * H * W * (double)N * 9.0 * C;
  if (int rc = dfh::gemm_launch(g, s, dfh::wino_gemm_tile(g), 0, -1)) return rc;
  return dfh::wino_output_launch(Mb, (bf16_t*)out, bias, rowvec, rv_ld, rv_off, (const bf16_t*)resid, batch, H, W, N, s);
}
// dfh_gemm over nbatch independent planes in one launch: plane z reads a0 + z * a_bs, W + z * w_bs and writes out + z * o_bs (elements)
int dfh_gemm_batched(const dfh_gemm_desc* d, int nbatch, long a_bs, long w_bs, long o_bs, void* stream) {
  GemmArgs g;
  if (int rc = fill_gemm(d, &g)) return rc;
  DFH_REQUIRE(nbatch >= 1, "nbatch must be positive");
  g.nbatch = nbatch; g.a_bs = a_bs; g.w_bs = w_bs; g.o_bs = o_bs;
  return dfh::gemm_launch(g, (hipStream_t)stream, d->force_tile, 0, d->force_order);
}

int dfh_gemm_fp8(const dfh_gemm_fp8_desc* d, void* stream) {
  DFH_REQUIRE(d != nullptr, "null descriptor");
  Fp8GemmArgs g; std::memset(&g, 0, sizeof(g));
  g.A = (const uint8_t*)d->A; g.lda = d->lda; g.sA = d->sA; g.sa_div = d->sa_div; g.sa_mul = d->sa_mul; g.sx = (const uint8_t*)d->sx;
  g.W = (const uint8_t*)d->W; g.sW = d->sW; g.M = d->M; g.N = d->N; g.K = d->K; g.bias = d->bias;
  g.resid = (const bf16_t*)d->resid; g.ld_res = d->ld_res; g.act = d->act; g.out = d->out; g.ld_out = d->ld_out; g.out_mode = d->out_mode;
  g.out_sx = (uint8_t*)d->out_sx; g.rows_per_b = d->rows_per_b; g.amax = d->amax; g.zero = (const uint8_t*)d->zero_page;
  return dfh::gemm_fp8_launch(g, (hipStream_t)stream);
}
size_t dfh_mlp_fused_image_bytes(void) { return dfh::mlp_fused_image_bytes(); }
int dfh_mlp_fused_pack(const void* w1, const float* s1, const float* b1, const void* w2p, void* img, int form, void* stream) {
#ifdef DFH_PROBES
  if (form == 1) return dfh::mlp_pack_launch((const bf16_t*)w1, s1, b1, (const bf16_t*)w2p, img, (hipStream_t)stream);
#endif
  DFH_REQUIRE(form == 2, "fused MLP: form 2 (form 1 is a probe kernel: scripts/probes)");
  return dfh::mlp2_pack_launch((const bf16_t*)w1, s1, b1, (const bf16_t*)w2p, img, (hipStream_t)stream);
}
int dfh_mlp_fused(const void* x, const void* resid, const void* img, const float* ln_stat, int ln_parts, int ln_cnt, float ln_eps,
                  const float* bias, void* out, int M, int form, float* gstat, int gstat_cpg, int gstat_hw, void* stream) {
#ifndef DFH_PROBES
  DFH_REQUIRE(form == 2, "fused MLP: form 2 (form 1 is a probe kernel: scripts/probes)");
#endif
  DFH_REQUIRE(form == 1 || form == 2, "fused MLP: form 1 or 2");
  DFH_REQUIRE(gstat == nullptr || form == 2, "fused MLP: output statistics are written by form 2 only");
  MlpArgs a; std::memset(&a, 0, sizeof(a));
  a.gstat = gstat; a.gstat_cpg = gstat_cpg; a.gstat_hw = gstat_hw;
  a.x = (const bf16_t*)x; a.resid = (const bf16_t*)resid; a.img = (const unsigned char*)img; a.ln_stat = ln_stat; a.ln_parts = ln_parts;
  a.ln_cnt = ln_cnt; a.ln_eps = ln_eps; a.bias = bias; a.out = (bf16_t*)out; a.M = M;
#ifdef DFH_PROBES
  if (form == 1) return dfh::mlp_fused_launch(a, (hipStream_t)stream);
#endif
  return dfh::mlp2_fused_launch(a, (hipStream_t)stream);
}
int dfh_groupnorm_fp8(const void* src, int batch, int HW, int C, int groups, float eps, float q_mul, void* q, float* partial, void* stream) {
  GnArgs a; std::memset(&a, 0, sizeof(a));
  a.src0 = (const bf16_t*)src; a.C0 = C; a.B = batch; a.HW = HW; a.G = groups; a.eps = eps; a.out8 = (uint8_t*)q; a.q_mul = q_mul;
  a.partial = partial;
  return dfh::groupnorm_launch(a, (hipStream_t)stream);
}
int dfh_attention_fp8out(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O8, int ldo, const float* v_amax,
                         int batch, int heads, int head_dim, int Nq, int Nk, float scale, void* stream) {
  AttnArgs a; std::memset(&a, 0, sizeof(a));
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.Vt = (const bf16_t*)Vt; a.ldvt = ldvt;
  a.O8 = (uint8_t*)O8; a.ldo = ldo; a.o_amax = v_amax; a.B = batch; a.H = heads; a.D = head_dim; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
  DFH_REQUIRE(O8 && v_amax, "fp8 attention output needs O8 and the per-batch maxima of V");
  return dfh::attention_launch(a, (hipStream_t)stream);
}
int dfh_attention_fp8(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo, const float* rq,
                      const float* rk, const float* rv, const float* hs, int batch, int heads, int head_dim, int Nq, int Nk, float scale,
                      void* stream) {
  AttnArgs a; std::memset(&a, 0, sizeof(a));
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.Vt = (const bf16_t*)Vt; a.ldvt = ldvt;
  a.O = (bf16_t*)O; a.ldo = ldo; a.B = batch; a.H = heads; a.D = head_dim; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
  a.f8_rq = rq; a.f8_rk = rk; a.f8_rv = rv; a.f8_hs = hs;
  DFH_REQUIRE(rq && rk && rv && hs, "fp8 attention needs its operand factors");
  DFH_REQUIRE((head_dim == 40 || head_dim == 80 || head_dim == 160) && Nk >= 64 && Nk % 64 == 0, "fp8 attention: head dim 40 / 80 / 160, whole 64-key tiles");
  return dfh::attention_launch(a, (hipStream_t)stream);
}
int dfh_attn_scales(const void* wf, const float* bf, int C, int heads, float* rq, float* rk, float* rv, float* hs, void* stream) {
  return dfh::attn_scales_launch((const bf16_t*)wf, bf, C, heads, rq, rk, rv, hs, (hipStream_t)stream);
}
int dfh_amax_slabs(const void* x, long bstride, int ld, int cols, const int* row0, const int* nrows, float* out, int nslab, int batch,
                   void* stream) {
  return dfh::amax_slabs_launch((const bf16_t*)x, bstride, ld, cols, row0, nrows, out, nslab, batch, (hipStream_t)stream);
}
int dfh_quantize_rows_fp8(const void* x, int ldx, void* q, float* scale, int R, int K, void* stream) {
  return dfh::quant_rows_fp8_launch((const bf16_t*)x, ldx, (uint8_t*)q, scale, R, K, (hipStream_t)stream);
}
int dfh_layernorm_fp8(const void* x, const float* gamma, const float* beta, void* q, float* scale, int M, int C, float eps, void* stream) {
  return dfh::layernorm_fp8_launch((const bf16_t*)x, gamma, beta, (uint8_t*)q, scale, M, C, eps, (hipStream_t)stream);
}

int dfh_attention(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo, int batch,
                  int heads, int head_dim, int Nq, int Nk, float scale, void* stream) {
  AttnArgs a; std::memset(&a, 0, sizeof(a));
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.Vt = (const bf16_t*)Vt; a.ldvt = ldvt;
  a.O = (bf16_t*)O; a.ldo = ldo; a.B = batch; a.H = heads; a.D = head_dim; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
  return dfh::attention_launch(a, (hipStream_t)stream);
}

int dfh_timestep_embedding(const float* t, void* out, int batch, int dim, void* stream) {
  return dfh::timestep_embed_launch(t, (bf16_t*)out, batch, dim, (hipStream_t)stream);
}
int dfh_nchw_to_nhwc_bf16(const void* x, int x_bf16, void* out, int batch, int C, int HW, void* stream) {
  return dfh::nchw_to_nhwc_launch(x, x_bf16, (bf16_t*)out, batch, C, HW, (hipStream_t)stream);
}
int dfh_cast_f32_to_bf16(const float* x, void* y, size_t n, void* stream) {
  return dfh::cast_f32_to_bf16_launch(x, (bf16_t*)y, (long)n, (hipStream_t)stream);
}
int dfh_pack_conv3x3(const float* w, void* out, int Cout, int Cin, int ldw, int col_off, void* stream) {
  return dfh::pack_conv3x3_launch(w, (bf16_t*)out, Cout, Cin, ldw, col_off, (hipStream_t)stream);
}
int dfh_pack_matrix(const float* w, void* out, int N, int K, int ldw, int row_off, int col_off, int geglu, void* stream) {
  return dfh::pack_matrix_launch(w, (bf16_t*)out, N, K, ldw, row_off, col_off, geglu, (hipStream_t)stream);
}
int dfh_pack_vector(const float* v, float* out, int N, int off, int geglu, int accumulate, void* stream) {
  return dfh::pack_vector_launch(v, out, N, off, geglu, accumulate, (hipStream_t)stream);
}

int dfh_mutual_reduce(const float* gen, const float* given, const int32_t* table, const float* wtab, void* out_bf16,
                      float* out_f32, int rows, int olen, int L, void* stream) {
  DFH_REQUIRE(gen && table && wtab && out_bf16, "null argument");
  return dfh::mutual_reduce_launch(gen, given, table, wtab, (bf16_t*)out_bf16, out_f32, rows, olen, L, (hipStream_t)stream);
}
int dfh_assemble_input(const float* latents, const float* mutual, const float* hist, const float* null_latent,
                       const uint8_t* mutual_real, const uint8_t* hist_real, float* x, int R, int F, int CL,
                       float one_minus_eta, float eta, int per_row_flags, void* stream) {
  DFH_REQUIRE(latents && mutual && hist && null_latent && mutual_real && hist_real && x, "null argument");
  return dfh::assemble_input_launch(latents, mutual, hist, null_latent, mutual_real, hist_real, x, R, F, CL, one_minus_eta,
                                    eta, per_row_flags, (hipStream_t)stream);
}
int dfh_cfg_step(const float* eps_all, float* latents, float* eps_out, const float* noise, size_t n, int mode,
                 float cate_scale, float hist_scale, float mutual_scale, const dfh_step_coef* k, void* stream) {
  DFH_REQUIRE(eps_all && k, "null argument");
  DFH_REQUIRE(k->kind == STEP_NONE || latents != nullptr, "latents required for a scheduler step");
  StepCoef c; c.kind = k->kind; c.vpred = k->vpred; c.sqrt_a_t = k->sqrt_a_t; c.sqrt_b_t = k->sqrt_b_t;
  c.sqrt_a_prev = k->sqrt_a_prev; c.dir_coef = k->dir_coef; c.std_dev = k->std_dev;
  return dfh::cfg_step_launch(eps_all, latents, eps_out, noise, (long)n, mode, cate_scale, hist_scale, mutual_scale, c,
                              (hipStream_t)stream);
}
int dfh_noise_mix(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp,
                  float* noisy, float* velocity, int rows, int L, void* stream) {
  return dfh::noise_mix_launch(x0, noise, (const long*)t, sqrt_acp, sqrt_1m_acp, noisy, velocity, rows, L, (hipStream_t)stream);
}
int dfh_mse_rows(const float* pred, const float* target, float* out, int rows, int L, void* stream) {
  return dfh::mse_rows_launch(pred, target, out, rows, L, (hipStream_t)stream);
}

}  // extern "C"
