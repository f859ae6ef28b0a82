// Host-side runtime of the conditional U-Net: parameter table, weight-packing plan, workspace
// planning and the forward walk that strings the gfx950 kernels together.
//
// Replaces (arithmetic): diffusers 0.18.2 UNet2DConditionModel.forward as used by the reference at
// DiFashion/models/difashion.py:249-253 (training) and :518-523 (sampling), topology per SURVEY.md
// Appendix A.2/A.3; parameter names per Appendix A.4 so reference checkpoints drop in
// (train.py:524-547).
//
// MI355X-first choices:
//   * activations stay NHWC bf16 from conv_in to conv_out: conv3x3, 1x1 conv, and every transformer
//     linear are the same implicit GEMM with no permutes between the conv and attention halves;
//   * fusions expressed as GEMM K-segments / epilogues: skip concat (two sources), nearest-2x
//     upsample, stride-2 downsample, resnet 1x1 shortcut accumulated into conv2, time-embedding
//     broadcast add, bias, residual add, GEGLU gate, V^T for attention, NCHW fp32 noise prediction;
//   * the 22 per-resnet time_emb_proj linears are one batched GEMM at the top of the step;
//   * no allocation after bind(): a dry run of this same walk sizes the workspace.
#include "unet_model.h"

// ------------------------------------------------------------------------------------------- C ABI
extern "C" {

int dfh_unet_create(const dfh_unet_config* cfg, dfh_unet** out) {
  DFH_REQUIRE(cfg && out, "null argument");
  DFH_REQUIRE(cfg->num_blocks >= 2 && cfg->num_blocks <= DFH_MAX_BLOCKS, "num_blocks out of range");
  DFH_REQUIRE(cfg->in_channels > 0 && cfg->in_channels <= 64, "in_channels out of range");
  DFH_REQUIRE(cfg->out_channels % 4 == 0, "out_channels must be a multiple of 4");
  DFH_REQUIRE(cfg->cross_attention_dim % 8 == 0, "cross_attention_dim must be a multiple of 8");
  DFH_REQUIRE(cfg->sample_size % (1 << (cfg->num_blocks - 1)) == 0, "sample_size not divisible by the down-sampling factor");
  for (int i = 0; i < cfg->num_blocks; ++i) {
    const int c = cfg->block_out_channels[i];
    DFH_REQUIRE(c % 8 == 0 && c % cfg->norm_num_groups == 0, "block_out_channels must be multiples of 8 and of norm_num_groups");
    DFH_REQUIRE(cfg->num_heads[i] > 0 && c % cfg->num_heads[i] == 0, "channels not divisible by heads");
    const int d = c / cfg->num_heads[i];
    DFH_REQUIRE(d == 32 || d == 40 || d == 64 || d == 80 || d == 128 || d == 160, "unsupported head dim (32,40,64,80,128,160)");
  }
  dfh_unet* u = new dfh_unet();
  u->cfg = *cfg;
  if (u->build()) { delete u; return -1; }
  *out = u;
  return 0;
}

void dfh_unet_destroy(dfh_unet* u) { delete u; }
int dfh_unet_num_params(const dfh_unet* u) { return (int)u->params.size(); }
const char* dfh_unet_param_name(const dfh_unet* u, int i) { return u->params[i].name.c_str(); }
int dfh_unet_param_ndim(const dfh_unet* u, int i) { return (int)u->params[i].shape.size(); }
int dfh_unet_param_dim(const dfh_unet* u, int i, int d) { return u->params[i].shape[d]; }
size_t dfh_unet_arena16_bytes(const dfh_unet* u) { return u->a16 * 2 + 256; }
size_t dfh_unet_arena32_bytes(const dfh_unet* u) { return u->a32 * 4 + 256; }

size_t dfh_unet_workspace_bytes(dfh_unet* u, int batch) {
  if (batch <= 0) return 0;
  u->run(nullptr, 0, nullptr, nullptr, 0, nullptr, batch, nullptr, /*dry=*/true);
  return u->plan_total;
}

int dfh_unet_bind(dfh_unet* u, void* arena16, void* arena32, void* workspace, size_t workspace_bytes, int max_batch) {
  DFH_REQUIRE(u && arena16 && arena32 && workspace, "null argument");
  DFH_REQUIRE(((uintptr_t)arena16 | (uintptr_t)arena32 | (uintptr_t)workspace) % 256 == 0, "buffers must be 256-byte aligned");
  u->run(nullptr, 0, nullptr, nullptr, 0, nullptr, max_batch, nullptr, /*dry=*/true);
  DFH_REQUIRE(workspace_bytes >= u->plan_total, "workspace smaller than dfh_unet_workspace_bytes(max_batch)");
  u->arena16 = (bf16_t*)arena16; u->arena32 = (float*)arena32;
  u->ws = (char*)workspace; u->ws_bytes = workspace_bytes; u->max_batch = max_batch;
  return 0;
}

int dfh_unet_enable_fp8(dfh_unet* u) {
  DFH_REQUIRE(u != nullptr, "null argument");
  DFH_REQUIRE(u->ws == nullptr, "dfh_unet_enable_fp8 must precede dfh_unet_workspace_bytes / dfh_unet_bind");
  return u->enable_fp8();
}
// before dfh_unet_enable_fp8: also run the self-attention products on the e4m3 MFMA (off by default: slower and less accurate than the bf16 kernels)
int dfh_unet_enable_fp8_attention(dfh_unet* u, int on) {
  DFH_REQUIRE(u != nullptr, "null context");
  DFH_REQUIRE(!u->fp8, "dfh_unet_enable_fp8_attention must precede dfh_unet_enable_fp8");
  u->fp8_attention = on != 0;
  return 0;
}
size_t dfh_unet_arena8_bytes(const dfh_unet* u) { return u && u->fp8 ? u->a8 : 0; }
int dfh_unet_bind_fp8(dfh_unet* u, void* arena8) {
  DFH_REQUIRE(u && arena8, "null argument");
  DFH_REQUIRE(u->fp8, "dfh_unet_enable_fp8 not called");
  DFH_REQUIRE((uintptr_t)arena8 % 256 == 0, "buffers must be 256-byte aligned");
  u->arena8 = (unsigned char*)arena8;
  return 0;
}

int dfh_unet_pack(dfh_unet* u, const float* const* master_params, int count, void* stream) {
  DFH_REQUIRE(u && master_params, "null argument");
  return u->pack(master_params, count, (hipStream_t)stream);
}

// The dup-tail hint is ONE-SHOT: whatever happens inside the forward entry points -- including every early DFH_REQUIRE return -- the hint is
// gone when they return, so a refused call can never leak it into a later forward on other inputs (round-5 advisor).
struct DupTailGuard {
  dfh_unet* u;
  ~DupTailGuard() { if (u) u->dup_tail = 0; }
};

int dfh_unet_forward(dfh_unet* u, const void* sample, int sample_bf16, const float* timestep, const void* ehs, int ehs_bf16,
                     float* out, int batch, void* stream) {
  DupTailGuard guard{u};
  DFH_REQUIRE(u && sample && timestep && ehs && out, "null argument");
  DFH_REQUIRE(u->ws != nullptr, "dfh_unet_bind not called");
  DFH_REQUIRE(batch > 0 && batch <= u->max_batch, "batch exceeds the bound max_batch");
  DFH_REQUIRE(!u->fp8 || u->arena8, "fp8 enabled but dfh_unet_bind_fp8 not called");
  if (batch != u->plan_batch) u->run(nullptr, 0, nullptr, nullptr, 0, nullptr, batch, nullptr, true);
  DFH_REQUIRE(u->plan_total <= u->ws_bytes, "workspace too small for this batch");
  return u->run(sample, sample_bf16, timestep, ehs, ehs_bf16, out, batch, (hipStream_t)stream, false);
}

size_t dfh_unet_run_cache_bytes(const dfh_unet* u, int batch, int n_timesteps) {
  return (u && batch > 0 && n_timesteps >= 0) ? u->run_cache_bytes(batch, n_timesteps) : 0;
}

int dfh_unet_run_cache(dfh_unet* u, const void* ehs, int ehs_bf16, int batch, const float* timesteps, int n_timesteps, void* cache,
                       size_t cache_bytes, void* stream) {
  DFH_REQUIRE(u && ehs && cache && (timesteps || n_timesteps == 0), "null argument");
  DFH_REQUIRE(u->ws != nullptr, "dfh_unet_bind not called");
  DFH_REQUIRE(batch > 0 && batch <= u->max_batch, "batch exceeds the bound max_batch");
  DFH_REQUIRE((uintptr_t)cache % 256 == 0 && cache_bytes >= u->run_cache_bytes(batch, n_timesteps), "run cache too small or misaligned");
  return u->run_cache(ehs, ehs_bf16, batch, timesteps, n_timesteps, cache, (hipStream_t)stream);
}

int dfh_unet_set_dup_tail(dfh_unet* u, int images) {
  DFH_REQUIRE(u && images >= 0, "bad argument");
  u->dup_tail = images;
  return 0;
}

int dfh_unet_forward_cached(dfh_unet* u, const void* sample, int sample_bf16, const void* cache, int batch, int n_timesteps, int t_index,
                            float* out, void* stream) {
  DupTailGuard guard{u};
  DFH_REQUIRE(u && sample && cache && out, "null argument");
  DFH_REQUIRE(u->ws != nullptr, "dfh_unet_bind not called");
  DFH_REQUIRE(batch > 0 && batch <= u->max_batch, "batch exceeds the bound max_batch");
  DFH_REQUIRE(t_index >= 0 && t_index < n_timesteps, "timestep index outside the cached schedule");
  DFH_REQUIRE(!u->fp8 || u->arena8, "fp8 enabled but dfh_unet_bind_fp8 not called");
  if (batch != u->plan_batch) u->run(nullptr, 0, nullptr, nullptr, 0, nullptr, batch, nullptr, true);
  DFH_REQUIRE(u->plan_total <= u->ws_bytes, "workspace too small for this batch");
  dfh_unet::RunCache rc;
  rc.kx = (const bf16_t*)cache;
  rc.vxt = (const bf16_t*)((const char*)cache + dfh_unet::cache_kx_bytes(*u, batch));
  rc.temb_row = (const float*)((const char*)cache + dfh_unet::cache_kx_bytes(*u, batch) + dfh_unet::cache_vxt_bytes(*u, batch)) +
                (size_t)t_index * u->temb_total;
  if (u->fp8)
    rc.xamax = (const float*)((const char*)cache + dfh_unet::cache_kx_bytes(*u, batch) + dfh_unet::cache_vxt_bytes(*u, batch) +
                              (((size_t)n_timesteps * u->temb_total * 4 + 255) & ~(size_t)255));
  return u->run(sample, sample_bf16, nullptr, nullptr, 0, out, batch, (hipStream_t)stream, false, &rc);
}

}  // extern "C"

// debug tap: NHWC bf16 -> NCHW fp32
namespace {
__global__ void tap_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int B, int HW, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * HW * C) return;
  const int c = (int)(i % C);
  const long bp = i / C;
  const int p = (int)(bp % HW), b = (int)(bp / HW);
  dst[((long)b * C + c) * HW + p] = bf2f(src[i]);
}
}  // namespace

extern "C" int dfh_unet_debug_tap(dfh_unet* u, const char* name, float* dst, size_t dst_floats, void* stream) {
  DFH_REQUIRE(u && name && dst, "null argument");
  auto it = u->taps.find(name);
  DFH_REQUIRE(it != u->taps.end(), std::string("unknown tap ") + name);
  const Tensor& t = it->second;
  const long n = (long)u->last_batch * t.H * t.W * t.C;
  DFH_REQUIRE((size_t)n <= dst_floats, "tap destination too small");
  hipLaunchKernelGGL(tap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t.p, dst, u->last_batch,
                     t.H * t.W, t.C);
  return dfh::check_launch("tap_kernel");
}
