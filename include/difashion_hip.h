/* libdifashion_hip.so -- C ABI of the MI355X (gfx950) DiFashion denoising path.
 *
 * The reference has no C ABI / FFI: its boundary is Python duck-typing on two objects held by
 * DiFashion (``self.unet``, ``self.noise_scheduler``).  This header is what the
 * Python host side (difashion_amd/, mirroring the diffusers call signatures) binds through ctypes;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.  Every entry point
 * names the reference interface it replaces (DiFashion/models/difashion.py = "df.py"; diffusers
 * 0.18.2 classes are external, pinned by the reference README.md:27).
 *
 * Conventions: plain pointers and sizes only (no torch types); all pointers are DEVICE pointers
 * unless marked host; ``stream`` is a hipStream_t passed as void*; every function returns 0 on
 * success or a negative status, with the message available from dfh_last_error(); nothing
 * throws; nothing allocates device memory (callers own arenas / workspaces); a context is
 * re-entrant per instance and not thread-safe when shared.
 *
 * dtypes: "bf16" = raw uint16 bfloat16 bits.  Activations inside the library are NHWC bf16; the
 * public tensors keep the reference's layouts (NCHW latents, [B][77][D] text states).
 */
#ifndef DIFASHION_HIP_H
#define DIFASHION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point, a struct layout or the meaning of an argument changes (round 4: 4).  The Python host side
 * (difashion_amd/_lib.py ABI_VERSION) refuses a library that reports another number: a stale .so next to new Python, or the reverse,
 * fails at load time instead of at a symbol lookup or silently. */
#define DFH_ABI_VERSION 7
#define DFH_MAX_BLOCKS 4

/* ------------------------------------------------------------------ library */
int dfh_abi_version(void);
const char* dfh_last_error(void);          /* message of the last failing call on this thread */
const char* dfh_build_info(void);          /* "gfx950 ..." */

/* Per-kernel-class timing with HIP events recorded on the launch stream (bench.py "roofline"):
 * between begin and end every launch is bracketed by two events; end synchronises and sums, per
 * class, launches / elapsed ms / ALGORITHMIC flops and bytes (DESIGN.md "Measurement").  Not for
 * use inside a timed region whose wall clock is being reported (adds two event records per launch). */
typedef struct dfh_prof_class {
  char name[32];
  int launches;
  double ms, flops, bytes;
} dfh_prof_class;
int dfh_prof_begin(void);
int dfh_prof_end(dfh_prof_class* out, int max_classes);   /* returns the number of classes written (7) */
/* flops of the reference algorithm that the launches of the window reported but did not execute: the upsampler convs
 * run 4 of the 9 taps (phase planes), the Winograd convs 16 multiply-adds per 2x2 outputs and channel instead of 36 */
double dfh_prof_saved_flops(void);

/* Launch census: how many times each kernel family was launched since the last reset (host-side counters bumped by the launchers).
 * Test infrastructure for "which kernels did this walk take" (tests/test_gpu_unet.py: the batch-16 forward of the bench workload). */
void dfh_census_reset(void);
int dfh_census_count(void);
const char* dfh_census_name(int i);
long dfh_census_get(int i);

/* ------------------------------------------------------------------ U-Net context
 * Replaces: diffusers UNet2DConditionModel as constructed at df.py:77-93 (in_channels widened to
 * 8) and called at df.py:249-253 (training) and df.py:518-523 (sampling). */
typedef struct dfh_unet_config {
  int sample_size;                  /* latent H = W (64) */
  int in_channels;                  /* 8: [latent | history latent], df.py:83-85 */
  int out_channels;                 /* 4 */
  int num_blocks;                   /* 4 */
  int block_out_channels[DFH_MAX_BLOCKS]; /* 320,640,1280,1280 */
  int layers_per_block;             /* 2 */
  int cross_attention_dim;          /* 768 (SD-1.5) / 1024 (SD-2) */
  int num_heads[DFH_MAX_BLOCKS];    /* diffusers "attention_head_dim": 8,8,8,8 / 5,10,20,20 */
  int down_attn[DFH_MAX_BLOCKS];    /* 1,1,1,0 */
  int use_linear_projection;        /* only changes parameter shapes (1x1 conv == linear in NHWC) */
  int norm_num_groups;              /* 32 */
  float norm_eps;                   /* 1e-5 */
  int text_len;                     /* 77 */
} dfh_unet_config;

typedef struct dfh_unet dfh_unet;

int dfh_unet_create(const dfh_unet_config* cfg, dfh_unet** out);   /* host-only work */
void dfh_unet_destroy(dfh_unet* u);

/* Parameter table, in diffusers state-dict order/names: the single source of truth
 * the Python module builds its nn.Parameters from (checkpoint drop-in, df.py:77-79, train.py:524-547). */
int dfh_unet_num_params(const dfh_unet* u);
const char* dfh_unet_param_name(const dfh_unet* u, int i);
int dfh_unet_param_ndim(const dfh_unet* u, int i);
int dfh_unet_param_dim(const dfh_unet* u, int i, int d);

/* Packed-weight arenas (bf16 GEMM operands in kernel layout, fp32 vectors) and activation workspace.
 * Sizes are bytes; buffers are caller-allocated device memory, 256-byte aligned. */
size_t dfh_unet_arena16_bytes(const dfh_unet* u);
size_t dfh_unet_arena32_bytes(const dfh_unet* u);
size_t dfh_unet_workspace_bytes(dfh_unet* u, int batch);
int dfh_unet_bind(dfh_unet* u, void* arena16, void* arena32, void* workspace, size_t workspace_bytes, int max_batch);

/* fp8 linears inside the U-Net walk (inference): call dfh_unet_enable_fp8 right after create (it changes the workspace plan),
 * allocate dfh_unet_arena8_bytes, bind it; dfh_unet_pack then also refreshes the e4m3 copies + per-channel scales. */
int dfh_unet_enable_fp8(dfh_unet* u);
/* optional, BEFORE dfh_unet_enable_fp8: also run the self-attention products QK^T / PV on the e4m3 MFMA (csrc/attention_fp8.hip; head dims
 * 40 / 80 / 160, whole 64-key tiles).  Off by default: measured slower than the bf16 kernels on this model (profiles/r04). */
int dfh_unet_enable_fp8_attention(dfh_unet* u, int on);
size_t dfh_unet_arena8_bytes(const dfh_unet* u);
int dfh_unet_bind_fp8(dfh_unet* u, void* arena8);

/* fp32 master parameters (device pointers, table order) -> packed arenas.  Call after every
 * weight update (optimizer step / load_state_dict). */
int dfh_unet_pack(dfh_unet* u, const float* const* master_params, int count, void* stream);

/* noise_pred = unet(sample, timestep, encoder_hidden_states).sample
 *   sample   : [B][in_channels][H][W]  fp32 (sample_bf16=0) or bf16 (1), NCHW as in df.py:216/:515
 *   timestep : [B] fp32 (the host side expands the 0-d / int / (B,) forms, df.py:251,:520)
 *   ehs      : [B][text_len][cross_attention_dim] fp32 (ehs_bf16=0) or bf16 (1)
 *   out      : [B][out_channels][H][W] fp32, NCHW */
int dfh_unet_forward(dfh_unet* u, const void* sample, int sample_bf16, const float* timestep,
                     const void* ehs, int ehs_bf16, float* out, int batch, void* stream);

/* Per-run constants of a sampling loop.  In fashion_generation the prompt states are fixed for the whole run (df.py:340-357, stacked
 * once at :388-427) and the timesteps are the schedule's list (:356, :456), the same for every row of the CFG batch: the
 * cross-attention K / V^T of all transformer blocks and the time-embedding rows (time_embedding MLP + all 22 time_emb_proj) depend on
 * nothing else.  dfh_unet_run_cache computes them once into a caller-owned, 256-byte-aligned buffer of dfh_unet_run_cache_bytes;
 * dfh_unet_forward_cached is dfh_unet_forward for a batch whose rows all sit at timesteps[t_index], with those launches skipped
 * (5 GEMMs + 2 small kernels per step).  The cache is valid until the weights are re-packed or the text states change. */
size_t dfh_unet_run_cache_bytes(const dfh_unet* u, int batch, int n_timesteps);
int dfh_unet_run_cache(dfh_unet* u, const void* ehs, int ehs_bf16, int batch, const float* timesteps /* device [n] */, int n_timesteps,
                       void* cache, size_t cache_bytes, void* stream);
/* One-shot hint for the NEXT dfh_unet_forward / dfh_unet_forward_cached call: the last `images` images of its batch have the same
 * sample and timestep as the `images` before them and differ only in their encoder_hidden_states -- the prompt-only branch of
 * classifier-free guidance (difashion.py:388-427, 494-512: category_prompts vs null_prompts over the same latent / mutual / history
 * input).  conv_in, the first resnet and the first transformer block up to its self-attention are then computed once for the pair.
 * The caller vouches for the equality (DFH_CHECK_DUP=1 verifies it with a synchronous compare: debugging aid); the call clears the hint. */
int dfh_unet_set_dup_tail(dfh_unet* u, int images);
int dfh_unet_forward_cached(dfh_unet* u, const void* sample, int sample_bf16, const void* cache, int batch, int n_timesteps, int t_index,
                            float* out, void* stream);

/* ------------------------------------------------------------------ training step (train.py:691-716)
 * Replaces torch autograd through the U-Net: accelerator.backward(loss) at DiFashion/train.py:699 for the module
 * called at DiFashion/models/difashion.py:249-253.  Shares arena16/arena32 with the inference path
 * (dfh_unet_bind + dfh_unet_pack first) and adds
 *   arena16t  : bf16 transposed/flipped weight packs for the data-gradient GEMMs   (dfh_unet_arena16t_bytes)
 *   grad16/32 : fp32 gradient arenas in the PACKED layouts of arena16 / arena32      (dfh_unet_grad16/32_bytes)
 *   workspace : every activation of one step + gradient buffers                      (dfh_unet_train_workspace_bytes)
 * All pointers 256-byte aligned device memory; arena16t must be zero-filled before the first pack. */
size_t dfh_unet_arena16t_bytes(dfh_unet* u);
size_t dfh_unet_grad16_bytes(const dfh_unet* u);
size_t dfh_unet_grad32_bytes(const dfh_unet* u);
size_t dfh_unet_train_workspace_bytes(dfh_unet* u, int batch);
int dfh_unet_bind_train(dfh_unet* u, void* arena16t, void* grad16, void* grad32, void* workspace, size_t workspace_bytes,
                        int max_batch);
/* master parameters -> arena16t; call together with dfh_unet_pack after every weight update */
int dfh_unet_pack_train(dfh_unet* u, const float* const* master_params, int count, void* stream);
/* dfh_unet_pack + dfh_unet_pack_train in one pass over the master parameters (each weight is read once and written to both arenas);
 * what a training step calls after the optimizer moved the weights.  Needs dfh_unet_bind_train. */
int dfh_unet_pack_all(dfh_unet* u, const float* const* master_params, int count, void* stream);
/* out != NULL: from now on the gradient un-pack at the end of dfh_unet_backward / _backward_finish also writes
 * out[0] = sum of the squares of every gradient value it wrote into master_grads (the final values, also when it accumulates): the
 * clip norm of train.py:700 without another pass over the gradients.  Deterministic (per-block partials, fixed-order reduce).
 * out == NULL switches it off.  The float must stay valid until then. */
int dfh_unet_grad_sumsq(dfh_unet* u, float* out);
/* same arguments and result as dfh_unet_forward; keeps the activations the backward needs */
int dfh_unet_forward_train(dfh_unet* u, const void* sample, int sample_bf16, const float* timestep,
                           const void* ehs, int ehs_bf16, float* out, int batch, void* stream);
/* Backward of the LAST dfh_unet_forward_train (one backward per forward).
 *   d_out        : [B][out_channels][H][W] fp32, gradient of the loss wrt the noise prediction
 *   d_sample     : [B][in_channels][H][W] fp32 or NULL (gradient wrt the assembled input -> MutualEncoder)
 *   master_grads : table-order device pointers to the fp32 .grad of each parameter; gradients are ADDED
 *                  (entries may be NULL to skip a frozen parameter).  encoder_hidden_states gets no gradient
 *                  (frozen CLIP text states, df.py:229-247).
 *   overwrite    : 1 = the gradients hold nothing yet (first backward after zero_grad): they are STORED, which saves
 *                  zero-filling and re-reading 3.4 GB; 0 = added (gradient accumulation over micro-batches). */
int dfh_unet_backward(dfh_unet* u, const float* d_out, float* d_sample, float* const* master_grads, int count, int overwrite,
                      void* stream);

/* The same backward in pieces, for data-parallel training: the caller all-reduces finished ranges of the packed fp32
 * gradient arena (the ``grad16`` buffer of dfh_unet_bind_train) on a second stream while the walk continues -- the
 * reference's DDP buckets (accelerate / torch DDP around train.py:611,699) without an autograd hook per parameter.
 *   begin  : arms the walk; ``bucket_floats`` = granularity of the ranges (floats of grad16).  Returns the tape length.
 *   next   : runs layers from the back of the tape until some range of grad16 is FINAL (no later layer writes into it),
 *            returns 1 with [*lo, *hi) in floats, 0 when the tape is exhausted (every range has been handed out), < 0 on error.
 *   finish : un-packs grad16 / grad32 (whatever they hold by then, e.g. the all-reduced average) into the master
 *            gradients exactly as dfh_unet_backward does.  dfh_unet_backward == begin + next until 0 + finish. */
int dfh_unet_backward_begin(dfh_unet* u, const float* d_out, float* d_sample, size_t bucket_floats, void* stream);
int dfh_unet_backward_next(dfh_unet* u, size_t* lo, size_t* hi, void* stream);
int dfh_unet_backward_finish(dfh_unet* u, float* const* master_grads, int count, int overwrite, void* stream);

/* Copies a named NHWC bf16 intermediate of the LAST forward into ``dst`` as fp32 NCHW (layer-level
 * parity tests).  Names: "conv_in", "down0".."down3", "mid", "up0".."up3". */
int dfh_unet_debug_tap(dfh_unet* u, const char* name, float* dst, size_t dst_floats, void* stream);

/* ------------------------------------------------------------------ AutoencoderKL
 * Replaces diffusers' AutoencoderKL as the reference calls it: vae.encode(images).latent_dist (df.py:129,144,376,435-437)
 * and vae.decode(latents / scaling_factor) (df.py:580).  Same ownership rules as dfh_unet: caller-owned, zero-filled
 * arenas + workspace, fp32 master parameters in table order with diffusers state-dict names. */
typedef struct dfh_vae_config {
  int in_channels;                 /* 3 */
  int out_channels;                /* 3 */
  int latent_channels;             /* 4 */
  int num_blocks;                  /* 4 */
  int block_out_channels[DFH_MAX_BLOCKS];   /* 128, 256, 512, 512 */
  int layers_per_block;            /* 2 */
  int norm_num_groups;             /* 32 */
} dfh_vae_config;
typedef struct dfh_vae dfh_vae;
int dfh_vae_create(const dfh_vae_config* cfg, dfh_vae** out);
void dfh_vae_destroy(dfh_vae* u);
int dfh_vae_num_params(const dfh_vae* u);
const char* dfh_vae_param_name(const dfh_vae* u, int i);
int dfh_vae_param_ndim(const dfh_vae* u, int i);
int dfh_vae_param_dim(const dfh_vae* u, int i, int d);
size_t dfh_vae_arena16_bytes(const dfh_vae* u);
size_t dfh_vae_arena32_bytes(const dfh_vae* u);
/* encode != 0: size = image side; else size = latent side */
size_t dfh_vae_workspace_bytes(dfh_vae* u, int encode, int batch, int size);
int dfh_vae_bind(dfh_vae* u, void* arena16, void* arena32, void* workspace, size_t workspace_bytes);
int dfh_vae_pack(dfh_vae* u, const float* const* master_params, int count, void* stream);
/* images [B][in_channels][S][S] fp32 -> moments [B][2*latent_channels][S/8][S/8] fp32 = [mean | logvar] */
int dfh_vae_encode(dfh_vae* u, const float* images, float* moments, int batch, int image_size, void* stream);
/* latents [B][latent_channels][s][s] fp32 -> images [B][4][8s][8s] fp32 (out_channels = 3 used, the 4th plane is padding) */
int dfh_vae_decode(dfh_vae* u, const float* latents, float* images, int batch, int latent_size, void* stream);

/* ------------------------------------------------------------------ CLIP text encoder
 * Replaces transformers' CLIPTextModel as the reference builds and calls it: CLIPTextModel.from_pretrained(..., subfolder="text_encoder")
 * at df.py:70-72, text_encoder(input_ids)[0] at df.py:224,234 (every training batch) and df.py:340-342,352 (every sampling call).
 * fp32 end to end (csrc/clip.hip: linears on v_mfma_f32_16x16x4_f32): the prompts are a closed set encoded once per run, so the
 * encoder is built for agreement with the fp32 class (1e-6), not for speed.  The fp32 master parameters are read in place
 * (nn.Linear layout): no arenas, no pack step; parameter table in transformers 4.32.1 state-dict order / names ("text_model.*"). */
typedef struct dfh_clip_config {
  int vocab_size;                  /* 49408 */
  int hidden_size;                 /* 768 (SD-1.5, CLIP ViT-L/14) / 1024 (SD-2, OpenCLIP ViT-H/14) */
  int intermediate_size;           /* 3072 / 4096 */
  int num_hidden_layers;           /* 12 / 23 */
  int num_attention_heads;         /* 12 / 16 */
  int max_position_embeddings;     /* 77 (<= 128) */
  int hidden_act;                  /* 1 quick_gelu (SD-1.5) / 2 gelu, erf form (SD-2) */
  float layer_norm_eps;            /* 1e-5 */
} dfh_clip_config;
typedef struct dfh_clip dfh_clip;
int dfh_clip_create(const dfh_clip_config* cfg, dfh_clip** out);   /* host-only work */
void dfh_clip_destroy(dfh_clip* c);
int dfh_clip_num_params(const dfh_clip* c);
const char* dfh_clip_param_name(const dfh_clip* c, int i);
int dfh_clip_param_ndim(const dfh_clip* c, int i);
int dfh_clip_param_dim(const dfh_clip* c, int i, int d);
size_t dfh_clip_workspace_bytes(const dfh_clip* c, int batch, int seq_len);
/* outputs = text_encoder(input_ids):
 *   master_params     : HOST array of `count` device pointers to the fp32 parameters, table order
 *   input_ids         : [batch][seq_len] int64 (tokenizer output, data_utils.py:107-110); causal mask only, no padding mask (the
 *                       reference passes input_ids alone)
 *   last_hidden_state : [batch][seq_len][hidden_size] fp32 = outputs[0], after final_layer_norm
 *   pooler_output     : [batch][hidden_size] fp32 or NULL; the row at argmax(input_ids) when eos_token_id == 2 (transformers 4.32.1),
 *                       else at the first eos_token_id
 *   hidden_states     : NULL, or [num_hidden_layers + 1][batch][seq_len][hidden_size] fp32 = output_hidden_states=True (embeddings
 *                       output, then every layer's output; layer-level parity tests)
 *   workspace         : >= dfh_clip_workspace_bytes(batch, seq_len), 256-byte aligned */
int dfh_clip_encode(dfh_clip* c, const float* const* master_params, int count, const int64_t* input_ids, float* last_hidden_state,
                    float* pooler_output, int eos_token_id, float* hidden_states, void* workspace, size_t workspace_bytes, int batch,
                    int seq_len, void* stream);

/* ------------------------------------------------------------------ op-level entry points (tests, profiling)
 * ResnetBlock2D conv3x3 / Downsample2D / Upsample2D / 1x1 conv / Linear, as one implicit GEMM:
 *   out[M][N] = act( conv3x3(x) (+ a0 . W[:, k0:] + a1 . W[:, k1:]) + bias + rowvec[b] ) + resid */
typedef struct dfh_gemm_desc {
  const void* conv_src; int conv_c; int conv;   /* conv != 0: 3x3 pad 1 over NHWC bf16 [B][Hin][Win][conv_c] */
  int batch, Hin, Win, stride, upsample;        /* stride 1|2; upsample: nearest 2x before the conv */
  const void* a0; int a0_c; const void* a1; int a1_c; /* plain K segments, rows [M][c] bf16 */
  const void* W; int ldw;                       /* bf16 [N][ldw], K-contiguous, segments in the order above */
  int M, N;
  const float* bias;                            /* [N] or NULL */
  const float* rowvec; int rv_ld, rv_off, rows_per_b;  /* + rowvec[m / rows_per_b][rv_off + n] (time embedding) */
  const void* resid; int ld_res;                /* + resid[m][n] bf16 */
  int act;                                      /* 0 none 1 silu 2 leaky_relu(0.01) 3 tanh 4 GEGLU (W/bias pre-interleaved) */
  void* out; int ld_out; int out_mode;          /* 0 bf16 [M][ld] 1 bf16 [b][N][ld] 2 fp32 [M][ld] 3 fp32 [b][N][ld] */
  float* partial; size_t partial_floats;        /* split-K slabs (dfh_gemm_partial_floats) */
  const void* zero_page;                        /* >= 256 zero bytes */
  int force_tile, force_split, force_order;     /* 0,0,-1 = heuristics; force_order 2 / 3 = tile ids n-major / m-major */
  float* gstat; int gstat_cpg, gstat_hw;        /* optional: GroupNorm statistics of the output for the consumer (channels per group,
                                                 * pixels per image): [image][group][hw / rows][2] sums / sums of squares, rows = 256
                                                 * or 128 (dfh_gemm_gstat reports which); only through dfh_gemm_gstat (dfh_gemm ignores the three fields) */
  size_t w_img_stride;                          /* 0, or per-IMAGE weights: rows [i * rows_per_b, (i + 1) * rows_per_b) multiply W + i * w_img_stride
                                                 * (elements) -- dfh_groupnorm_fold; 128-row-tile launches without conv taps / split-K */
} dfh_gemm_desc;
size_t dfh_gemm_partial_floats(const dfh_gemm_desc* d);
int dfh_gemm(const dfh_gemm_desc* d, void* stream);
/* dfh_gemm + the output statistics for the consuming GroupNorm (d->gstat ..., sized for gstat_hw / 128 chunks); *written = the pixel rows
 * per statistics chunk they were produced with (256 or 128: pass gstat_hw / *written as the chunk count to dfh_groupnorm_pre), 0 when the
 * launch ran on a kernel that cannot (the caller then runs plain dfh_groupnorm) */
int dfh_gemm_gstat(const dfh_gemm_desc* d, void* stream, int* written);
/* GroupNorm folded into the 1x1 projection that consumes it (transformer entry, difashion.py:249-253): from x [B][HW][C] (statistics: the
 * producer's partials `pre` [B][G][pre_chunks][2], or NULL -> summed here into `partial`, >= B * 64 * G * 2 floats), gamma / beta and the
 * projection W [N][ldw] (+ bias) it writes per-image weights Wimg [B][N][C] = bf16(W gamma rstd) and the row vectors rv [B][N] =
 * bias + W . beta - Wimg . mean; then  proj(GroupNorm(x)) = dfh_gemm(a0 = x, W = Wimg, w_img_stride = N * C, rowvec = rv, rows_per_b = HW). */
int dfh_groupnorm_fold(const void* x, int B, int HW, int C, int G, const float* gamma, const float* beta, float eps, const float* pre,
                       int pre_chunks, float* partial, const void* W, int ldw, int N, const float* bias, void* Wimg, float* rv, void* stream);
/* Weight gradient of the same op (train.py:699 backward): dW[n][k] += sum_m dY[m][n] * A[m][k], where A is the
 * forward operand described by d (conv_src / a0 / a1 segments, M, N, zero_page; W / out / epilogue fields unused).
 * dW: fp32 [N][ldw] in the PACKED weight layout, accumulated (dW += ...; the sum over pixel slices is a fixed-order slab reduce, so
 * reruns are bit-identical).  msplit: 0 = the launcher's plan; n > 0 = n equal pixel slices of every 160 x 160 output tile;
 * -n = whole tiles in full rounds of 512 blocks and the remaining tiles in n slices. */
int dfh_gemm_wgrad(const dfh_gemm_desc* d, const void* dY, int ldy, float* dW, int ldw, int msplit, void* stream);
/* fp32 slab floats (d->partial / d->partial_floats) that call needs when it splits the pixel range; 0 = none */
size_t dfh_gemm_wgrad_partial_floats(const dfh_gemm_desc* d, int msplit);
/* the decomposition that call would launch (host code, no GPU needed): output tiles, how many of them run whole, pixel slices of the rest */
int dfh_gemm_wgrad_plan(const dfh_gemm_desc* d, int msplit, int* tiles, int* whole_tiles, int* slices);
/* out[g][n] += sum over the rows of group g of Y[m][n] (bias gradient: groups = 1; time-embedding gradient:
 * groups = batch, rows_per_group = H*W). */
int dfh_colsum(const void* Y, int ldy, int N, int groups, int rows_per_group, float* out, int ld_out, void* stream);

/* GroupNorm(32, C, eps)(+SiLU) over NHWC bf16, optional fused channel concat of two sources
 * (ResnetBlock2D.norm1/norm2, conv_norm_out, Transformer2DModel.norm).  partial: >= B*64*G*2 floats */
int dfh_groupnorm(const void* src0, int c0, const void* src1, int c1, int batch, int hw, int groups,
                  const float* gamma, const float* beta, float eps, int silu, void* out, float* partial, void* stream);
/* the same with the statistics supplied by the producing dfh_gemm (gstat, chunks = hw / 256): one launch, the tensor is read once */
int dfh_groupnorm_pre(const void* src, int c, int batch, int hw, int groups, const float* gamma, const float* beta, float eps,
                      int silu, void* out, const float* gstat, int chunks, float* stats_out, void* stream);
/* LayerNorm over the last dim of [M][C] bf16 (BasicTransformerBlock.norm1/2/3) */
int dfh_layernorm(const void* x, const float* gamma, const float* beta, void* y, int M, int C, float eps, void* stream);
/* dfh_gemm with a SECOND destination: output columns [n_split, N) leave transposed per batch of d->rows_per_b rows into out2
 * ([batch][N - n_split][ld_out2] bf16), columns [0, n_split) into d->out as usual -- attention's q | k and V^T from one launch over the
 * rows both projections share (diffusers Attention.to_q / to_k / to_v of attn1).  Fails when the launch heuristics would split K or
 * pick a column tile that does not divide n_split. */
int dfh_gemm_out2(const dfh_gemm_desc* d, void* out2, int ld_out2, int n_split, void* stream);
/* ---- LayerNorm folded into the projections around it (inference walk of BasicTransformerBlock: x + attn1(LN1(x)), + attn2(LN2(x)),
 * + ff(LN3(x)); reached from df.py:518-523).  LN(x) . W^T = rstd * (x . W'^T - mean * s) + b' with W' = W * gamma, s = rowsum(W'),
 * b' = bias + W . beta: the consumer GEMM runs on the raw rows and fixes them up in its epilogue, the statistics come from the
 * epilogue of the GEMM that produced x -- no LayerNorm launch, no normalised copy of the tensor in HBM.
 *   dfh_ln_fold : packed bf16 W [N][ldw] -> WF [N][K] bf16, s [N], b [N] (bias may be NULL)
 *   dfh_gemm_ln : dfh_gemm plus, as producer, rowstat (per output row and column tile (mean, centred sum of squares),
 *                 [N / bn][M][2] floats; *rowstat_bn = bn, 0 when this launch could not write them) and / or, as consumer, ln_stat =
 *                 the producer's rowstat (ln_parts column tiles of ln_cnt columns each), ln_s = s, d->bias = b', d->W = WF */
int dfh_ln_fold(const void* W, int ldw, const float* gamma, const float* beta, const float* bias, void* WF, float* s, float* b, int N, int K,
                void* stream);
int dfh_gemm_ln(const dfh_gemm_desc* d, float* rowstat, int* rowstat_bn, const float* ln_stat, int ln_parts, int ln_cnt, float ln_eps,
                const float* ln_s, void* stream);
/* ---- nearest-2x upsample + 3x3 conv (diffusers Upsample2D: F.interpolate(scale_factor=2, mode="nearest") then conv, the up-block
 * upsamplers reached from df.py:518-523) as FOUR 2x2 convs over the source image: output pixel (2y + py, 2x + px) sees source rows
 * {y - 1 + py, y + py} and columns alike, each with the sum of the 3x3 taps that land on it -- 4/9 of the multiply-adds.
 *   dfh_ups_phase_fold : packed bf16 W [N][ldw >= 9 C] (tap-major columns) -> WP [4][N][4 C] bf16, the summed taps per phase py * 2 + px
 *   dfh_conv_up2x      : src [batch][H][W][C] bf16 -> out [batch][2H][2W][N] bf16 = conv3x3(upsample2x(src)) + bias, one launch */
int dfh_ups_phase_fold(const void* W, int ldw, void* WP, int N, int C, void* stream);
int dfh_conv_up2x(const void* src, int batch, int H, int W, int C, const void* WP, int N, const float* bias, void* out,
                  const void* zero_page, void* stream);
/* ---- Winograd F(2x2, 3x3) for stride-1 3x3 convs (ResnetBlock2D.conv1 / conv2 at the deep levels; df.py:249-253,518-523): sixteen
 * transform-domain GEMMs over the 2x2 output tiles instead of nine taps per pixel (2.25 x fewer multiply-adds), U / V / M rounded to bf16.
 *   dfh_wino_weights : packed bf16 W [N][ldw >= 9 C] -> U [16][N][C] bf16 (G g G^T, fp32 arithmetic, rounded once)
 *   dfh_conv3x3_wino : src [batch][H][W][C] bf16 (H, W even) -> out [batch][H][W][N] bf16 = conv3x3(src) + bias (+ rowvec row of the
 *                      image: rowvec[b * rv_ld + rv_off + n]) (+ resid [batch][H][W][N] bf16); three launches (input transform, one
 *                      batched GEMM, output transform) over `scratch` (dfh_conv3x3_wino_scratch_bytes, 256-byte aligned) */
int dfh_wino_weights(const void* W, int ldw, void* U, int N, int C, int blocked, void* stream);
/* the input transform alone: src [batch][H][W][C] bf16 -> V [16][batch * H/2 * W/2][C] bf16 = B^T d B per 4x4 patch (stride 2, zero padded);
 * dfh_gn_wino_input: the same over silu(GroupNorm(concat(src0, src1))) (ResnetBlock2D.norm1 / norm2 + nonlinearity in front of the conv),
 * one launch, the normalised tensor never goes to HBM; dfh_gn_wino_input_ok tells whether the (image, group) slab fits the kernel */
int dfh_wino_input(const void* src, void* V, int batch, int H, int W, int C, void* stream);
int dfh_gn_wino_input_ok(int c0, int c1, int groups, int H, int W);
int dfh_gn_wino_input(const void* src0, int c0, const void* src1, int c1, const float* gamma, const float* beta, float eps, int groups,
                      void* V, int batch, int H, int W, void* stream);
/* conv1 -> conv2 of a ResnetBlock2D: the tensor between the two convs is rebuilt from conv1's transform-domain planes Mprev [16][batch H W / 4][C]
 * (A^T m A + pbias + the image's row prowvec[b * prv_ld + prv_off + c], rounded to bf16 as the stored tensor would be) inside conv2's
 * GroupNorm + input transform: no output-transform launch, no round trip of that tensor */
int dfh_gn_wino_input_chain(const void* Mprev, const float* pbias, const float* prowvec, int prv_ld, int prv_off, int C, const float* gamma,
                            const float* beta, float eps, int groups, void* V, int batch, int H, int W, void* stream);
/* 1 when the library stores U of an N x C conv in 16-row x 64-column blocks ([16][N / 16][C / 64][16][64]: every 2-KB DRAM burst of the
 * weight stream is used whole); pass the same value as `blocked` / `u_blocked` */
int dfh_wino_blocked(int N, int C);
size_t dfh_conv3x3_wino_scratch_bytes(int batch, int H, int W, int C, int N);
int dfh_conv3x3_wino(const void* src, int batch, int H, int W, int C, const void* U, int u_blocked, int N, const float* bias, const float* rowvec,
                     int rv_ld, int rv_off, const void* resid, void* out, void* scratch, size_t scratch_bytes, const void* zero_page,
                     void* stream);
/* dfh_gemm over nbatch independent planes in ONE launch (grid.y): plane z reads d->a0 + z * a_bs, d->W + z * w_bs and writes
 * d->out + z * o_bs (strides in bf16 elements).  One plain K segment, bias-only epilogue, bf16 row-major output, no split-K. */
int dfh_gemm_batched(const dfh_gemm_desc* d, int nbatch, long a_bs, long w_bs, long o_bs, void* stream);
/* ---- fp8 (OCP e4m3fn) linears: BASELINE configs[4].  The reference has no fp8 path (fp16 autocast, run_inf4eval.sh:1); these
 * replace the same diffusers linears as dfh_gemm (BasicTransformerBlock attn1.to_q/k/v, attn2.to_q, ff.net.0.proj reached from
 * df.py:249-253,518-523) when the caller opts in.
 *   dfh_quantize_rows_fp8 : bf16 [R][K] (row stride ldx) -> e4m3 [R][K] + scale[R] = amax / 448 (weights: one scale per output channel)
 *   dfh_layernorm_fp8     : dfh_layernorm whose output is quantised per token: q [M][C] e4m3, scale [M]
 *   dfh_gemm_fp8          : out = epilogue(sa(m) * sW[n] * sum_k 2^(sx[k / 32][m] - 127) A[m][k] W[n][k]); K % 64 == 0; act 0 or 4
 *                           (GEGLU, packed rows).  Activation scales, any combination: sA (a float per group of sa_div rows: per token,
 *                           or per image), sa_mul (a constant), sx (E8M0 block scales per row and 32 contraction elements, the
 *                           hardware's scale operand; layout [K / 32][M] bytes).  out_mode 0 (bf16 [M][ld_out]), 1 (bf16 transposed per
 *                           batch of rows_per_b rows; amax != NULL: amax[b] = max(amax[b], max |out| of batch element b), the
 *                           caller zeroes it) or 4 (e4m3 [M][ld_out] + E8M0 block scales of the OUTPUT in out_sx [N_out / 32][M]:
 *                           the operand of the next fp8 GEMM, e.g. the GEGLU hidden tensor for ff.net.2).  Zero-initialise the descriptor.
 *   dfh_groupnorm_fp8     : GroupNorm WITHOUT its affine, as e4m3: q = e4m3(clamp((x - mean) * rstd * q_mul)) -- the operand of an fp8
 *                           proj_in whose weights carry gamma (and whose bias carries W . beta): a static scale, no statistics of q needed
 *   dfh_attention_fp8out  : dfh_attention whose output leaves as e4m3 scaled by 448 / v_amax[b] (|O| <= max |V| of the batch element)
 *   dfh_amax_slabs        : out[s][b] = max |x| over rows row0[s] .. + nrows[s] of batch element b of a bf16 [B][.][ld] tensor */
typedef struct dfh_gemm_fp8_desc {
  const void* A; int lda;                    /* e4m3 [M][lda]; lda 0 = K */
  const float* sA; int sa_div; float sa_mul;
  const void* sx;
  const void* W; const float* sW;            /* e4m3 [N][K], one scale per output channel */
  int M, N, K;
  const float* bias; const void* resid; int ld_res; int act;
  void* out; int ld_out; int out_mode; void* out_sx; int rows_per_b; float* amax;
  const void* zero_page;
} dfh_gemm_fp8_desc;
int dfh_quantize_rows_fp8(const void* x, int ldx, void* q, float* scale, int R, int K, void* stream);
int dfh_layernorm_fp8(const void* x, const float* gamma, const float* beta, void* q, float* scale, int M, int C, float eps, void* stream);
int dfh_gemm_fp8(const dfh_gemm_fp8_desc* d, void* stream);
/* Fused GEGLU feed-forward + proj_out of a transformer block at C = 320 (csrc/mlp_fused2.hip; replaces the ff.net.0 / ff.net.2 / proj_out
 * launches of diffusers BasicTransformerBlock.ff + Transformer2DModel.proj_out at the 64x64 level, reference call site
 * DiFashion/models/difashion.py:518-523):  out = [pout . ff2 | pout] . [GEGLU(LN3(x) . W1^T + b1) | x] + bias + resid.
 *   dfh_mlp_fused_pack : builds the layer's weight image (dfh_mlp_fused_image_bytes() bytes) from the LayerNorm-folded GEGLU projection
 *                        w1 ([8 C][C] bf16, rows packed 16 values | 16 gates; s1 / b1 its fold vectors, dfh_ln_fold) and w2p = [C][5 C]
 *   dfh_mlp_fused      : x / resid / out [M][320] bf16, M a multiple of 128; ln_stat = per-row statistics of x ([ln_parts][M][2]: mean
 *                        and centred sum of squares per column tile of ln_cnt columns, as dfh_gemm's row statistics) */
size_t dfh_mlp_fused_image_bytes(void);
/* form: 2 = the kernel (eight waves of 16 tokens on v_mfma_f32_16x16x32_bf16, two per SIMD).  1 = its first form (four waves of 32 tokens, one per
 * SIMD) exists in the probe library only (scripts/probes); an image packed for one form is only valid for that form */
int dfh_mlp_fused_pack(const void* w1, const float* s1, const float* b1, const void* w2p, void* img, int form, void* stream);
/* gstat (form 2, may be NULL): GroupNorm statistics of out for its consumer, [image][320 / gstat_cpg][gstat_hw / 128][2] = (sum, sum of squares) of
 * the bf16-rounded outputs per (image of gstat_hw tokens, group of gstat_cpg channels, 128-token chunk) */
int dfh_mlp_fused(const void* x, const void* resid, const void* img, const float* ln_stat, int ln_parts, int ln_cnt, float ln_eps,
                  const float* bias, void* out, int M, int form, float* gstat, int gstat_cpg, int gstat_hw, void* stream);
int dfh_groupnorm_fp8(const void* src, int batch, int HW, int C, int groups, float eps, float q_mul, void* q, float* partial, void* stream);
int dfh_attention_fp8out(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O8, int ldo, const float* v_amax,
                         int batch, int heads, int head_dim, int Nq, int Nk, float scale, void* stream);
int dfh_amax_slabs(const void* x, long bstride, int ld, int cols, const int* row0, const int* nrows, float* out, int nslab, int batch,
                   void* stream);
/* Self-attention with both products on the e4m3 MFMA (csrc/attention_fp8.hip): bf16 Q / K / V^T in HBM, quantised on the way into the
 * matrix pipe with STATIC per-channel factors (q * rq, k * rk with rq * rk = 1 / hs[h] per head, v * rv; [heads * head_dim] each, hs
 * [heads]); bf16 output.  head_dim 40 / 80 / 160, Nk a multiple of 64.  dfh_attn_scales derives the factors from the LayerNorm-folded
 * projection weights wf [3C][C] (rows q | k | v, bf16) and biases bf [3C]: bound = |row|_2 sqrt(C) + |bias| for a LayerNorm-ed input. */
int dfh_attention_fp8(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo, const float* rq,
                      const float* rk, const float* rv, const float* hs, int batch, int heads, int head_dim, int Nq, int Nk, float scale,
                      void* stream);
int dfh_attn_scales(const void* wf, const float* bf, int C, int heads, float* rq, float* rk, float* rv, float* hs, void* stream);
/* softmax(Q K^T * scale) V; Q [B][Nq][ldq], K [B][Nk][ldk], Vt [B][H*D][ldvt] (V transposed), O [B][Nq][ldo]; bf16 */
int dfh_attention(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                  int batch, int heads, int head_dim, int Nq, int Nk, float scale, void* stream);
/* Timesteps(flip_sin_to_cos=True, freq_shift=0): t [B] fp32 -> [B][dim] bf16 */
int dfh_timestep_embedding(const float* t, void* out, int batch, int dim, void* stream);
int dfh_nchw_to_nhwc_bf16(const void* x, int x_bf16, void* out, int batch, int C, int HW, void* stream);
int dfh_cast_f32_to_bf16(const float* x, void* y, size_t n, void* stream);

/* weight packing (fp32 master -> bf16 kernel layout) */
int dfh_pack_conv3x3(const float* w_oihw, void* out, int Cout, int Cin, int ldw, int col_off, void* stream);
int dfh_pack_matrix(const float* w, void* out, int N, int K, int ldw, int row_off, int col_off, int geglu, void* stream);
int dfh_pack_vector(const float* v, float* out, int N, int off, int geglu, int accumulate, void* stream);

/* ------------------------------------------------------------------ backward / training-step ops (train.py:691-716)
 * dfh_groupnorm with the (mean, rstd) pairs [B][G][2] kept for the backward pass */
int dfh_groupnorm_stats(const void* src0, int c0, const void* src1, int c1, int batch, int hw, int groups,
                        const float* gamma, const float* beta, float eps, int silu, void* out, float* partial,
                        float* stats_out, void* stream);
/* GroupNorm(+SiLU) backward: dx0/dx1 bf16 (acc != 0: add into the buffer), dgamma/dbeta fp32 += (atomics) */
int dfh_groupnorm_bwd(const void* src0, int c0, const void* src1, int c1, const void* dy, int batch, int hw, int groups,
                      const float* gamma, const float* beta, const float* stats, int silu, void* dx0, int acc0,
                      void* dx1, int acc1, float* dgamma, float* dbeta, float* partial, void* stream);
/* attention for training: forward that also returns the per-row log2-domain log-sum-exp [B][H][Nq], and the
 * flash-style backward (V ROW-major [B][Nk][ldv] here; delta = rowsum(dO * O) from dfh_attention_delta) */
int dfh_attention_lse(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                      int batch, int heads, int head_dim, int Nq, int Nk, float scale, float* lse, void* stream);
int dfh_attention_delta(const void* O, const void* dO, int ld, float* delta, int batch, int heads, int head_dim, int Nq, void* stream);
int dfh_attention_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* dO, int ldo,
                      const float* lse, const float* delta, void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv,
                      int batch, int heads, int head_dim, int Nq, int Nk, float scale, void* stream);
int dfh_layernorm_bwd(const void* x, const void* dy, const float* gamma, void* dx, int accumulate, float* dgamma,
                      float* dbeta, int M, int C, float eps, void* stream);
/* transposed / flipped weight packing so that data gradients reuse dfh_gemm:
 *   linear: Wt[t_row_off + k][t_col_off + n] = w[n][k];  conv3x3: W'[c][t_col_off + (8-t)*o_pad + o] = w[o][c][t] */
int dfh_pack_matrix_t(const float* w, void* out, int N, int K, int ldt, int t_row_off, int t_col_off, int geglu, void* stream);
int dfh_pack_conv3x3_t(const float* w, void* out, int Cout, int Cin, int ldt, int t_col_off, int o_pad, void* stream);
/* packed fp32 gradient -> master layout (+=): inverses of dfh_pack_matrix / dfh_pack_conv3x3 / dfh_pack_vector */
int dfh_unpack_matrix(const float* g, float* grad, int N, int K, int ldw, int row_off, int col_off, int geglu, void* stream);
int dfh_unpack_conv3x3(const float* g, float* grad, int Cout, int Cin, int ldw, int col_off, int cin_pad, void* stream);
int dfh_unpack_vector(const float* g, float* grad, int N, int off, int geglu, void* stream);
int dfh_pool2x2_sum(const void* in, void* out, int batch, int H, int W, int C, void* stream);   /* nearest-2x backward */
int dfh_add_bf16(void* dst, const void* src, size_t n, int accumulate, void* stream);
int dfh_geglu_fwd(const void* pre, void* y, size_t M, int N2, void* stream);
int dfh_geglu_bwd(const void* pre, const void* dy, void* dpre, size_t M, int N2, void* stream);
int dfh_act_fwd(const void* pre, void* y, size_t n, int kind, void* stream);      /* 1 silu 2 leaky_relu 3 tanh */
int dfh_act_bwd(const void* ref_bf16, const float* ref_f32, const void* dy_bf16, const float* dy_f32, void* dpre, size_t n,
                int kind, float scale, void* stream);
int dfh_nhwc_to_nchw_f32(const void* src, float* dst, int batch, int HW, int Cp, int C, float scale, int accumulate, void* stream);
int dfh_transpose_bf16(const void* in, void* out, int batch, int R, int C, int ld_in, int ld_out, size_t in_bstride,
                       size_t out_bstride, void* stream);
int dfh_mse_bwd(const float* pred, const float* target, const float* w, float* dpred, int rows, int L, float loss_scale,
                const float* scale_dev /* optional device scalar multiplied in (upstream d loss) */, void* stream);
int dfh_assemble_bwd(const float* dx, const uint8_t* mutual_real, float* dmutual, int rows, int CL, float eta, void* stream);
/* optimizer: torch.optim.AdamW step with the clip_grad_norm_ coefficient derived on device from *sumsq (may be NULL),
 * squared-norm accumulation, diffusers EMAModel update */
int dfh_sumsq(const float* g, size_t n, float* out, void* stream);
int dfh_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, const float* sumsq, float max_norm, void* stream);
/* dfh_adamw with the EMA update of the same elements folded in (one pass over the parameters instead of two) */
int dfh_adamw_ema(float* p, const float* g, float* m, float* v, float* shadow, size_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, const float* sumsq, float max_norm, float ema_decay, void* stream);
int dfh_ema(float* shadow, const float* p, size_t n, float decay, void* stream);
/* bf16 wire format of the data-parallel gradient exchange (replaces DDP's fp32 all-reduce behind accelerator.backward, train.py:611,699;
 * difashion_amd/dist.py exchange_bf16): fp32 range -> bf16 wire of n_pad elements (zero padding, n_pad % 8 == 0) | this rank's shard =
 * the `world` received contributions ([world][per] bf16) summed in fp32 in rank order, / world, rounded once | bf16 wire -> fp32 range */
int dfh_wire_pack(const float* g, void* wire, size_t n, size_t n_pad, void* stream);
int dfh_wire_shard_mean(const void* recv, void* shard, int world, size_t per, void* stream);
int dfh_wire_unpack(const void* wire, float* g, size_t n, void* stream);

/* ------------------------------------------------------------------ DiFashion glue (reference-owned arithmetic)
 * Sibling reduce feeding MutualEncoder: df.py:160-170 (training mean) / df.py:475-489 (sampling sum).
 *   out[j] = sum_k wtab[j][k] * (table[j][k] >= 0 ? gen[table] : given[-(table+1)])   (slot order)
 *   gen/given: fp32 rows of L = 4*H*W; out bf16 [rows][L] (MLP operand), out_f32 optional. */
int dfh_mutual_reduce(const float* gen, const float* given, const int32_t* table, const float* wtab,
                      void* out_bf16, float* out_f32, int rows, int olen, int L, void* stream);
/* Input assembly df.py:215-216 / :514-515 incl. CFG replica stacking: x [R*F][8][H][W] fp32 NCHW */
int dfh_assemble_input(const float* latents, const float* mutual, const float* hist, const float* null_latent,
                       const uint8_t* mutual_real, const uint8_t* hist_real, float* x, int R, int F, int CL,
                       float one_minus_eta, float eta, int per_row_flags, void* stream);
/* Guidance combine df.py:525-566 fused with scheduler.step df.py:569 (DDIMScheduler.step / PNDM transfer) */
typedef struct dfh_step_coef {
  int kind;            /* -1 combine only, 0 DDIM, 1 linear (x' = a*x - b*eps) */
  int vpred;
  float sqrt_a_t, sqrt_b_t, sqrt_a_prev, dir_coef, std_dev;
} dfh_step_coef;
int dfh_cfg_step(const float* eps_all, float* latents, float* eps_out, const float* noise, size_t n, int mode,
                 float cate_scale, float hist_scale, float mutual_scale, const dfh_step_coef* k, void* stream);
/* DDIMScheduler.add_noise / get_velocity df.py:158,:244 */
int dfh_noise_mix(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp,
                  float* noisy, float* velocity, int rows, int L, void* stream);
/* per-row MSE of df.py:256-264 (fp32) */
int dfh_mse_rows(const float* pred, const float* target, float* out, int rows, int L, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFASHION_HIP_H */
