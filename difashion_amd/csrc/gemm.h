// Implicit-GEMM descriptor shared by gemm.hip (device) and the host orchestration.
//   out[M][N] = epilogue( sum over K-segments of A_seg[M][len] . W[N][seg_off : seg_off+len]^T )
// K is a concatenation of segments; each is padded (virtually, through a zero page) to a
// multiple of 64 so that a 64-deep k-step never straddles two segments:
//   * ntaps == 9 : the nine taps of a 3x3 / pad-1 convolution over ``conv_src`` (NHWC, conv_c
//                  channels, optional stride 2, optional fused nearest-2x upsample of the input)
//   * then up to two "plain" segments: row m of p_src[i] ([M][p_c[i]]) -- a linear / 1x1 conv
//                  operand, the second half of a channel concat, or a resnet's 1x1 shortcut
//                  accumulated into its conv2.
#pragma once
#include <stdint.h>
#include "dfh_common.h"

enum GemmAct { ACT_NONE = 0, ACT_SILU = 1, ACT_LEAKY = 2, ACT_TANH = 3, ACT_GEGLU = 4 };
enum GemmOut {
  OUT_BF16 = 0,    // bf16 [M][ld_out]
  OUT_BF16_T = 1,  // bf16 transposed per batch: out[b][n][ld_out], m = b*rows_per_b + mm (attention V^T)
  OUT_F32 = 2,     // fp32 [M][ld_out]
  OUT_F32_T = 3,   // fp32 transposed per batch (conv_out -> NCHW noise prediction)
  OUT_FP8_MX = 4,  // gemm_fp8.hip only: e4m3 [M][ld_out bytes] + one E8M0 block scale per row and 32 output columns (out_sx [N / 32][M])
};

struct GemmArgs {
  // --- K segments
  const bf16_t* conv_src; int conv_c; int ntaps;
  int Hin, Win, Hout, Wout, stride, ups;
  int pad0;        // 1: conv padding 0 with the input zero-extended on the right / bottom (VAE Downsample2D: F.pad(0,1,0,1) + stride 2)
  const bf16_t* p_src[2]; int p_c[2]; int nplain;
  const bf16_t* W; int ldw;
  const bf16_t* zero;  // >= 16 bytes of zeros in device memory
  int M, N;
  int ksteps;      // total 64-deep k-steps over all segments
  int ksplit;      // grid.z; each z handles a contiguous range of k-steps
  int n_major;     // tile ids enumerate column tiles slowest (set by the launcher when the weights outweigh the pixels)
  int tm_xm, tm_gm;  // XCD grid rows / row-tile group of the tile order (dfh_common.h tile_coords); 0 = legacy order
  // --- epilogue:  v = acc + bias[n] + rowvec[m / rows_per_b][rv_off + n];  v = act(v);  v += resid[m][n]
  const float* bias;
  const float* rowvec; int rv_ld; int rv_off; int rows_per_b;
  const bf16_t* resid; int ld_res;
  int act;
  void* out; int ld_out; int out_mode;
  // optional SECOND destination: output columns [n_split, N) leave as OUT_BF16_T into out2 ([batch][N - n_split][ld_out2], batches of
  // rows_per_b rows) while columns [0, n_split) follow out / out_mode -- attention's q | k and V^T from ONE launch over the shared
  // LayerNorm-ed (or folded-LayerNorm) rows.  n_split must be a multiple of the column tile; single-pass gemm_bf16_kernel launches only.
  void* out2; int ld_out2; int n_split;
  // GEGLU launches of the TRAINING walk: the bias-added pre-activations [M][N] (packed value / gate layout, bf16) as a second output of
  // the epilogue that gates them -- the backward needs them, and a separate gating pass would read them back (gemm_wide.hip, 256 x 128 tile)
  void* pre_out; int ld_pre;
  // BATCHED launch (grid.y): batch z reads plain segment 0 at p_src[0] + z * a_bs, the weights at W + z * w_bs and writes out + z * o_bs
  // (elements).  The sixteen transform-domain GEMMs of a Winograd F(2x2, 3x3) convolution (winograd.hip) in ONE launch: 16 x the tiles of
  // one, so the deep levels fill the chip without split-K.  Single plain segment, bf16 row-major output, gemm_bf16_kernel tiles only.
  int nbatch; long a_bs, w_bs, o_bs;
  // per-IMAGE weights: rows [i * rows_per_b, (i + 1) * rows_per_b) multiply W + i * w_img_bs (elements) -- a GroupNorm folded into the 1x1
  // projection that consumes it (norm.hip gn_fold_kernel: W_i = W . gamma . rstd_i per image, the mean / beta terms as the per-image row
  // vector `rowvec`).  gemm_bf16_kernel tiles only, no split-K, rows_per_b a multiple of the row tile.
  long w_img_bs;
  // PHASE-DECOMPOSED nearest-2x upsample + 3x3 conv (the up-block upsamplers): output pixel (2y + py, 2x + px) sees only a 2 x 2
  // neighbourhood of the SOURCE image -- rows {y - 1 + py, y + py}, columns likewise -- each with the SUM of the 3x3 taps that land on
  // it, so the conv is four independent 2x2 convs (4/9 of the multiply-adds of the conv over the upsampled image).  phase2x = 1:
  // nbatch = 4 planes (grid.y = py * 2 + px), ntaps = 4 (tap s = (s >> 1, s & 1) reads the 3x3-tap position (s >> 1) + py,
  // (s & 1) + px of the source), W plane z = the summed weights [N][4 * conv_c] (lnfold.hip ups_phase_fold), Hin = Win = Hout = Wout =
  // source size, M = rows of ONE phase; row m = (b, y, x) is stored at output pixel (b, 2y + py, 2x + px) of the 2H x 2W image.
  // phase2x = 2: the DATA GRADIENT of that conv (training): plane z convolves its own image of output-gradient pixels (conv_src + z * a_bs,
  // the output gradient gathered phase-major: bwd_elementwise.hip phase_gather) with W plane z = the TRANSPOSED phase weights [Cin][4 * Cout],
  // tap s at the mirrored position (2 - py - (s >> 1), 2 - px - (s & 1)); plane z writes plain rows at out + z * o_bs (summed afterwards).
  int phase2x;
  // profile accounting override (0 = derive from the shape): a launch that is one stage of a convolution (the batched transform-domain
  // GEMM of winograd.hip) reports the REFERENCE algorithm's multiply-adds (the direct conv) under the conv3x3 class
  double prof_flops;
  // W is stored in 16-row x 64-column blocks, [N / 16][ldw / 64][16][64] (2 KB each): a k-step's W tile is then BN / 16 contiguous 2-KB
  // chunks instead of BN separate 128-byte pieces of rows ldw * 2 bytes apart -- for weight streams that come from HBM (the deep levels:
  // few pixel rows share a weight tile) every DRAM page opened is used whole.  Derived weights only (Winograd U: winograd.hip); needs the
  // LEAN k-loop (plain segments of whole 64-channel slices), N % 16 == 0, ldw % 64 == 0.
  int w_blocked;
  float* partial;  // [ksplit][M][N] fp32 when ksplit > 1
  // --- optional GroupNorm statistics of the OUTPUT, written by the 256-row epilogue (gemm_wide_epilogue.h) when the launcher finds
  //     the launch eligible: per (image, group, row tile) the sum and the sum of squares of the bf16-rounded outputs, in the layout
  //     gn_apply_kernel reads ([image][group][chunk][2], chunk = row tile inside the image), so the consumer skips gn_stats_kernel
  float* gstat; int gstat_cpg, gstat_hw;   // channels per group of the consuming GroupNorm; pixels per image
  // --- LayerNorm folded into the GEMMs around it (transformer blocks, inference walk).  LN(x) . W^T = rstd * (x . W'^T - mean * s) + b'
  //     with W' = W * gamma (per input channel), s[n] = sum_k W'[n][k], b'[n] = bias[n] + sum_k W[n][k] * beta[k]: the consumer runs on
  //     the RAW rows x with W' as its weights, b' as its bias and fixes each output row up in the epilogue; the per-row statistics
  //     come from the epilogue of the GEMM that produced x.
  //   producer: rowstat != null -> per output row and column tile the mean and the centred sum of squares of the bf16-rounded outputs,
  //             layout [N / BN][M][2] (one 8-byte record per row and column tile; equal counts BN, combined exactly by the consumer)
  //   consumer: ln_stat = the producer's rowstat, ln_parts = its column tiles, ln_cnt = its BN, ln_s = s, bias = b'
  float* rowstat;
  const float* ln_stat; int ln_parts, ln_cnt; float ln_eps; const float* ln_s;
};

// fp8 (OCP e4m3fn) linear on v_mfma_scale_f32_32x32x64_f8f6f4 (gemm_fp8.hip):
//   out[m][n] = epilogue(sa(m) * sW[n] * sum_k 2^(sx[k / 32][m] - 127) * A8[m][k] * W8[n][k])
// Activation scaling, any combination: a float per row or per group of sa_div rows (sA: the per-token scale of a LayerNorm output, or one
// value per image), a constant (sa_mul) and E8M0 block scales per row and 32 contraction elements (sx: what the hardware's scale operand
// takes; written by the producers that cannot see a whole row -- the GEGLU epilogue, a column tile of another fp8 GEMM).
struct Fp8GemmArgs {
  const uint8_t* A; int lda;            // [M][lda] e4m3 activations (lda bytes per row; 0 = K)
  const float* sA; int sa_div; float sa_mul;   // sa(m) = (sA ? sA[m / sa_div] : 1) * sa_mul   (sa_div 0 = 1, sa_mul 0 = 1)
  const uint8_t* sx;                    // optional E8M0 block scales [K / 32][M]; null = 2^0
  const uint8_t* W; const float* sW;   // [N][K] e4m3 weights, one scale per row (output channel)
  int M, N, K;                          // K a multiple of 64
  const float* bias;                    // [N] or null
  const bf16_t* resid; int ld_res;      // optional residual (row-major outputs)
  int act;                              // ACT_NONE / ACT_GEGLU (packed rows interleaved in 16-row value / gate blocks)
  void* out; int ld_out; int out_mode;  // OUT_BF16 / OUT_BF16_T / OUT_FP8_MX (ld_out in elements of the output type)
  uint8_t* out_sx;                      // OUT_FP8_MX: E8M0 scales of the output, [N_out / 32][M]
  int rows_per_b;                       // OUT_BF16_T: rows per batch element
  float* amax;                          // OUT_BF16_T, optional: amax[b] = max(amax[b], max |out| of batch element b) (atomic; caller zeroes)
  const uint8_t* zero;                  // >= 16 bytes of zeros in device memory
};

#ifdef __HIPCC__
// (mean, rstd) of row m of a folded-LayerNorm consumer: the producer's per-column-tile records (mean_t, M2_t over ln_cnt columns each)
// combined exactly (equal counts: mean = average of the means, M2 = sum M2_t + cnt * sum (mean_t - mean)^2), fixed order
DFH_DEVICE float2 ln_row_stats(const GemmArgs& a, int m) {
  if (m >= a.M) return float2{0.f, 1.f};
  // one pass, all loads in flight together (kernels call this at ENTRY, so that the round trip hides behind the pipeline prologue
  // instead of standing between the last k-step and the epilogue of every tile: +3..15 us per launch when it did)
  float sm = 0.f, sq = 0.f, s2 = 0.f;
  for (int t = 0; t < a.ln_parts; ++t) {
    const float2 r = *(const float2*)(a.ln_stat + ((long)t * a.M + m) * 2);
    sm += r.x; sq = fmaf(r.x, r.x, sq); s2 += r.y;
  }
  const float inv = 1.0f / (float)a.ln_parts;
  const float mean = sm * inv;
  const float m2 = s2 + (float)a.ln_cnt * fmaxf(sq - sm * mean, 0.f);      // sum_t (mean_t - mean)^2 = sum mean_t^2 - (sum mean_t)^2 / parts
  const float var = m2 * inv / (float)a.ln_cnt;
  return float2{mean, rsqrtf(var + a.ln_eps)};
}
#endif

namespace dfh {
int gemm_fp8_launch(Fp8GemmArgs a, hipStream_t stream);
// out[sl][b] = max |x| over rows row0[sl] .. row0[sl] + nrows[sl] - 1 (cols columns each) of batch element b of a bf16 [B][.][ld] tensor
int amax_slabs_launch(const bf16_t* x, long bstride, int ld, int cols, const int* row0, const int* nrows, float* out, int nslab, int B,
                      hipStream_t stream);
// bf16 [R][K] (row stride ldx) -> e4m3 [R][K] + one scale per row (amax / 448)
int quant_rows_fp8_launch(const bf16_t* x, int ldx, uint8_t* q, float* scale, int R, int K, hipStream_t stream);
// LayerNorm whose output is quantised per token: q [M][C] e4m3, scale [M]
int layernorm_fp8_launch(const bf16_t* x, const float* gamma, const float* beta, uint8_t* q, float* scale, int M, int C, float eps,
                         hipStream_t stream);
// Picks a tile shape + split-K factor, launches, and (if split) launches the reduce.  ``partial``
// must hold gemm_partial_floats(...) floats when the heuristic splits.
int gemm_launch(GemmArgs a, hipStream_t stream, int force_tile = 0, int force_split = 0, int force_order = -1,
                int* gstat_rows = nullptr,        // *gstat_rows: pixel rows per statistics chunk a.gstat was filled with (256 / 128), 0 = not filled
                int* rowstat_bn = nullptr);       // *rowstat_bn: column tile of the a.rowstat records written (0 = none: kernel / shape cannot)
// lnfold.hip: W' = bf16(W * gamma) [N][K], s[n] = sum_k W'[n][k], b[n] = bias[n] (or 0) + sum_k W[n][k] * beta[k]
int ln_fold_launch(const bf16_t* W, int ldw, const float* gamma, const float* beta, const float* bias, bf16_t* WF, float* s, float* b,
                   int N, int K, hipStream_t stream);
// Winograd F(2x2, 3x3) transforms (winograd.hip): U = G g G^T per weight pack, V = B^T d B of the conv's input, out = A^T m A + epilogue
int wino_weight_launch(const bf16_t* W, int ldw, bf16_t* U, int N, int C, int blocked, hipStream_t stream);
// does the batched transform-domain GEMM of an N x C Winograd conv take U in the blocked layout (GemmArgs::w_blocked)?
bool wino_blocked(int N, int C);
int wino_input_launch(const bf16_t* g, bf16_t* V, int B, int H, int W, int C, hipStream_t stream);
// GroupNorm(+SiLU) fused with the input transform: V = B^T silu(GN(concat(src0, src1))) B (gn_wino_ok: is the shape supported?)
bool gn_wino_ok(int C0, int C1, int G, int H, int W);
// Mprev != null (conv1 -> conv2 of a resnet): the normalised tensor is the output transform of the previous conv's transform-domain planes
// Mprev [16][B H W / 4][C0] (+ pbias + the image's row of prowvec), rebuilt inside the kernel: src0 is not read
int gn_wino_input_launch(const bf16_t* src0, int C0, const bf16_t* src1, int C1, const float* gamma, const float* beta, float eps, int G,
                         bf16_t* V, int B, int H, int W, hipStream_t stream, const bf16_t* Mprev = nullptr, const float* pbias = nullptr,
                         const float* prowvec = nullptr, int prv_ld = 0, int prv_off = 0);
int wino_output_launch(const bf16_t* Mb, bf16_t* out, const float* bias, const float* rowvec, int rv_ld, int rv_off, const bf16_t* resid,
                       int B, int H, int W, int N, hipStream_t stream);
// tile id (gemm_launch force_tile) for the batched transform-domain GEMM: the 256-row ring when a plane has at most 256 rows (its
// weights are then read by ONE row tile), else 0 = the batched default (eight-wave 128 x 160)
int wino_gemm_tile(const GemmArgs& a);
// summed phase weights [4][N][4 * C] of a nearest-2x upsample + 3x3 conv from its packed [N][9 * C] matrix (GemmArgs::phase2x)
int ups_phase_fold_launch(const bf16_t* W, int ldw, bf16_t* WP, int N, int C, hipStream_t stream);
int matvec_bias_launch(const bf16_t* W, int ldw, const float* v, const float* b_add, float* b_out, int N, int K, hipStream_t stream);
// can a launch carry GemmArgs::out2 (its heuristic tile divides n_split, no split-K)?
bool gemm_out2_ok(GemmArgs a);
// can a launch of this shape consume folded-LayerNorm statistics (single pass, an epilogue that implements the fix-up)?
bool gemm_ln_consumer_ok(GemmArgs a);
int gemm_pick_split(const GemmArgs& a, int* tile_out);
size_t gemm_partial_floats(const GemmArgs& a);
int gemm_count_ksteps(const GemmArgs& a);
// 256 x 160 wide-tile variant (gemm_wide.hip): higher arithmetic intensity against the LDS staging path
bool gemm_wide_eligible(const GemmArgs& a);
int gemm_wide_pick(const GemmArgs& a);                       // 0 none, 1 = 256 x 160, 2 = 128 x 160
int gemm_wide_launch(GemmArgs a, hipStream_t stream, int variant = 1);
#ifdef DFH_PROBES   // scripts/probes/kernels: experiments that lost their A/B, built only into the probe library
// wave-specialised 256 x {160,128} kernel (gemm_ws.hip): loader waves + matrix waves, one workgroup per CU
bool gemm_halo_eligible(const GemmArgs& a);                  // gemm_halo.hip: 3x3 conv, pixels staged once per channel slice
int gemm_halo_launch(GemmArgs a, hipStream_t stream);
int gemm_ws_pick(const GemmArgs& a, int min_tiles);           // 0 = not eligible, else the column tile (160 / 128)
int gemm_ws_launch(GemmArgs a, hipStream_t stream, int bn);
// persistent GEGLU projection kernel (scripts/probes/kernels/gemm_geglu.hip): one workgroup per CU walks 256 x 256 tiles and fetches the
// next tile's first stage during the current tile's epilogue; bit-identical to gemm_bf16_kernel<256,256,2,4,2,LEAN,WEPI>, measured equal
// inside the step (profiles/r03/geglu_phases.txt)
bool gemm_geglu_rows_ok(const GemmArgs& a);
int gemm_geglu_rows_launch(const GemmArgs& a, hipStream_t stream);
// persistent short-K token linear (scripts/probes/kernels/gemm_persist.hip, round 6): one workgroup per CU walks its 128 x 160 tiles, the LDS ring
// runs across tiles; bit-identical to gemm_bf16_kernel<128,160,4,2,*,LEAN>, measured 1.2-1.5 x SLOWER (profiles/r06/persistent_lean_gemm.md)
bool gemm_persist_ok(const GemmArgs& a);
int gemm_persist_launch(const GemmArgs& a, hipStream_t stream, int max_wgs = 0);
#endif
}  // namespace dfh
