#pragma once
#include "dfh_common.h"

// guidance modes of fashion_generation (df.py:309-325); replica order as stacked by the reference
enum CfgMode {
  CFG_NONE = 0,         // 1 replica
  CFG_FULL = 1,         // 4: [all, cate+mutual, cate, uncond]
  CFG_CATE_HIST = 2,    // 3: [cate+hist, cate, uncond]
  CFG_CATE_MUTUAL = 3,  // 3: [cate+mutual, cate, uncond]
  CFG_CATE = 4,         // 2: [cate, uncond]
  CFG_HIST = 5,         // 2: [hist, uncond]   (also the hist+mutual, no-category case)
  CFG_MUTUAL = 6,       // 2: [mutual, uncond]
};

enum StepKind { STEP_NONE = -1, STEP_DDIM = 0, STEP_LINEAR = 1 };

struct StepCoef {
  int kind;          // StepKind
  int vpred;         // prediction_type == "v_prediction"
  float sqrt_a_t;    // DDIM: alpha_prod_t ** 0.5           | LINEAR: coefficient of x
  float sqrt_b_t;    // DDIM: (1 - alpha_prod_t) ** 0.5     | LINEAR: coefficient of eps
  float sqrt_a_prev; // alpha_prod_t_prev ** 0.5
  float dir_coef;    // (1 - alpha_prod_t_prev - std^2) ** 0.5
  float std_dev;     // eta * variance ** 0.5
};

namespace dfh {
int timestep_embed_launch(const float* t, bf16_t* out, int B, int dim, hipStream_t s);
int nchw_to_nhwc_launch(const void* x, int is_bf16, bf16_t* out, int B, int C, int HW, hipStream_t s);
int cast_f32_to_bf16_launch(const float* x, bf16_t* y, long n, hipStream_t s);
int mutual_reduce_launch(const float* gen, const float* given, const int* table, const float* wtab, bf16_t* out,
                         float* out_f32, int rows, int olen, int L, hipStream_t s);
int assemble_input_launch(const float* lat, const float* mutual, const float* hist, const float* null_latent,
                          const unsigned char* mutual_real, const unsigned char* hist_real, float* x, int R, int F, int CL,
                          float one_minus_eta, float eta, int per_row_flags, hipStream_t s);
int cfg_step_launch(const float* eps_all, float* lat, float* eps_out, const float* noise, long n, int mode, float sc,
                    float sh, float sm, StepCoef k, hipStream_t s);
int noise_mix_launch(const float* x0, const float* noise, const long* t, const float* sqrt_a, const float* sqrt_1ma,
                     float* noisy, float* velocity, int rows, int L, hipStream_t s);
int mse_rows_launch(const float* pred, const float* target, float* out, int rows, int L, hipStream_t s);
int pack_conv3x3_launch(const float* w, bf16_t* out, int Cout, int Cin, int ldw, int col_off, hipStream_t s, int cin_pad = 0);
int pack_matrix_launch(const float* w, bf16_t* out, int N, int K, int ldw, int row_off, int col_off, int geglu, hipStream_t s);
int pack_vector_launch(const float* v, float* out, int N, int off, int geglu, int accumulate, hipStream_t s);
}  // namespace dfh
