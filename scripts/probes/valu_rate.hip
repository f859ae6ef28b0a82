// Issue / throughput cost of the softmax VALU instructions on gfx950: v_exp_f32 (transcendental), v_fma_f32, v_max3_f32,
// v_cvt_pk_bf16_f32, v_pk_fma_f32 -- alone, mixed with each other, and beside v_mfma_f32_32x32x16_bf16 -- in shader cycles per
// wave-instruction at 1 / 2 waves per SIMD.  What bounds attention at head dim 40: one exp per score.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define REP8(x) x x x x x x x x
// MODE 0: 16 exp | 1: 16 fma | 2: 16 max3 | 3: 16 cvt_pk | 4: 8 exp + 8 fma interleaved | 5: 8 exp + 24 fma | 6: 16 pk_fma
// 7: 2 MFMA 32x32x16 + 16 exp | 8: 2 MFMA + 16 fma | 9: 2 MFMA alone | 10: 2 MFMA + 8 exp + 16 fma | 11: 16 v_ldexp | 12: 16 v_fract
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* stamps) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  float w[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) w[i] = 0.5f + 0.001f * i;
  bf16x8_t a8, b8;
  for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(0.01f * (threadIdx.x & 15) + j); b8[j] = (__bf16)(0.5f - 0.01f * j); }
  f32x16_t acc0, acc1;
  for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 7 || MODE == 8 || MODE == 9 || MODE == 10) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc1, 0, 0, 0);
    }
    if (MODE == 0 || MODE == 7) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if (MODE == 1 || MODE == 8) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[0]));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[0]), "v"(w[1]));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[0]));
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[8 + i]) : "v"(w[0])); }
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[3 * i]) : "v"(v[15]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[3 * i + 1]) : "v"(v[15]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[3 * i + 2]) : "v"(v[15]));
      }
    } else if (MODE == 6) {
      typedef __attribute__((ext_vector_type(2))) float f2;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 x = {v[i], v[i + 1]}; const f2 y = {w[0], w[1]};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
        v[i] = x[0]; v[i + 1] = x[1];
      }
    } else if (MODE == 10) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[2 * i]) : "v"(v[15]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[2 * i + 1]) : "v"(v[15]));
      }
    } else if (MODE == 11) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1));
    } else if (MODE == 12) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fract_f32 %0, %0" : "+v"(v[i]));
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = acc0[0] + acc1[3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 24; ++i) s += w[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = c1 - c0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* stamps) {
  const int iters = 2000;
  for (int wps = 1; wps <= 2; ++wps) {
    hipLaunchKernelGGL((k<MODE>), dim3(256 * wps), dim3(256), 0, 0, out, 10, stamps);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE>), dim3(256 * wps), dim3(256), 0, 0, out, iters, stamps);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, stamps, 8, hipMemcpyDeviceToHost);
    printf("%-44s %d wave(s)/SIMD: %7.1f shader cycles per loop iteration (one wave)\n", name, wps, (double)h / iters);
  }
}

int main() {
  float* out; unsigned long long* stamps;
  hipMalloc(&out, 2 * 256 * 256 * 4); hipMalloc(&stamps, 16);
  run<0>("16 v_exp_f32", out, stamps);
  run<1>("16 v_fma_f32", out, stamps);
  run<2>("16 v_max3_f32", out, stamps);
  run<3>("16 v_cvt_pk_bf16_f32", out, stamps);
  run<11>("16 v_ldexp_f32", out, stamps);
  run<12>("16 v_fract_f32", out, stamps);
  run<6>("16 v_pk_fma_f32 (32 fma)", out, stamps);
  run<4>("8 v_exp + 8 v_fma interleaved", out, stamps);
  run<5>("8 v_exp + 24 v_fma interleaved", out, stamps);
  run<9>("2 mfma_32x32x16 alone", out, stamps);
  run<7>("2 mfma_32x32x16 + 16 v_exp", out, stamps);
  run<8>("2 mfma_32x32x16 + 16 v_fma", out, stamps);
  run<10>("2 mfma_32x32x16 + 8 v_exp + 16 v_fma", out, stamps);
  return 0;
}
