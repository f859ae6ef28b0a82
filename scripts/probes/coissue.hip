// Do the matrix pipe and the VALU / transcendental unit of one SIMD run concurrently when they are fed by TWO waves?
// One 512-thread workgroup per CU = two waves per SIMD, guaranteed co-resident.  Waves 0-3 run role A, waves 4-7 role B;
// each role's loop is timed by its own wave (s_memtime) alone (other half idle: early exit) and together.
//   roles: 0 idle | 1 MFMA 32x32x16 only | 2 v_exp_f32 only | 3 v_fma_f32 only | 4 attention-like mix per "tile": 28 MFMA + 64 exp + 32 cvt_pk
//   hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int role>
__device__ __forceinline__ float role_loop(int iters) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  bf16x8_t a8, b8;
  for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(0.01f * (threadIdx.x & 15) + j); b8[j] = (__bf16)(0.5f - 0.01f * j); }
  f32x16_t acc[4];
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
  const float w0 = 0.999f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (role == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[k], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(acc[k]));
    } else if constexpr (role == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (role == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w0));
    } else if constexpr (role == 4) {      // per iteration: 7 MFMA + 16 exp + 8 cvt_pk (a quarter of a d = 40 attention tile), interleaved
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[k & 3], 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[2 * k]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[2 * k + 1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[(2 * k + 5) & 15]) : "v"(w0));
      }
      asm volatile("v_exp_f32 %0, %0" : "+v"(v[14]));
      asm volatile("v_exp_f32 %0, %0" : "+v"(v[15]));
      asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[3]) : "v"(w0));
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(acc[k]));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int k = 0; k < 4; ++k) s += acc[k][0];
  return s;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* stamps) {
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);                   // waves 0-3 / 4-7: one of each per SIMD
  const unsigned long long c0 = __builtin_readcyclecounter();
  float s = 0.f;
  if (half == 0) { if constexpr (RA != 0) s = role_loop<RA>(iters); }
  else { if constexpr (RB != 0) s = role_loop<RB>(iters); }
  const unsigned long long c1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) stamps[half] = c1 - c0;
}

template <int RA, int RB>
void run(float* out, unsigned long long* stamps) {
  const char* names[5] = {"idle", "MFMA 32x32x16 x4", "v_exp_f32 x16", "v_fma_f32 x16", "7 MFMA + 16 exp + 8 cvt"};
  const int iters = 2000;
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, out, 10, stamps);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, out, iters, stamps);
  hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
  printf("wave A: %-24s wave B: %-24s -> A %7.1f cycles / iteration, B %7.1f\n", names[RA], names[RB], (double)h[0] / iters,
         RB ? (double)h[1] / iters : 0.0);
}

int main() {
  float* out; unsigned long long* stamps;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&stamps, 16);
  run<1, 0>(out, stamps); run<2, 0>(out, stamps); run<3, 0>(out, stamps); run<4, 0>(out, stamps);
  run<1, 1>(out, stamps); run<2, 2>(out, stamps); run<3, 3>(out, stamps);
  run<1, 2>(out, stamps); run<1, 3>(out, stamps); run<2, 3>(out, stamps); run<4, 4>(out, stamps);
  return 0;
}
