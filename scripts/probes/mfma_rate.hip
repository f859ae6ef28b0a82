// Issue rate of v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_16x16x16_bf16 on gfx950 (cycles per instruction, one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc, int pattern = 1) {
  f32x4_t acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
  bf16x8_t a8 = {1, 1, 1, 1, 1, 1, 1, 1}, b8 = a8;
  if (pattern == 0) { a8 = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; b8 = a8; }
  if (pattern == 2) {           // pseudo-random bf16 bit patterns in [-2, 2), different per lane
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    u16x8_t ua, ub;
    for (int j = 0; j < 8; ++j) { h = h * 1664525u + 1013904223u; ua[j] = (unsigned short)((h >> 16) & 0xBFFF) | 0x3000; h = h * 1664525u + 1013904223u; ub[j] = (unsigned short)((h >> 16) & 0xBFFF) | 0x3000; }
    a8 = __builtin_bit_cast(bf16x8_t, ua); b8 = __builtin_bit_cast(bf16x8_t, ub);
  }
  s16x4_t a4 = {0x3f80, 0x3f80, 0x3f80, 0x3f80}, b4 = a4;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; long long* cyc; long long h;
  hipMalloc(&out, 4 * 256 * 256 * 4 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int pattern = 0; pattern < 3; ++pattern)
  for (int wps = 1; wps <= 4; wps += (wps == 1 ? 1 : 2)) {        // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, 10, cyc, pattern);
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, iters, cyc, pattern);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("16x16x32_bf16, operands %s, %d waves per SIMD x 8 independent accumulators: %.0f TFLOP/s over the chip\n",
           pattern == 0 ? "zeros " : (pattern == 1 ? "ones  " : "random"), wps, 5 * 256.0 * wps * 4 * iters * 8 * 16384.0 / ms / 1e9);
  }
  for (int kind = 0; kind < 2; ++kind) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, 10, cyc); else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, 10, cyc);
    hipEventRecord(e0);
    if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc); else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    const double flops = 256.0 * 4 * n * (kind == 0 ? 16384.0 : 8192.0);
    printf("%s: %.2f clock64 ticks per MFMA (per wave), %.3f ms, %.0f TFLOP/s over the chip\n", kind == 0 ? "16x16x32_bf16" : "16x16x16_bf16_1k",
           (double)h / n, ms, flops / ms / 1e9);
  }
  return 0;
}
