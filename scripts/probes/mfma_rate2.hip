// Sustained MFMA issue rate on gfx950, round 2: v_mfma_f32_32x32x16_bf16 beside v_mfma_f32_16x16x32_bf16, with the effective
// shader clock measured in the same launch (s_memtime ticks = shader cycles, s_memrealtime = 100 MHz constant clock).
// Register-resident loops: nothing but MFMAs (+ the loop branch).  Operands: zeros / random bf16 in [-2, 2).
//   hipcc --offload-arch=gfx950 -O3 mfma_rate2.hip -o mfma_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ inline void rnd_operands(bf16x8_t& a8, bf16x8_t& b8, int pattern) {
  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
  u16x8_t ua = {0, 0, 0, 0, 0, 0, 0, 0}, ub = ua;
  if (pattern == 1) {
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int j = 0; j < 8; ++j) {
      h = h * 1664525u + 1013904223u; ua[j] = (unsigned short)((h >> 16) & 0xBFFF) | 0x3000;
      h = h * 1664525u + 1013904223u; ub[j] = (unsigned short)((h >> 16) & 0xBFFF) | 0x3000;
    }
  }
  a8 = __builtin_bit_cast(bf16x8_t, ua); b8 = __builtin_bit_cast(bf16x8_t, ub);
}

// KIND 0: 16x16x32, NACC independent 4-register accumulators; KIND 1: 32x32x16, NACC independent 16-register accumulators
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* stamps, int pattern) {
  bf16x8_t a8, b8;
  rnd_operands(a8, b8, pattern);
  float s = 0;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  if (KIND == 0) {
    // hand-placed: hipcc allocates an array of f32x4 accumulators as ONE rotating register tuple (a[24:27] <- a[22:25] ...,
    // with ~48 v_accvgpr moves per iteration) -- the chains become dependent and the loop measures the compiler, not the pipe
    // (the flaw of the round-1 probe: 45 cycles per MFMA).  NACC chains on fixed AGPR tuples, zero-initialised.
    asm volatile(
        "v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n"
        "v_accvgpr_write_b32 a4, 0\n v_accvgpr_write_b32 a5, 0\n v_accvgpr_write_b32 a6, 0\n v_accvgpr_write_b32 a7, 0\n"
        "v_accvgpr_write_b32 a8, 0\n v_accvgpr_write_b32 a9, 0\n v_accvgpr_write_b32 a10, 0\n v_accvgpr_write_b32 a11, 0\n"
        "v_accvgpr_write_b32 a12, 0\n v_accvgpr_write_b32 a13, 0\n v_accvgpr_write_b32 a14, 0\n v_accvgpr_write_b32 a15, 0\n"
        "v_accvgpr_write_b32 a16, 0\n v_accvgpr_write_b32 a17, 0\n v_accvgpr_write_b32 a18, 0\n v_accvgpr_write_b32 a19, 0\n"
        "v_accvgpr_write_b32 a20, 0\n v_accvgpr_write_b32 a21, 0\n v_accvgpr_write_b32 a22, 0\n v_accvgpr_write_b32 a23, 0\n"
        "v_accvgpr_write_b32 a24, 0\n v_accvgpr_write_b32 a25, 0\n v_accvgpr_write_b32 a26, 0\n v_accvgpr_write_b32 a27, 0\n"
        "v_accvgpr_write_b32 a28, 0\n v_accvgpr_write_b32 a29, 0\n v_accvgpr_write_b32 a30, 0\n v_accvgpr_write_b32 a31, 0\n s_nop 4\n"
        ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20",
            "a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
    bf16x8_t b2 = b8; if (pattern) b2[3] = (__bf16)0.75f;
    for (int it = 0; it < iters; ++it) {
      if (NACC == 8)
        asm volatile(
            "v_mfma_f32_16x16x32_bf16 a[0:3], %0, %1, a[0:3]\n v_mfma_f32_16x16x32_bf16 a[4:7], %0, %2, a[4:7]\n"
            "v_mfma_f32_16x16x32_bf16 a[8:11], %0, %1, a[8:11]\n v_mfma_f32_16x16x32_bf16 a[12:15], %0, %2, a[12:15]\n"
            "v_mfma_f32_16x16x32_bf16 a[16:19], %0, %1, a[16:19]\n v_mfma_f32_16x16x32_bf16 a[20:23], %0, %2, a[20:23]\n"
            "v_mfma_f32_16x16x32_bf16 a[24:27], %0, %1, a[24:27]\n v_mfma_f32_16x16x32_bf16 a[28:31], %0, %2, a[28:31]\n"
            :: "v"(a8), "v"(b8), "v"(b2)
            : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20",
              "a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
      else
        asm volatile(
            "v_mfma_f32_16x16x32_bf16 a[0:3], %0, %1, a[0:3]\n v_mfma_f32_16x16x32_bf16 a[4:7], %0, %2, a[4:7]\n"
            "v_mfma_f32_16x16x32_bf16 a[8:11], %0, %1, a[8:11]\n v_mfma_f32_16x16x32_bf16 a[12:15], %0, %2, a[12:15]\n"
            :: "v"(a8), "v"(b8), "v"(b2)
            : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
    }
    float r;
    asm volatile("s_nop 15\n s_nop 15\n v_accvgpr_read_b32 %0, a0" : "=v"(r));
    s += r;
  } else {
    f32x16_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8_t bv[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { bv[i] = b8; if (pattern) bv[i][i & 7] = (__bf16)(0.5f + 0.125f * i); asm volatile("" : "+v"(bv[i])); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bv[i], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}

template <int KIND, int NACC>
void run(const char* name, float* out, unsigned long long* stamps) {
  const int iters = 4000;
  const double flops_per = KIND == 0 ? 16384.0 : 32768.0;
  for (int pattern = 0; pattern < 2; ++pattern)
    for (int wps = 1; wps <= 4; wps *= 2) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL((k<KIND, NACC>), dim3(256 * wps), dim3(256), 0, 0, out, 10, stamps, pattern);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<KIND, NACC>), dim3(256 * wps), dim3(256), 0, 0, out, iters, stamps, pattern);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
      const double n = (double)iters * NACC;
      const double ghz = (double)h[0] / ((double)h[1] * 10.0);   // shader cycles per ns (s_memrealtime: 100 MHz)
      printf("%-14s x%d acc, %s, %d wave(s)/SIMD: %7.0f TFLOP/s chip | %.1f shader cycles per MFMA per wave | effective clock %.2f GHz\n",
             name, NACC, pattern ? "random" : "zeros ", wps, 5 * 256.0 * wps * 4 * n * flops_per / ms / 1e9, (double)h[0] / n, ghz);
    }
}

int main() {
  float* out; unsigned long long* stamps;
  hipMalloc(&out, 4 * 256 * 256 * 4 * 4); hipMalloc(&stamps, 16);
  run<0, 8>("16x16x32_bf16", out, stamps);
  run<0, 4>("16x16x32_bf16", out, stamps);
  run<1, 4>("32x32x16_bf16", out, stamps);
  run<1, 2>("32x32x16_bf16", out, stamps);
  return 0;
}
