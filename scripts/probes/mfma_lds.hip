// How much MFMA issue does an LDS fragment read cost?  A register-resident loop of 40 x v_mfma_f32_16x16x32_bf16 (the k-step of the
// 128 x 80 wave tile of gemm_wide.hip) with NR ds_read_b128 (or ds_read_b64) per iteration, reads issued in groups in front of
// rows of 5 MFMAs and waited for one row later (never on the critical path of a dependent MFMA).  1 and 2 waves per SIMD (160 accumulator registers: no room for more).
//   hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

// MODE 0: reads feed nothing (results discarded); 1: reads ARE the MFMA operands of the next row; 2: the reads are
// global_load_dwordx4 from a small L2-resident buffer (fragment-shaped: 16 rows x 64 B per instruction) instead of LDS reads;
// WIDE 1: b128, 0: b64
template <int NR, int MODE, int WIDE, int WPS>
__global__ __launch_bounds__(256 * WPS) void k(float* out, int iters, unsigned long long* stamps, const unsigned char* gbuf) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) ((unsigned*)lds)[i] = 0x3c003c00u + i;   // small bf16 values
  __syncthreads();
  // conflict-free fragment address: row = lane & 15, slot swizzled as in the kernels
  const unsigned row = lane & 15, fg = lane >> 4;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (threadIdx.x >> 6) * 1024 * 3 +
                        row * 64 + ((fg ^ ((row >> 1) & 3)) << 4);
  f32x4_t acc[40];
  for (int i = 0; i < 40; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t a[8], b[5];
  for (int i = 0; i < 8; ++i) a[i] = u32x4_t{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  for (int i = 0; i < 5; ++i) b[i] = u32x4_t{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  u32x4_t sink[4];
  for (int i = 0; i < 4; ++i) sink[i] = u32x4_t{0, 0, 0, 0};
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    constexpr int PER_ROW = (NR + 7) / 8;            // reads issued in front of each of the 8 rows
    int issued = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int q = 0; q < PER_ROW; ++q) {
        if (issued < NR) {
          if (MODE == 2) {
            const unsigned char* gp = gbuf + ((blockIdx.x & 7) * 65536 + (threadIdx.x >> 6) * 4096 + row * 128 + fg * 16 + (q & 1) * 2048);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sink[q & 3]) : "v"(gp));
          } else if (MODE == 1) {
            if (WIDE) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(r + 1) & 7]) : "v"(addr), "n"(1024 * (q & 1)));
            else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*(u32x2_t*)&a[(r + 1) & 7]) : "v"(addr), "n"(1024 * (q & 1)));
          } else {
            if (WIDE) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink[q & 3]) : "v"(addr), "n"(1024 * (q & 1)));
            else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*(u32x2_t*)&sink[q & 3]) : "v"(addr), "n"(1024 * (q & 1)));
          }
          ++issued;
        }
      }
#pragma unroll
      for (int j = 0; j < 5; ++j)
        acc[r * 5 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[j]), __builtin_bit_cast(bf16x8_t, a[r]),
                                                                acc[r * 5 + j], 0, 0, 0);
      // LDS reads: waited for after the row (100 MFMA cycles to return).  Global loads: waited for one whole ITERATION later
      // (vmcnt retires in order: allow this iteration's loads in flight), as a register-prefetched operand would be
      if (MODE == 2) { if (r == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NR > 63 ? 63 : NR) : "memory"); }
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(sink[q]));
    }
#pragma unroll
    for (int i = 0; i < 40; ++i) asm volatile("" : "+v"(acc[i]));
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 40; ++i) s += acc[i][0];
  s += (float)sink[0][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = c1 - c0;
}

static unsigned char* g_gbuf = nullptr;
template <int NR, int MODE, int WIDE, int WPS>
void run(float* out, unsigned long long* stamps) {
  const int iters = 2000, threads = 256 * WPS, waves_per_simd = WPS;
  hipLaunchKernelGGL((k<NR, MODE, WIDE, WPS>), dim3(256), dim3(threads), 0, 0, out, 10, stamps, g_gbuf);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NR, MODE, WIDE, WPS>), dim3(256), dim3(threads), 0, 0, out, iters, stamps, g_gbuf);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, stamps, 8, hipMemcpyDeviceToHost);
  const double cyc = (double)h / iters;
  printf("%2d x %s per 40 MFMA, %s, %d wave(s)/SIMD: %7.1f cycles per iteration and wave = %5.1f per MFMA (x waves: %6.1f per SIMD)\n", NR,
         MODE == 2 ? "global_load_x4" : (WIDE ? "ds_read_b128" : "ds_read_b64 "), MODE == 1 ? "reads feed the MFMAs" : "reads discarded    ", waves_per_simd, cyc, cyc / 40,
         cyc / waves_per_simd);
}

int main() {
  float* out; unsigned long long* stamps;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&stamps, 64);
  hipMalloc((void**)&g_gbuf, 8 * 65536); hipMemset(g_gbuf, 0, 8 * 65536);
#define ALL(W)                                                                                          \
  run<0, 0, 1, W>(out, stamps); run<5, 0, 1, W>(out, stamps); run<13, 0, 1, W>(out, stamps); run<13, 1, 1, W>(out, stamps); \
  run<26, 0, 1, W>(out, stamps); run<13, 0, 0, W>(out, stamps); run<26, 0, 0, W>(out, stamps);                  \
  run<5, 2, 1, W>(out, stamps); run<13, 2, 1, W>(out, stamps); run<26, 2, 1, W>(out, stamps);
  ALL(1) ALL(2)
  return 0;
}
