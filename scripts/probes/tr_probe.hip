// Hardware probe: exact lane/element mapping of ds_read_b64_tr_b16 on gfx950.
// LDS holds u16 element e at index e.  mode 0: lane l passes byte address base + l*8.
// mode 1: row-major tile T[m][n], 64 n per row; lane L of 16-lane group g points at &T[g*4 + L/4][(L%4)*4].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x, L = l & 15, g = l >> 4;
  const int elem = mode == 0 ? l * 4 : (g * 4 + L / 4) * 64 + (L % 4) * 4;
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)(lds + elem);
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
  if (l == 0) out[256] = 0xBEEF;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2 + 64);
  uint16_t h[257];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipError_t e2 = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d sync=%s sentinel=%x\n", mode, hipGetErrorString(e2), h[256]);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("\n"); }
  }
  return 0;
}
