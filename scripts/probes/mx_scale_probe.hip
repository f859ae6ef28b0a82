// Two questions about gfx950, answered on the device:
//  (1) which elements does the scale operand of v_mfma_scale_f32_32x32x64_f8f6f4 scale?  A = all ones; B = 1.0 in the first 32 contraction
//      elements of every column and 2.0 in the last 32 (so the two k-halves are told apart); ONE lane's B (or A) scale is 2^1, the rest 2^0,
//      upper bytes of the scale register zero (what ds_read_u8 leaves there).
//  (2) where does an EXEC-masked global_load_lds_dword put its data?  lanes 0..15 load, the LDS region is dumped.
// hipcc --offload-arch=gfx950 -O2 mx_scale_probe.hip -o mx_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ void probe(const int* sa, const int* sb, float* out) {
  const int lane = threadIdx.x;
  i32x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = lane < 32 ? 0x38383838 : 0x40404040; }
  f32x16_t c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
__global__ void dma_probe(const unsigned* src, unsigned* dump, int nact) {
  __shared__ unsigned lds[128];
  const int lane = threadIdx.x;
  lds[lane] = 0xdeadbeefu; lds[lane + 64] = 0xdeadbeefu;
  __syncthreads();
  if (lane < nact)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane),
                                     (__attribute__((address_space(3))) void*)(lds + 16), 4, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  dump[lane] = lds[lane]; dump[lane + 64] = lds[lane + 64];
}
int main() {
  int *sa, *sb; float* out;
  hipMalloc(&sa, 256); hipMalloc(&sb, 256); hipMalloc(&out, 64 * 16 * 4);
  int ha[64], hb[64]; float ho[1024];
  for (int which = 0; which < 2; ++which)
    for (int L : {0, 5, 32, 37}) {
      for (int i = 0; i < 64; ++i) { ha[i] = 0x7f; hb[i] = 0x7f; }
      (which ? ha : hb)[L] = 0x80;
      hipMemcpy(sa, ha, 256, hipMemcpyHostToDevice); hipMemcpy(sb, hb, 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, sa, sb, out);
      hipMemcpy(ho, out, 4096, hipMemcpyDeviceToHost);
      printf("scale_%c of lane %2d = 2^1 (baseline 96 everywhere): ", which ? 'a' : 'b', L);
      int n = 0;
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (ho[l * 16 + r] != 96.f) {
        if (n < 4) printf("[row %d col %d] = %g  ", (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31, ho[l * 16 + r]);
        ++n;
      }
      printf(" (%d outputs changed)\n", n);
    }
  unsigned *src, *dump, hs[64], hd[128];
  hipMalloc(&src, 256); hipMalloc(&dump, 512);
  for (int i = 0; i < 64; ++i) hs[i] = 0x1000 + i;
  hipMemcpy(src, hs, 256, hipMemcpyHostToDevice);
  for (int nact : {64, 16}) {
    hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 0, 0, src, dump, nact);
    hipMemcpy(hd, dump, 512, hipMemcpyDeviceToHost);
    printf("global_load_lds_dword, lanes 0..%d active, LDS base = word 16:", nact - 1);
    for (int i = 0; i < 128; ++i) if (hd[i] != 0xdeadbeefu) printf(" [%d]=%x", i, hd[i]);
    printf("\n");
  }
  return 0;
}
