// Peak global->LDS staging rate (global_load_lds_dwordx4, L2-resident source): what the LDS-DMA path of one CU sustains.
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/glds_rate.hip -o gpurun_out/glds_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

__global__ __launch_bounds__(256) void glds_kernel(const unsigned char* __restrict__ src, size_t src_bytes, int iters, int row_bytes, int pitch, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every wave streams its own region; lanes fetch 16-byte chunks of rows of `row_bytes` (64 or 128: the k-step depth)
  const int lanes_per_row = row_bytes / 16;
  // pitch > row_bytes: the chunks of one instruction come from 64/lanes_per_row different rows (the GEMM A operand: one
  // pixel row of `pitch` bytes per tile row, of which the k-step takes row_bytes); pitch == row_bytes: fully contiguous
  const size_t limit = src_bytes - (1u << 17);
  size_t base = (((size_t)blockIdx.x * 4 + wave) * 1024 * (pitch / row_bytes)) % limit;
  const size_t lane_off = (size_t)(lane / lanes_per_row) * pitch + (lane % lanes_per_row) * 16;
  const size_t stride = ((size_t)gridDim.x * 4 * 1024 * (pitch / row_bytes)) % limit;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + lane_off),
                                       (__attribute__((address_space(3))) void*)(smem + (wave * 8 + u) * 1024), 16, 0, 0);
      base += stride;
      if (base >= limit) base -= limit;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = smem[0];
}

int main(int argc, char** argv) {
  const size_t src_bytes = (argc > 1 ? atoi(argv[1]) : 16) * (size_t)(1u << 20);          // 16 MB: L2-resident across 8 XCDs (4 MB each) mostly; MALL-resident for sure
  unsigned char* src; float* sink;
  hipMalloc(&src, src_bytes); hipMemset(src, 1, src_bytes);
  hipMalloc(&sink, 4096 * sizeof(float));
  hipFuncSetAttribute((const void*)glds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int cfg = 0; cfg < 4; ++cfg)
  for (int blocks_per_cu : {2}) {
    const int row_bytes = cfg == 0 ? 1024 : (cfg == 1 ? 128 : 64), pitch = cfg == 0 ? 1024 : (cfg == 3 ? 2560 : 640);
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(glds_kernel, dim3(blocks), dim3(256), 32 * 1024, 0, src, src_bytes, 10, row_bytes, pitch, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(glds_kernel, dim3(blocks), dim3(256), 32 * 1024, 0, src, src_bytes, iters, row_bytes, pitch, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * iters * 8 * 1024;
    printf("rows of %4d B at pitch %4d, %d blocks/CU x 4 waves: %.1f TB/s aggregate = %.1f GB/s per CU (%.1f B/clk at 2.4 GHz)\n", row_bytes, pitch, blocks_per_cu, bytes / ms / 1e9,
           bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
  }
  return 0;
}
