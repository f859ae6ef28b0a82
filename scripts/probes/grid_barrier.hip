// Cost of a grid-wide barrier on this part: a persistent grid of G co-resident workgroups x 256 threads runs N rounds of
// (optional: touch `bytes_per_round` of memory) + barrier (one agent-scope atomic per workgroup on a shared counter, spin on it with
// s_sleep, workgroup barrier).  Prices "one cooperative kernel per U-Net block" for the 8x8 / 16x16 levels: such a kernel needs a
// grid barrier wherever today's launches have a dependency (GroupNorm statistics, conv -> GroupNorm, attention -> projection ...).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/grid_barrier.hip -o scripts/probes/grid_barrier && scripts/probes/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, int rounds, float* buf, long floats_per_round) {
  const unsigned G = gridDim.x;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (floats_per_round) {                     // a slice of streaming work between barriers, spread over the grid
      const long per = floats_per_round / G;
      const float* p = buf + (long)blockIdx.x * per;
      for (long i = threadIdx.x; i < per; i += 256) acc += p[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * G;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
  }
  if (acc == 12345.678f) buf[0] = acc;
}

// the same rounds as separate launches (what the product does today): one tiny kernel per round
__global__ __launch_bounds__(256) void one_round(float* buf, long floats_per_round) {
  float acc = 0.f;
  if (floats_per_round) {
    const long per = floats_per_round / gridDim.x;
    const float* p = buf + (long)blockIdx.x * per;
    for (long i = threadIdx.x; i < per; i += 256) acc += p[i];
  }
  if (acc == 12345.678f) buf[0] = acc;
}

int main() {
  unsigned* counter; float* buf;
  const long max_floats = 16l << 20;
  hipMalloc(&counter, 4); hipMalloc(&buf, max_floats * 4); hipMemset(buf, 0, max_floats * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int rounds = 200;
  for (long mb : {0l, 1l, 8l, 32l}) {
    const long fl = mb * (1l << 20) / 4;
    for (int G : {256, 512, 1024}) {
      float ms_b = 0.f, ms_l = 0.f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counter, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(barrier_loop, dim3(G), dim3(256), 0, 0, counter, rounds, buf, fl);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_b, e0, e1);
        hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(one_round, dim3(G), dim3(256), 0, 0, buf, fl);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_l, e0, e1);
      }
      printf("%3ld MB per round, %4d workgroups: persistent + grid barrier %7.2f us per round   one launch per round %7.2f us per round\n",
             mb, G, ms_b * 1e3 / rounds, ms_l * 1e3 / rounds);
    }
  }
  return 0;
}
