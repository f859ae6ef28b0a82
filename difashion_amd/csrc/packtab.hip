// One-launch weight packing / gradient un-packing over a device-resident op table (see packtab.h).
// Element semantics are those of the single-op kernels in elementwise.hip / bwd_elementwise.hip (kept for the op-level
// C ABI and tests): fp32 master layouts of diffusers <-> K-contiguous bf16 GEMM layouts.
#include "packtab.h"

namespace {

DFH_DEVICE int tab_geglu_row(int n, int N) {
  const int half = N >> 1;
  const int j = n < half ? n : n - half;
  return (j >> 4) * 32 + (n < half ? 0 : 16) + (j & 15);
}

// Work decomposition per kind (tab_blocks() below): every global access is a coalesced run --
//   VEC / PACK_MAT / UNPACK_MAT : 2048 consecutive master elements per block (k is contiguous on both sides);
//   PACK_CONV / UNPACK_CONV     : 256 (o, c) pairs per block, each thread walks its 9 taps (36 contiguous master bytes;
//                                 per tap the wave touches 64 consecutive packed channels);
//   PACKT_MAT / PACKT_CONV      : 32 x 32 tiles transposed through LDS (rows of the master become columns of the pack).
__global__ __launch_bounds__(256) void table_kernel(const TabOp* __restrict__ ops, int nops, void* arena_vec, void* arena_mat) {
  __shared__ int s_op;
  __shared__ bf16_t tile[32][32 * 9 + 2];
  if (threadIdx.x == 0) {
    int lo = 0, hi = nops - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (ops[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    s_op = lo;
  }
  __syncthreads();
  const TabOp op = ops[s_op];
  const long blk = (long)(blockIdx.x - op.first_block);
  const int N = op.N, K = op.K, ld = op.ld, tid = threadIdx.x;
  switch (op.kind) {
    case TAB_PACK_VEC: {        // p0 = geglu, p1 = accumulate
      for (int j = 0; j < TAB_ELEMS_PER_BLOCK / 256; ++j) {
        const long i = blk * TAB_ELEMS_PER_BLOCK + j * 256 + tid;
        if (i >= N) break;
        const int r = op.p0 ? tab_geglu_row((int)i, N) : (int)i;
        float* out = (float*)arena_vec + op.dst + r;
        const float v = ((const float*)op.master)[i];
        *out = op.p1 ? *out + v : v;
      }
      break;
    }
    case TAB_UNPACK_VEC: {      // p0 = geglu, p1 = overwrite
      for (int j = 0; j < TAB_ELEMS_PER_BLOCK / 256; ++j) {
        const long i = blk * TAB_ELEMS_PER_BLOCK + j * 256 + tid;
        if (i >= N) break;
        const int r = op.p0 ? tab_geglu_row((int)i, N) : (int)i;
        const float gv = ((const float*)arena_vec)[op.dst + r];
        ((float*)op.master)[i] = op.p1 ? gv : ((float*)op.master)[i] + gv;      // p1 = overwrite
      }
      break;
    }
    case TAB_PACK_MAT: case TAB_UNPACK_MAT: {        // p0 = row_off, p1 = col_off, p2 = geglu, p3 = overwrite (unpack)
      for (int j = 0; j < TAB_ELEMS_PER_BLOCK / 256; ++j) {
        const long i = blk * TAB_ELEMS_PER_BLOCK + j * 256 + tid;
        if (i >= (long)N * K) break;
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        const int r = op.p2 ? tab_geglu_row(n, N) : n;
        const long at = op.dst + (long)(op.p0 + r) * ld + op.p1 + k;
        if (op.kind == TAB_PACK_MAT) ((bf16_t*)arena_mat)[at] = f2bf(((const float*)op.master)[i]);
        else ((float*)op.master)[i] = op.p3 ? ((const float*)arena_mat)[at] : ((float*)op.master)[i] + ((const float*)arena_mat)[at];
      }
      break;
    }
    case TAB_PACK_CONV: case TAB_UNPACK_CONV: {      // N = Cout, K = Cin, p1 = col_off, p3 = cin_pad, p0 = overwrite (unpack)
      const long oc = blk * 256 + tid;
      if (oc >= (long)N * K) break;
      const int c = (int)(oc % K), o = (int)(oc / K);
      const long at = op.dst + (long)o * ld + op.p1 + c;
      if (op.kind == TAB_PACK_CONV) {
        const float* w = (const float*)op.master + oc * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) ((bf16_t*)arena_mat)[at + t * op.p3] = f2bf(w[t]);
      } else {
        float* g = (float*)op.master + oc * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t] = op.p0 ? ((const float*)arena_mat)[at + t * op.p3] : g[t] + ((const float*)arena_mat)[at + t * op.p3];
      }
      break;
    }
    case TAB_PACKT_MAT: {       // out[(p0 + k) * ld + p1 + r(n)] = w[n][k];  p2 = geglu; tiles of 32 n x 32 k
      const int tk = (K + 31) / 32;
      const int n0 = (int)(blk / tk) * 32, k0 = (int)(blk % tk) * 32;
      for (int j = tid; j < 1024; j += 256) {
        const int n = n0 + (j >> 5), k = k0 + (j & 31);
        tile[j >> 5][j & 31] = (n < N && k < K) ? f2bf(((const float*)op.master)[(long)n * K + k]) : (bf16_t)0;
      }
      __syncthreads();
      for (int j = tid; j < 1024; j += 256) {
        const int k = k0 + (j >> 5), n = n0 + (j & 31);
        if (n < N && k < K) {
          const int r = op.p2 ? tab_geglu_row(n, N) : n;      // runs of 16 consecutive n stay consecutive
          ((bf16_t*)arena_mat)[op.dst + (long)(op.p0 + k) * ld + op.p1 + r] = tile[j & 31][j >> 5];
        }
      }
      break;
    }
    case TAB_PACKT_CONV: {      // out[c * ld + p1 + (8 - t) * o_pad + o] = w[o][c][t];  N = Cout, K = Cin, p3 = o_pad; 32 o x 32 c tiles
      const int tc = (K + 31) / 32;
      const int o0 = (int)(blk / tc) * 32, c0 = (int)(blk % tc) * 32;
      const int cw = min(32, K - c0);                // channels of this tile: a master row piece of cw * 9 contiguous floats
      for (int ol = 0; ol < 32; ++ol) {
        const int o = o0 + ol;
        for (int j = tid; j < cw * 9; j += 256)
          tile[ol][j] = o < N ? f2bf(((const float*)op.master)[((long)o * K + c0) * 9 + j]) : (bf16_t)0;
      }
      __syncthreads();
      for (int j = tid; j < cw * 9 * 32; j += 256) {
        const int ol = j & 31, ct = j >> 5;          // ct = c_local * 9 + t
        const int cl = ct / 9, t = ct - cl * 9, o = o0 + ol;
        if (o < N) ((bf16_t*)arena_mat)[op.dst + (long)(c0 + cl) * ld + op.p1 + (8 - t) * op.p3 + o] = tile[ol][ct];
      }
      break;
    }
    default: break;
  }
}

}  // namespace

namespace dfh {
unsigned tab_blocks(int kind, int N, int K) {
  switch (kind) {
    case TAB_PACK_VEC: case TAB_UNPACK_VEC: return (unsigned)((N + TAB_ELEMS_PER_BLOCK - 1) / TAB_ELEMS_PER_BLOCK);
    case TAB_PACK_MAT: case TAB_UNPACK_MAT: return (unsigned)(((long)N * K + TAB_ELEMS_PER_BLOCK - 1) / TAB_ELEMS_PER_BLOCK);
    case TAB_PACK_CONV: case TAB_UNPACK_CONV: return (unsigned)(((long)N * K + 255) / 256);
    default: return (unsigned)(((N + 31) / 32) * ((K + 31) / 32));       // transposed packs: 32 x 32 tiles
  }
}
int table_launch(const TabOp* dev_ops, int nops, unsigned total_blocks, void* arena_vec, void* arena_mat, hipStream_t s) {
  if (nops <= 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(table_kernel, dim3(total_blocks), dim3(256), 0, s, dev_ops, nops, arena_vec, arena_mat);
  return check_launch("table_kernel");
}
}  // namespace dfh
