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
//   PACK_CONV / UNPACK_CONV     : 256 (o, c) pairs per block = 2304 consecutive master floats, moved as float4 and turned through LDS
//                                 (per tap the wave touches 64 consecutive packed channels);
//   PACKT_MAT / PACKT_CONV      : 64 x 32 tiles transposed through LDS (rows of the master become columns of the pack; 64 outputs =
//                                 one full 128-byte line per store).
constexpr int TT_O = 64, TT_C = 32, TT_LD = TT_C * 9 + 2;       // transposed-pack tile: 64 outputs x 32 inputs (x 9 taps), odd dword stride
__global__ __launch_bounds__(256) void table_kernel(const TabOp* __restrict__ ops, int nops, void* arena_vec, void* arena_mat, void* arena_mat2,
                                                    float* __restrict__ sq_partials) {
  __shared__ int s_op;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[TT_O * TT_LD * 2];
  bf16_t (*tile)[TT_LD] = (bf16_t (*)[TT_LD])s_raw;
  float* stage = (float*)s_raw;                                  // conv kinds: 256 pairs x 9 taps
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
  float sq = 0.f;              // un-pack kinds: sum of the squares of the gradient values this thread wrote (sq_partials, below)
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
        const float nv = op.p1 ? gv : ((float*)op.master)[i] + gv;              // p1 = overwrite
        ((float*)op.master)[i] = nv;
        sq += nv * nv;
      }
      break;
    }
    case TAB_PACK_MAT: case TAB_UNPACK_MAT: {        // p0 = row_off, p1 = col_off, p2 = geglu, p3 = overwrite (unpack)
      const long total = (long)N * K;
      if ((K & 3) == 0 && ((uintptr_t)op.master & 15) == 0 && ((op.dst | ld | op.p1) & 3) == 0) {      // four consecutive k of one row per thread
        for (int j = 0; j < TAB_ELEMS_PER_BLOCK / 1024; ++j) {
          const long i = blk * TAB_ELEMS_PER_BLOCK + (j * 256 + tid) * 4;
          if (i >= total) break;
          const int n = (int)(i / K), k = (int)(i - (long)n * K);
          const int r = op.p2 ? tab_geglu_row(n, N) : n;
          const long at = op.dst + (long)(op.p0 + r) * ld + op.p1 + k;
          float4* m4 = (float4*)((float*)op.master + i);
          if (op.kind == TAB_PACK_MAT) {
            const float4 v = *m4;
            *(uint2*)((bf16_t*)arena_mat + at) = uint2{pack2bf(v.x, v.y), pack2bf(v.z, v.w)};
          } else {
            float4 g = *(const float4*)((const float*)arena_mat + at);
            if (!op.p3) { const float4 v = *m4; g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w; }
            *m4 = g;
            sq += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
          }
        }
        break;
      }
      for (int j = 0; j < TAB_ELEMS_PER_BLOCK / 256; ++j) {
        const long i = blk * TAB_ELEMS_PER_BLOCK + j * 256 + tid;
        if (i >= total) break;
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        const int r = op.p2 ? tab_geglu_row(n, N) : n;
        const long at = op.dst + (long)(op.p0 + r) * ld + op.p1 + k;
        if (op.kind == TAB_PACK_MAT) ((bf16_t*)arena_mat)[at] = f2bf(((const float*)op.master)[i]);
        else {
          const float nv = op.p3 ? ((const float*)arena_mat)[at] : ((float*)op.master)[i] + ((const float*)arena_mat)[at];
          ((float*)op.master)[i] = nv;
          sq += nv * nv;
        }
      }
      break;
    }
    case TAB_PACK_CONV: case TAB_UNPACK_CONV: {      // N = Cout, K = Cin, p1 = col_off, p3 = cin_pad, p0 = overwrite (unpack)
      const long pairs = (long)N * K, oc = blk * 256 + tid;
      const long f0 = blk * 2304, fn = min((long)2304, pairs * 9 - f0);      // this block's run of master floats
      float* mst = (float*)op.master + f0;
      const bool vec4 = ((uintptr_t)mst & 15) == 0 && (fn & 3) == 0;
      const bool live = oc < pairs;
      const int c = live ? (int)(oc % K) : 0, o = live ? (int)(oc / K) : 0;
      const long at = op.dst + (long)o * ld + op.p1 + c;
      if (op.kind == TAB_PACK_CONV) {                  // reads: nine strided passes over the same 36 lines, absorbed by the L1
        if (live) {
          const float* w = (const float*)op.master + oc * 9;
#pragma unroll
          for (int t = 0; t < 9; ++t) ((bf16_t*)arena_mat)[at + t * op.p3] = f2bf(w[t]);
        }
      } else {
        if (live) {
#pragma unroll
          for (int t = 0; t < 9; ++t) stage[tid * 9 + t] = ((const float*)arena_mat)[at + t * op.p3];
        }
        __syncthreads();
        if (vec4) {
          for (int j = tid * 4; j < fn; j += 1024) {
            float4 v = *(const float4*)(stage + j);
            if (!op.p0) { const float4 g = *(const float4*)(mst + j); v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w; }
            *(float4*)(mst + j) = v;
            sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          }
        } else {
          for (int j = tid; j < fn; j += 256) { const float nv = op.p0 ? stage[j] : mst[j] + stage[j]; mst[j] = nv; sq += nv * nv; }
        }
      }
      break;
    }
    case TAB_PACKT_MAT: {       // out[(p0 + k) * ld + p1 + r(n)] = w[n][k];  p2 = geglu; tiles of 64 n x 32 k
      const int tk = (K + TT_C - 1) / TT_C;
      const int n0 = (int)(blk / tk) * TT_O, k0 = (int)(blk % tk) * TT_C;
      const bool fast = (K & 31) == 0 && n0 + TT_O <= N && ((uintptr_t)op.master & 15) == 0 && ((op.dst | ld | op.p1) & 1) == 0;
      if (fast) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {                // 64 rows x 8 float4
          const int j = u * 256 + tid, nl = j >> 3, q = j & 7;
          const float4 v = *(const float4*)((const float*)op.master + (long)(n0 + nl) * K + k0 + q * 4);
          *(uint32_t*)&tile[nl][q * 4] = pack2bf(v.x, v.y);
          *(uint32_t*)&tile[nl][q * 4 + 2] = pack2bf(v.z, v.w);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {                // 32 k x 32 pairs of n: one 128-byte line per 32 lanes
          const int j = u * 256 + tid, k = j >> 5, nl = (j & 31) * 2;
          const uint32_t pr = (uint32_t)tile[nl][k] | ((uint32_t)tile[nl + 1][k] << 16);
          const int r = op.p2 ? tab_geglu_row(n0 + nl, N) : n0 + nl;      // runs of 16 consecutive n stay consecutive: pairs stay adjacent
          *(uint32_t*)((bf16_t*)arena_mat + op.dst + (long)(op.p0 + k0 + k) * ld + op.p1 + r) = pr;
        }
        break;
      }
      for (int j = tid; j < TT_O * TT_C; j += 256) {
        const int n = n0 + (j >> 5), k = k0 + (j & 31);
        tile[j >> 5][j & 31] = (n < N && k < K) ? f2bf(((const float*)op.master)[(long)n * K + k]) : (bf16_t)0;
      }
      __syncthreads();
      for (int j = tid; j < TT_O * TT_C; j += 256) {
        const int k = k0 + (j >> 6), n = n0 + (j & 63);
        if (n < N && k < K) {
          const int r = op.p2 ? tab_geglu_row(n, N) : n;      // runs of 16 consecutive n stay consecutive
          ((bf16_t*)arena_mat)[op.dst + (long)(op.p0 + k) * ld + op.p1 + r] = tile[j & 63][j >> 6];
        }
      }
      break;
    }
    case TAB_PACKT_CONV: {      // out[c * ld + p1 + (8 - t) * o_pad + o] = w[o][c][t];  N = Cout, K = Cin, p3 = o_pad; 64 o x 32 c tiles
      const int tc = (K + TT_C - 1) / TT_C;
      const int o0 = (int)(blk / tc) * TT_O, c0 = (int)(blk % tc) * TT_C;
      const int cw = min(TT_C, K - c0);              // channels of this tile: a master row piece of cw * 9 contiguous floats
      const int run = cw * 9;
      const bool fast = (K & 31) == 0 && o0 + TT_O <= N && ((uintptr_t)op.master & 15) == 0 && ((op.dst | ld | op.p1 | op.p3) & 1) == 0;
      if (fast) {                                    // run = 288 floats = 72 float4 per output row; fixed trip counts, loads batched
#pragma unroll 6
        for (int u = 0; u < 18; ++u) {
          const int j = u * 256 + tid, ol = j / 72, q = j - ol * 72;
          const float4 v = *(const float4*)((const float*)op.master + ((long)(o0 + ol) * K + c0) * 9 + q * 4);
          *(uint32_t*)&tile[ol][q * 4] = pack2bf(v.x, v.y);
          *(uint32_t*)&tile[ol][q * 4 + 2] = pack2bf(v.z, v.w);
        }
        __syncthreads();
#pragma unroll 6
        for (int u = 0; u < 36; ++u) {               // 288 (c, t) columns x 32 pairs of outputs
          const int j = u * 256 + tid, ct = j >> 5, ol = (j & 31) * 2;
          const int cl = ct / 9, t = ct - cl * 9;
          const uint32_t pr = (uint32_t)tile[ol][ct] | ((uint32_t)tile[ol + 1][ct] << 16);
          *(uint32_t*)((bf16_t*)arena_mat + op.dst + (long)(c0 + cl) * ld + op.p1 + (8 - t) * op.p3 + o0 + ol) = pr;
        }
        break;
      }
      for (int j = tid; j < TT_O * run; j += 256) {
        const int ol = j / run, jj = j - ol * run, o = o0 + ol;
        tile[ol][jj] = o < N ? f2bf(((const float*)op.master)[((long)o * K + c0) * 9 + jj]) : (bf16_t)0;
      }
      __syncthreads();
      for (int j = tid; j < run * TT_O; j += 256) {
        const int ol = j & 63, ct = j >> 6;          // ct = c_local * 9 + t
        const int cl = ct / 9, t = ct - cl * 9, o = o0 + ol;
        if (o < N) ((bf16_t*)arena_mat)[op.dst + (long)(c0 + cl) * ld + op.p1 + (8 - t) * op.p3 + o] = tile[ol][ct];
      }
      break;
    }
    case TAB_PACK2_MAT: {       // PACK_MAT (dst, ld, p0 = row_off, p1 = col_off, p2 = geglu) + PACKT_MAT (dst2, ld2, q0, q1) from one read
      const int tk = (K + TT_C - 1) / TT_C;
      const int n0 = (int)(blk / tk) * TT_O, k0 = (int)(blk % tk) * TT_C;
      const bool fast = (K & 31) == 0 && n0 + TT_O <= N && ((uintptr_t)op.master & 15) == 0 && ((op.dst2 | op.ld2 | op.q1) & 1) == 0 &&
                        ((op.dst | ld | op.p1) & 1) == 0;
      if (fast) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = u * 256 + tid, nl = j >> 3, q = j & 7;
          const float4 v = *(const float4*)((const float*)op.master + (long)(n0 + nl) * K + k0 + q * 4);
          const uint32_t lo = pack2bf(v.x, v.y), hi = pack2bf(v.z, v.w);
          *(uint32_t*)&tile[nl][q * 4] = lo;
          *(uint32_t*)&tile[nl][q * 4 + 2] = hi;
          const int r = op.p2 ? tab_geglu_row(n0 + nl, N) : n0 + nl;       // plain pack: straight from the registers, 8 bytes per lane
          *(uint2*)((bf16_t*)arena_mat + op.dst + (long)(op.p0 + r) * ld + op.p1 + k0 + q * 4) = uint2{lo, hi};
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = u * 256 + tid, k = j >> 5, nl = (j & 31) * 2;
          const uint32_t pr = (uint32_t)tile[nl][k] | ((uint32_t)tile[nl + 1][k] << 16);
          const int r = op.p2 ? tab_geglu_row(n0 + nl, N) : n0 + nl;
          *(uint32_t*)((bf16_t*)arena_mat2 + op.dst2 + (long)(op.q0 + k0 + k) * op.ld2 + op.q1 + r) = pr;
        }
        break;
      }
      for (int j = tid; j < TT_O * TT_C; j += 256) {
        const int n = n0 + (j >> 5), k = k0 + (j & 31);
        bf16_t v = 0;
        if (n < N && k < K) {
          v = f2bf(((const float*)op.master)[(long)n * K + k]);
          const int r = op.p2 ? tab_geglu_row(n, N) : n;
          ((bf16_t*)arena_mat)[op.dst + (long)(op.p0 + r) * ld + op.p1 + k] = v;
        }
        tile[j >> 5][j & 31] = v;
      }
      __syncthreads();
      for (int j = tid; j < TT_O * TT_C; j += 256) {
        const int k = k0 + (j >> 6), n = n0 + (j & 63);
        if (n < N && k < K) {
          const int r = op.p2 ? tab_geglu_row(n, N) : n;
          ((bf16_t*)arena_mat2)[op.dst2 + (long)(op.q0 + k) * op.ld2 + op.q1 + r] = tile[j & 63][j >> 6];
        }
      }
      break;
    }
    case TAB_PACK2_CONV: {      // PACK_CONV (dst, ld, p1 = col_off, p3 = cin_pad) + PACKT_CONV (dst2, ld2, q1 = t_col_off, q3 = o_pad) from one read
      const int tc = (K + TT_C - 1) / TT_C;
      const int o0 = (int)(blk / tc) * TT_O, c0 = (int)(blk % tc) * TT_C;
      const int cw = min(TT_C, K - c0);
      const int run = cw * 9;
      const bool fast = (K & 31) == 0 && o0 + TT_O <= N && ((uintptr_t)op.master & 15) == 0 && ((op.dst2 | op.ld2 | op.q1 | op.q3) & 1) == 0 &&
                        ((op.dst | ld | op.p1 | op.p3) & 1) == 0;
      if (fast) {
#pragma unroll 6
        for (int u = 0; u < 18; ++u) {
          const int j = u * 256 + tid, ol = j / 72, q = j - ol * 72;
          const float4 v = *(const float4*)((const float*)op.master + ((long)(o0 + ol) * K + c0) * 9 + q * 4);
          *(uint32_t*)&tile[ol][q * 4] = pack2bf(v.x, v.y);
          *(uint32_t*)&tile[ol][q * 4 + 2] = pack2bf(v.z, v.w);
        }
        __syncthreads();
#pragma unroll 6
        for (int u = 0; u < 36; ++u) {               // transposed pack: 288 (c, t) columns x 32 pairs of outputs
          const int j = u * 256 + tid, ct = j >> 5, ol = (j & 31) * 2;
          const int cl = ct / 9, t = ct - cl * 9;
          const uint32_t pr = (uint32_t)tile[ol][ct] | ((uint32_t)tile[ol + 1][ct] << 16);
          *(uint32_t*)((bf16_t*)arena_mat2 + op.dst2 + (long)(c0 + cl) * op.ld2 + op.q1 + (8 - t) * op.q3 + o0 + ol) = pr;
        }
#pragma unroll 6
        for (int u = 0; u < 36; ++u) {               // plain pack: 64 outputs x 9 taps x 16 pairs of channels (64-byte runs per (o, t))
          const int j = u * 256 + tid, cl = (j & 15) * 2, ot = j >> 4;
          const int ol = ot / 9, t = ot - ol * 9;
          const uint32_t pr = (uint32_t)tile[ol][cl * 9 + t] | ((uint32_t)tile[ol][(cl + 1) * 9 + t] << 16);
          *(uint32_t*)((bf16_t*)arena_mat + op.dst + (long)(o0 + ol) * ld + op.p1 + t * op.p3 + c0 + cl) = pr;
        }
        break;
      }
      for (int j = tid; j < TT_O * run; j += 256) {
        const int ol = j / run, jj = j - ol * run, o = o0 + ol;
        bf16_t v = 0;
        if (o < N) {
          v = f2bf(((const float*)op.master)[((long)o * K + c0) * 9 + jj]);
          const int cl = jj / 9, t = jj - cl * 9;
          ((bf16_t*)arena_mat)[op.dst + (long)o * ld + op.p1 + t * op.p3 + c0 + cl] = v;
        }
        tile[ol][jj] = v;
      }
      __syncthreads();
      for (int j = tid; j < run * TT_O; j += 256) {
        const int ol = j & 63, ct = j >> 6;
        const int cl = ct / 9, t = ct - cl * 9, o = o0 + ol;
        if (o < N) ((bf16_t*)arena_mat2)[op.dst2 + (long)(c0 + cl) * op.ld2 + op.q1 + (8 - t) * op.q3 + o] = tile[ol][ct];
      }
      break;
    }
    default: break;
  }
  // gradient norm (dfh_unet_grad_sumsq): one partial per block, fixed order inside the block; table_sq_reduce sums the blocks in order
  if (sq_partials) {
    __shared__ float sq_red[4];
    sq = wave_sum(sq);
    __syncthreads();                                  // (the cases above are done with the shared tile)
    if ((tid & 63) == 0) sq_red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) sq_partials[blockIdx.x] = (sq_red[0] + sq_red[1]) + (sq_red[2] + sq_red[3]);
  }
}

// out[0] = sum of p[0..n): 256 blocks of 256 threads, each thread a strided run in ascending order, then the block, then (the last block
// to arrive, by ticket) the 256 block sums in order -- no float atomics: the same bits on every run and every rank
__global__ __launch_bounds__(256) void table_sq_reduce_kernel(const float* __restrict__ p, long n, float* __restrict__ block_sums,
                                                              unsigned* __restrict__ counter, float* __restrict__ out) {
  __shared__ float red[4];
  __shared__ bool last;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += p[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    block_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __threadfence();
    last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = threadIdx.x < gridDim.x ? ((const volatile float*)block_sums)[threadIdx.x] : 0.f;
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) { *out = (red[0] + red[1]) + (red[2] + red[3]); *counter = 0u; }
}

}  // namespace

namespace dfh {
unsigned tab_blocks(int kind, int N, int K) {
  switch (kind) {
    case TAB_PACK_VEC: case TAB_UNPACK_VEC: return (unsigned)((N + TAB_ELEMS_PER_BLOCK - 1) / TAB_ELEMS_PER_BLOCK);
    case TAB_PACK_MAT: case TAB_UNPACK_MAT: return (unsigned)(((long)N * K + TAB_ELEMS_PER_BLOCK - 1) / TAB_ELEMS_PER_BLOCK);
    case TAB_PACK_CONV: case TAB_UNPACK_CONV: return (unsigned)(((long)N * K + 255) / 256);
    default: return (unsigned)(((N + 63) / 64) * ((K + 31) / 32));       // transposed packs: 64 x 32 tiles
  }
}
int table_launch(const TabOp* dev_ops, int nops, unsigned total_blocks, void* arena_vec, void* arena_mat, hipStream_t s, void* arena_mat2,
                 float* sq_partials) {
  if (nops <= 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(table_kernel, dim3(total_blocks), dim3(256), 0, s, dev_ops, nops, arena_vec, arena_mat, arena_mat2, sq_partials);
  return check_launch("table_kernel");
}
int table_sq_reduce_launch(const float* partials, long n, float* scratch, float* out, hipStream_t s) {
  // scratch: 256 block sums + the ticket counter (zeroed once by the owner; the kernel resets it)
  hipLaunchKernelGGL(table_sq_reduce_kernel, dim3(256), dim3(256), 0, s, partials, n, scratch, (unsigned*)(scratch + 256), out);
  return check_launch("table_sq_reduce_kernel");
}
}  // namespace dfh
