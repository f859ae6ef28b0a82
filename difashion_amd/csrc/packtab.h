// Table-driven weight packing / gradient un-packing: every PackOp of the model in ONE launch.
// (pack() used to be ~1100 launches of tiny kernels per optimizer step, the gradient un-pack ~700 per backward: host launch
// time, not HBM time, set their cost.)
#pragma once
#include "dfh_common.h"

enum TabKind {
  TAB_PACK_VEC = 0, TAB_PACK_MAT = 1, TAB_PACK_CONV = 2,      // master fp32 -> arena32 / arena16
  TAB_PACKT_MAT = 3, TAB_PACKT_CONV = 4,                      // master fp32 -> arena16t (transposed packs)
  TAB_UNPACK_VEC = 5, TAB_UNPACK_MAT = 6, TAB_UNPACK_CONV = 7, // grad32 / grad16 (packed fp32) -> master .grad (+=)
  // training re-pack: ONE read of a master weight feeds both the plain pack (arena16: dst / ld / p*) and the transposed pack (arena16t:
  // dst2 / ld2 / q*) -- PACK_MAT + PACKT_MAT, PACK_CONV + PACKT_CONV of the same parameter in one pass
  TAB_PACK2_MAT = 8, TAB_PACK2_CONV = 9
};

struct TabOp {
  void* master;            // fp32 parameter (pack: read) or its gradient (unpack: +=)
  long dst;                // element offset inside the arena
  int kind, N, K, ld, p0, p1, p2, p3;   // per kind: see packtab.hip
  unsigned first_block;    // first block of this op in the fused grid
  long dst2;               // PACK2 kinds: element offset inside the transposed arena
  int ld2, q0, q1, q3;     // PACK2 kinds: ldt, t_row_off, t_col_off, o_pad of the transposed pack
};

constexpr int TAB_ELEMS_PER_BLOCK = 2048;

namespace dfh {
unsigned tab_blocks(int kind, int N, int K);   // blocks one op occupies in the fused grid
// arena: arena32 (VEC) / arena16 / arena16t / grad32 / grad16 according to the op kinds in the table
// arena_mat2: the transposed arena of the PACK2 kinds (nullptr otherwise)
// sq_partials (un-pack tables): one float per block = sum of the squares of the gradient values the block wrote
int table_launch(const TabOp* dev_ops, int nops, unsigned total_blocks, void* arena_vec, void* arena_mat, hipStream_t s, void* arena_mat2 = nullptr,
                 float* sq_partials = nullptr);
// out[0] = sum of the n block partials, in a fixed order; scratch = 257 floats (256 block sums + a zero-initialised ticket counter)
int table_sq_reduce_launch(const float* partials, long n, float* scratch, float* out, hipStream_t s);
}
