// Weight-gradient GEMM and bias / row-vector gradient reductions for the training step
// (reference: accelerator.backward(loss) at DiFashion/train.py:699 -> torch autograd of every
// conv3x3 / 1x1 conv / linear on the path).
//
//   dW[n][kcol] += sum_m dY[m][n] * A[m][kcol]
// where A[m][kcol] is exactly the forward implicit-GEMM operand (gemm.hip): kcol runs over the same
// K segments (nine conv taps of Cin, then plain row segments), so stride-2, fused nearest-2x upsample
// and concatenated inputs need no extra code.  The contraction index is m (pixels), which is the SLOW
// index of both operands in HBM; tiles are staged m-major by LDS-DMA exactly like the forward pass and
// the MFMA fragments are read TRANSPOSED with gfx950's ds_read_b64_tr_b16 (lane L of a 16-lane group
// points at &T[m0 + L/4][n0 + 4*(L%4)] and receives T[m0..m0+3][n0 + L]; verified on hardware,
// scripts/probes/tr_probe.hip).  D^T[kcol][n] is accumulated so a lane ends with 4 consecutive kcol of
// one n = 16 contiguous bytes of the packed fp32 gradient.  M is split over blockIdx.z with fp32 atomics.
#include "gemm.h"
#include "wgrad.h"

namespace {

// Block tile: 160 output channels (n) x 160 packed weight columns (kcol); 4 waves as 2 x 2, each 80 x 80 (5 x 5 MFMA
// fragments, 100 accumulator VGPRs).  160 divides every channel count of the model (320/640/960/1280/1920/2560).
// The contraction runs over pixels m in stages of 32 rows (one 16x16x32 MFMA k-step), 3 stages in flight.
constexpr int TN = 160, TK = 160, BM = 32, NST = 3;
// Both operand tiles are [32 m][160 cols] bf16 with a 352-byte row stride (320 B of data + 32 B pad): one
// ds_read_b64_tr_b16 lane group (32 lanes) touches 8 consecutive m rows x 32 B, and 352 = 96 (mod 256) spreads those
// over all 64 banks (0,96,192,32,128,224,64,160) -- conflict-free, where 256-/128-byte rows are 8-/4-way conflicted.
constexpr int ROWB = 352, TILE_BYTES = BM * ROWB;          // 11264 B = 11 LDS-DMA instructions of 1 KB
constexpr int REGION = 12 * 1024;                           // per operand: 12 slots (the 12th only holds the tail padding)
constexpr int WSTAGE = 2 * REGION;                          // every wave issues 3 dY + 3 A slots per stage
constexpr int ISSUE = 6;

// ds_read_b64_tr_b16 through inline asm: the builtin makes hipcc wait vmcnt(0) before the first LDS read of every
// stage (it cannot see that the LDS-DMA in flight targets another stage), which serialises the whole pipeline.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// wait until at most N LDS reads are outstanding; the fragments it guards are tied in as operands so that no MFMA
// consuming them can be scheduled above the wait
#define LGKM_WAIT5(N, f) asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), \
    "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]), "+v"(f[4][0]), "+v"(f[4][1]) : "n"(N))
#define LGKM_WAIT1(N, f) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N))

DFH_DEVICE bf16x8_t frag_of(const u32x2_t lo, const u32x2_t hi) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8_t, v);
}
template <int N> DFH_DEVICE void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// x / d for 0 <= x < 2^24 through a float reciprocal (exact after one fix-up step)
DFH_DEVICE int fdiv(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  const int r = x - q * d;
  q += (r >= d) - (r < 0);
  return q;
}

// One output tile of the launch: bx = n-tile + ntn * chunk.  Conv chunks are enumerated channel-major (chunk = channel_chunk * 9 + tap): a
// contiguous chunk range -- what one XCD gets when there is a single m-slice -- then re-reads ONE 160-channel slice of the activation under
// nine shifts (L2-resident) instead of all channels.
struct WgTile { int n0, chunk, seg, c0, wcol, seglen; };
DFH_DEVICE WgTile wg_tile(const WgradArgs& a, int bx) {
  WgTile t;
  const int ntn = (a.N + TN - 1) / TN;
  t.n0 = (bx % ntn) * TN;
  t.chunk = bx / ntn;
  int base, kc = t.chunk;
  const int cchunks = (a.conv_c + TK - 1) / TK;
  if (kc < a.ntaps * cchunks) {
    t.seg = kc % a.ntaps; kc /= a.ntaps; t.seglen = a.conv_c; base = t.seg * a.conv_c;
  } else {
    kc -= a.ntaps * cchunks; t.seg = a.ntaps; base = a.ntaps * a.conv_c; t.seglen = a.p_c[0];
    const int n0c = (t.seglen + TK - 1) / TK;
    if (a.nplain > 1 && kc >= n0c) { kc -= n0c; base += t.seglen; t.seglen = a.p_c[1]; t.seg = a.ntaps + 1; }
  }
  t.c0 = kc * TK; t.wcol = base + t.c0;
  return t;
}
// fp32 slabs are TILE-LOCAL: slot [40 kcol quads][160 n] float4, so that the 16 lanes of an MFMA column group store 256 contiguous bytes
constexpr int SLOT_FLOATS = TN * TK;

// One PIECE of the launch: rows [m_begin, m_end) of output tile bx, accumulated over the pixels and stored either into the fp32 slab slot
// `slab` or (slab == nullptr) straight into dW.  POW2: Hout and Wout are powers of two (every level of the U-Net): pixel decode by shifts.
template <bool POW2>
DFH_DEVICE void wgrad_piece(const WgradArgs& a, unsigned char* smem, const int bx, const int m_begin, const int m_end, float* slab) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = lane & 15, fg = lane >> 4;
  const int wave_n = wave & 1, wave_k = wave >> 1;
  const WgTile tl = wg_tile(a, bx);
  const int n0 = tl.n0, chunk = tl.chunk, seg = tl.seg, c0 = tl.c0, wcol = tl.wcol, seglen = tl.seglen;
  const bool conv = seg < a.ntaps;
  // tap2 (phase plane (py, px) of an upsample conv, wgrad.h): the four segments are the 3x3-tap positions (py + (s >> 1), px + (s & 1))
  const int ky = conv ? (a.tap2 ? a.tap_py + (seg >> 1) : seg / 3) : 0, kx = conv ? (a.tap2 ? a.tap_px + (seg & 1) : seg - (seg / 3) * 3) : 0;
  const bf16_t* psrc = conv ? a.conv_src : (seg == a.ntaps ? a.p_src[0] : a.p_src[1]);
  const int pc = seglen;
  const int nsteps = m_end > m_begin ? (m_end - m_begin + BM - 1) / BM : 0;

  const int HWo = a.Hout * a.Wout;
  const float inv_hw = 1.0f / (float)HWo, inv_w = 1.0f / (float)a.Wout;
  const int lw = 31 - __builtin_clz(a.Wout), lhw = 31 - __builtin_clz(HWo);
  const int Hv = a.ups ? a.Hin * 2 : a.Hin, Wv = a.ups ? a.Win * 2 : a.Win;

  // per-lane geometry of this wave's LDS-DMA slots: i = 0..2 -> dY slot i*4 + wave, i = 3..5 -> A slot (i-3)*4 + wave;
  // byte p of an operand region is (row p / 352, column (p % 352) / 2); columns >= 160 and rows >= 32 are padding
  int s_row[ISSUE], s_off[ISSUE]; bool s_ok[ISSUE];
#pragma unroll
  for (int i = 0; i < ISSUE; ++i) {
    const int p = ((i % 3) * 4 + wave) * 1024 + lane * 16;
    const int row = p / ROWB, col = (p - row * ROWB) >> 1;
    s_row[i] = row;
    if (i < 3) { s_ok[i] = (row < BM) & (col < TN) & (n0 + col < a.N); s_off[i] = n0 + col; }
    else { s_ok[i] = (row < BM) & (col < TK) & (c0 + col < seglen); s_off[i] = c0 + col; }
  }
  auto glds = [&](const bf16_t* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // straight-line, select-only address generation (a branch per slot costs more than the masked math)
  auto issue = [&](int step, int buf) {
    unsigned char* st = smem + buf * WSTAGE + wave * 1024;
    const int mb = m_begin + step * BM;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int m = mb + s_row[i];
      const bool ok = s_ok[i] & (m < m_end);
      const long off = (long)m * a.ldy + s_off[i];
      glds(ok ? a.dY + off : a.zero, st + i * 4096);
    }
    if (conv) {
#pragma unroll
      for (int i = 3; i < ISSUE; ++i) {
        const int m = mb + s_row[i];
        const int mm = min(m, a.M - 1);
        int b, oy, ox;
        if (POW2) { b = mm >> lhw; oy = (mm >> lw) & (a.Hout - 1); ox = mm & (a.Wout - 1); }
        else { b = fdiv(mm, HWo, inv_hw); const int rem = mm - b * HWo; oy = fdiv(rem, a.Wout, inv_w); ox = rem - oy * a.Wout; }
        const int yy = oy * a.stride - 1 + ky, xx = ox * a.stride - 1 + kx;
        const bool ok = s_ok[i] & (m < m_end) & ((unsigned)yy < (unsigned)Hv) & ((unsigned)xx < (unsigned)Wv);
        const int sy = a.ups ? (yy >> 1) : yy, sx = a.ups ? (xx >> 1) : xx;
        const long off = (long)((b * a.Hin + sy) * a.Win + sx) * pc + s_off[i];
        glds(ok ? psrc + off : a.zero, st + REGION + (i - 3) * 4096);
      }
    } else {
#pragma unroll
      for (int i = 3; i < ISSUE; ++i) {
        const int m = mb + s_row[i];
        const bool ok = s_ok[i] & (m < m_end);
        const long off = (long)m * pc + s_off[i];
        glds(ok ? psrc + off : a.zero, st + REGION + (i - 3) * 4096);
      }
    }
  };

  f32x4_t acc[5][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment address of this lane inside a stage: row fg*4 + L/4, 4 columns starting at (L%4)*4 of the wave's 80-column
  // slice.  MFMA k index (fg, j): j < 4 -> row fg*4 + j, j >= 4 -> row 16 + fg*4 + (j - 4) -- the same map for both
  // operands, chosen so that a 32-lane LDS group reads 8 CONSECUTIVE rows.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned ylane = lds0 + (fg * 4 + (L >> 2)) * ROWB + (wave_n * 80 + (L & 3) * 4) * 2;
  const unsigned alane = lds0 + REGION + (fg * 4 + (L >> 2)) * ROWB + (wave_k * 80 + (L & 3) * 4) * 2;

  // bias gradient (optional): the blocks of the first kcol chunk also reduce their dY tile over the pixels
  const bool do_bias = a.dbias != nullptr && chunk == 0 && wave_k == 0;
  f32x4_t accb[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) accb[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, u16x8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

  if (nsteps > 0) {
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nsteps) { issue(s, s); ++issued; }
    int buf = 0;
    for (int t = 0; t < nsteps; ++t) {
      const int ahead = issued - 1 - t;               // stages younger than t still in flight (block-uniform)
      if (ahead == 0) wg_wait_vmcnt<0>();
      else if (ahead == 1) wg_wait_vmcnt<ISSUE>();
      else wg_wait_vmcnt<2 * ISSUE>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();                   // stage t visible to all waves; everyone is done with stage t-1
      asm volatile("" ::: "memory");
      if (issued < nsteps) {
        int nb = buf - 1; if (nb < 0) nb += NST;
        issue(issued, nb);
        ++issued;
      }
      const unsigned ya = ylane + buf * WSTAGE, aa = alane + buf * WSTAGE;
      u32x2_t y[5][2], x[5][2];
#define TR_PAIR(dst, addr, nf) TR_READ(dst[nf][0], addr, nf * 32); TR_READ(dst[nf][1], addr, nf * 32 + 16 * ROWB)
      TR_PAIR(y, ya, 0); TR_PAIR(y, ya, 1); TR_PAIR(y, ya, 2); TR_PAIR(y, ya, 3); TR_PAIR(y, ya, 4);
      TR_PAIR(x, aa, 0); TR_PAIR(x, aa, 1); TR_PAIR(x, aa, 2); TR_PAIR(x, aa, 3); TR_PAIR(x, aa, 4);
#undef TR_PAIR
      LGKM_WAIT5(10, y);                              // the ten dY reads are back (LDS returns in order)
      bf16x8_t yf[5];
#pragma unroll
      for (int nf = 0; nf < 5; ++nf) yf[nf] = frag_of(y[nf][0], y[nf][1]);
      if (do_bias) {                                  // column sums of dY = a row of ones as the A operand
#pragma unroll
        for (int nf = 0; nf < 5; ++nf) accb[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[nf], accb[nf], 0, 0, 0);
      }
#define WG_ROW(kf, N)                                                                                        \
      LGKM_WAIT1(N, x[kf]);                                                                                  \
      { const bf16x8_t af = frag_of(x[kf][0], x[kf][1]);                                                      \
        _Pragma("unroll") for (int nf = 0; nf < 5; ++nf)                                                      \
          acc[kf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, yf[nf], acc[kf][nf], 0, 0, 0); }
      // D^T[row = kcol (fg*4+r)][col = n (L)]
      WG_ROW(0, 8) WG_ROW(1, 6) WG_ROW(2, 4) WG_ROW(3, 2) WG_ROW(4, 0)
#undef WG_ROW
      if (++buf == NST) buf = 0;
    }
  }

  if (do_bias && fg == 0) {
#pragma unroll
    for (int nf = 0; nf < 5; ++nf) {
      const int n = n0 + wave_n * 80 + nf * 16 + L;
      if (n < a.N) atomicAdd(a.dbias + n, accb[nf][0]);      // N x msplit atomics per layer: negligible
    }
  }
  // a lane ends with 4 consecutive packed columns of one output channel: one 16-byte access.  A piece that covers all the pixels of its
  // tile adds straight into dW (each element belongs to exactly one piece); the others write fp32 slab slots that wgrad_reduce_kernel
  // sums in a fixed order (deterministic; scalar fp32 atomics cap out near 70 G/s on this part)
  if (slab) {
#pragma unroll
    for (int nf = 0; nf < 5; ++nf) {
      const int nl = wave_n * 80 + nf * 16 + L;
#pragma unroll
      for (int kf = 0; kf < 5; ++kf) {
        const int kq = wave_k * 20 + kf * 4 + fg;          // padding rows / columns hold exact zeros (zero-page loads)
        *(float4*)(slab + ((long)kq * TN + nl) * 4) = float4{acc[kf][nf][0], acc[kf][nf][1], acc[kf][nf][2], acc[kf][nf][3]};
      }
    }
    return;
  }
#pragma unroll
  for (int nf = 0; nf < 5; ++nf) {
    const int n = n0 + wave_n * 80 + nf * 16 + L;
    if (n >= a.N) continue;
#pragma unroll
    for (int kf = 0; kf < 5; ++kf) {
      const int kcol = wave_k * 80 + kf * 16 + fg * 4;
      if (c0 + kcol >= seglen) continue;               // segment lengths are multiples of 8: all four or none
      const float4 v = float4{acc[kf][nf][0], acc[kf][nf][1], acc[kf][nf][2], acc[kf][nf][3]};
      float4* dst = (float4*)(a.dW + (long)n * a.ldw + wcol + kcol);
      if (a.overwrite) { *dst = v; continue; }
      float4 o = *dst;
      o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
      *dst = o;
    }
  }
}

// The launch as a grid of pieces: the first a.whole tiles (0 or a multiple of 512, the chip's block slots) as one block each over all the
// pixels, the remaining tiles as a.msplit equal pixel slices each.  XCD-aware within both groups: workgroups go round-robin over the 8 XCDs
// (linear id % 8), each with its own L2, and blocks of one m-slice read the SAME rows of dY and A, so the blocks an XCD receives are a
// CONTIGUOUS range of the (slice, tile) order -- one or two slices sweep through an L2 together, for any slice count (round 5: the count
// used to be 1 or a multiple of 8, which left e.g. 36 tiles x 8 slices = 288 of the 512 slots filled), and with one slice an XCD gets a
// contiguous chunk range (one 160-channel slice of the activation under nine shifts).  The whole / tail split is for the deep levels, where
// there are a few more tiles than slots but few pixels (576 tiles x 64..256 stages = 1.125 rounds of whole-tile blocks: the last 64 tiles
// go as 8 slices each, 512 short blocks, 1.125 rounds of WORK).  A stream-K decomposition of the same launches (persistent workgroups over
// tile-major stage ranges) measured slower on every shape but one: its workgroups sweep different pixel rows at the same time, so nothing
// is shared through L2 (profiles/r05/wgrad_plan_sweep_streamk.txt, wgrad_streamk_experiment.patch).
template <bool POW2>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int id = blockIdx.x, nblk = a.whole, first = 0, X = a.whole, ms = 1;
  if (id >= a.whole) { id -= a.whole; nblk = gridDim.x - a.whole; first = a.whole; X = a.xblocks - a.whole; ms = a.msplit; }
  const int xcd = id & 7, j = id >> 3;                            // a.whole is a multiple of 8: id % 8 is still the XCD
  const int g = xcd * (nblk >> 3) + min(xcd, nblk & 7) + j;       // position in the XCD-contiguous order
  const int bz = g / X, bx = g - bz * X;
  const int m_per = (((a.M + ms - 1) / ms) + BM - 1) / BM * BM;
  const int m_begin = bz * m_per, m_end = min(a.M, m_begin + m_per);
  wgrad_piece<POW2>(a, smem, first + bx, m_begin, m_end, ms > 1 ? a.partial + ((long)bz * X + bx) * SLOT_FLOATS : nullptr);
}

// dW tile (+)= sum of the tile's slab slots, in slice order.  grid (sliced tiles, 25): a block covers 8 kcol quads x 32 n of the tile-local
// slot layout (512-byte reads per quad row, 128-byte writes per output channel).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a) {
  const int X = a.xblocks - a.whole, bx = blockIdx.x;
  const WgTile tl = wg_tile(a, a.whole + bx);
  const int kq = (blockIdx.y % 5) * 8 + (threadIdx.x & 7), nl = (blockIdx.y / 5) * 32 + (threadIdx.x >> 3);
  const int n = tl.n0 + nl, kcol = kq * 4;
  if (n >= a.N || tl.c0 + kcol >= tl.seglen) return;
  const float* src = a.partial + (long)bx * SLOT_FLOATS + ((long)kq * TN + nl) * 4;
  float4 s = float4{0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < a.msplit; ++z) {
    const float4 p = *(const float4*)(src + (long)z * X * SLOT_FLOATS);
    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
  }
  float4* dst = (float4*)(a.dW + (long)n * a.ldw + tl.wcol + kcol);
  if (a.overwrite) { *dst = s; return; }
  float4 o = *dst;
  o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
  *dst = o;
}

// out[g][n] (+)= sum over rows m of group g (rows_per_group consecutive rows) of Y[m][n]; fp32 atomics over row blocks
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ Y, int ldy, int N, int rows_per_group,
                                                     int rows_per_block, float* __restrict__ out, int ld_out) {
  const int g = blockIdx.z;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows_per_group, r0 + rows_per_block);
  const int col = (blockIdx.x * 32 + (threadIdx.x & 31)) * 8;      // 8 columns per thread, 8 row lanes per block
  if (col >= N) return;
  __shared__ float red[8][32][8];
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16_t* base = Y + (long)g * rows_per_group * ldy + col;
  int r = r0 + (threadIdx.x >> 5);
  for (; r + 56 < r1; r += 64) {          // eight independent 16-byte loads in flight per thread
    uint4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = *(const uint4*)(base + (long)(r + 8 * u) * ldy);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float f[8];
      unpack8(q[u], f);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += f[k];
    }
  }
  for (; r < r1; r += 8) {
    float f[8];
    unpack8(*(const uint4*)(base + (long)r * ldy), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += f[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x >> 5][threadIdx.x & 31][k] = s[k];
  __syncthreads();
  if (threadIdx.x < 32) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
      for (int j = 0; j < 8; ++j) t += red[j][threadIdx.x][k];
      atomicAdd(out + (long)g * ld_out + col + k, t);
    }
  }
}

}  // namespace

namespace dfh {

// Block slots of the chip for this kernel: 256 CUs x 2 resident workgroups.
constexpr int SLOTS = 512;

static int wgrad_plan(WgradArgs& a) {
  int chunks = a.ntaps * ((a.conv_c + TK - 1) / TK);
  for (int i = 0; i < a.nplain; ++i) chunks += (a.p_c[i] + TK - 1) / TK;
  const int ntn = (a.N + TN - 1) / TN;
  a.ktot = a.ntaps * a.conv_c;
  for (int i = 0; i < a.nplain; ++i) a.ktot += a.p_c[i];
  a.xblocks = ntn * chunks;
  a.whole = 0;
  const int blocks = a.xblocks;
  if (a.msplit < 0) {                                   // forced: whole tiles in full rounds, the remainder in -msplit slices
    a.msplit = -a.msplit; a.whole = blocks / SLOTS * SLOTS;
    if (a.whole == blocks) { a.whole = 0; a.msplit = 1; }
    return 0;
  }
  if (a.msplit > 0) return 0;
  // The plan is the cheapest candidate of a small cost model fitted to a sweep over the layer shapes of the step on the MI355X
  // (profiles/r05/wgrad_plan_sweep.txt; round 4: wgrad_msplit_sweep.txt): the chip holds 512 blocks at a time, so equal blocks run in
  // rounds of 512 -- a last round of <= 256 blocks has a CU to itself per block and takes 0.8 of a full one; a 32-row stage of a block
  // takes 0.85 us with the chip full (~0.98 PFLOP/s); a block costs 5 us of pipeline fill and epilogue; a slab slot 0.04 us (100 KB
  // written and read back).
  static const int mode = [] { const char* e = getenv("DFH_WGRAD_PLAN"); return e ? atoi(e) : 2; }();     // 1: round-4 candidates (1 or multiples of 8)
  constexpr double stage_us = 0.85, lone = 0.8, fill_us = 5.0, slot_us = 0.04;
  auto rounds_of = [&](long b) { const long full = b / SLOTS, rem = b % SLOTS; return full + (rem == 0 ? 0.0 : rem <= SLOTS / 2 ? lone : 1.0); };
  auto stages_of = [&](int ms) { return (double)(((a.M + ms - 1) / ms + BM - 1) / BM); };
  auto cost_slices = [&](int tiles, int ms) {
    const long b = (long)tiles * ms;
    return rounds_of(b) * (stages_of(ms) * stage_us + fill_us) + (ms > 1 ? (double)b * slot_us : 0.0);
  };
  int best = 1, best_whole = 0;
  double best_c = cost_slices(blocks, 1);
  for (int ms = (mode == 1 ? 8 : 2); ms <= 64 && a.M / ms >= 512; ms += (mode == 1 ? 8 : 1)) {
    const double c = cost_slices(blocks, ms);
    if (c < best_c) { best_c = c; best = ms; }
  }
  const int whole = blocks / SLOTS * SLOTS, tail = blocks - whole;
  if (mode >= 2 && whole > 0 && tail > 0) {
    const double cw = (whole / SLOTS) * (stages_of(1) * stage_us + fill_us);
    for (int ms = 2; ms <= 16 && a.M / ms >= 128; ++ms) {
      const double c = cw + cost_slices(tail, ms);
      if (c < best_c) { best_c = c; best = ms; best_whole = whole; }
    }
  }
  a.msplit = best; a.whole = best_whole;
  return 0;
}

static size_t wgrad_slots(const WgradArgs& a) { return a.msplit > 1 ? (size_t)a.msplit * (a.xblocks - a.whole) : 0; }

void wgrad_plan_only(WgradArgs& a) { wgrad_plan(a); }

size_t wgrad_partial_floats(WgradArgs a) {
  wgrad_plan(a);
  return wgrad_slots(a) * SLOT_FLOATS;
}

int wgrad_launch(WgradArgs a, hipStream_t s) {
  DFH_REQUIRE(a.M > 0 && a.N > 0 && a.N % 4 == 0 && a.ldy % 8 == 0 && a.ldy >= ((a.N + 7) & ~7),
              "wgrad: N must be a positive multiple of 4, dY rows padded to a multiple of 8 columns");
  DFH_REQUIRE(a.M < (1 << 24), "wgrad: M must be below 2^24");
  DFH_REQUIRE(a.ntaps == 0 || a.ntaps == 9 || (a.ntaps == 4 && a.tap2 && a.stride == 1 && a.ups == 0), "ntaps must be 0 or 9 (4: one phase plane of an upsample conv)");
  DFH_REQUIRE(a.ntaps + a.nplain >= 1 && a.zero && a.dY && a.dW, "wgrad: missing operand");
  DFH_REQUIRE(a.ldw % 4 == 0 && ((uintptr_t)a.dW & 15) == 0, "wgrad: dW rows must be 16-byte aligned");
  wgrad_plan(a);
  const size_t need = wgrad_slots(a) * SLOT_FLOATS;
  if (need && (a.partial == nullptr || a.partial_cap < need)) {
    set_error("wgrad: slab buffer missing or smaller than wgrad_partial_floats()");
    return -1;
  }
  constexpr int lds = NST * WSTAGE;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  double kreal = (double)a.ntaps * a.conv_c;
  for (int i = 0; i < a.nplain; ++i) kreal += a.p_c[i];
  ProfScope ps(PC_WGRAD, 2.0 * a.M * a.N * kreal, 2.0 * a.M * (a.N + kreal) + 4.0 * a.N * kreal, s);
  const bool pow2 = a.ntaps == 0 || (((a.Hout & (a.Hout - 1)) | (a.Wout & (a.Wout - 1))) == 0);
  const int grid = a.whole + (a.xblocks - a.whole) * a.msplit;
  if (pow2) hipLaunchKernelGGL(gemm_wgrad_kernel<true>, dim3(grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(gemm_wgrad_kernel<false>, dim3(grid), dim3(256), lds, s, a);
  if (int rc = check_launch("gemm_wgrad_kernel")) return rc;
  if (need) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a.xblocks - a.whole, 25), dim3(256), 0, s, a);
    return check_launch("wgrad_reduce_kernel");
  }
  return 0;
}

int colsum_launch(const bf16_t* Y, int ldy, int N, int groups, int rows_per_group, float* out, int ld_out, hipStream_t s) {
  DFH_REQUIRE(N % 8 == 0 && ldy % 8 == 0, "colsum: N and ldy must be multiples of 8");
  // about two blocks per CU: more row blocks only add same-address atomics (they serialise in L2)
  const int xb = (N / 8 + 31) / 32;
  const int want_y = std::max(1, 512 / (xb * groups));
  int rpb = std::max(64, (rows_per_group + want_y - 1) / want_y);
  rpb = (rpb + 63) / 64 * 64;
  const dim3 grid((N / 8 + 31) / 32, (rows_per_group + rpb - 1) / rpb, groups);
  ProfScope ps(PC_OTHER, 0.0, 2.0 * groups * rows_per_group * N, s);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, Y, ldy, N, rows_per_group, rpb, out, ld_out);
  return check_launch("colsum_kernel");
}

}  // namespace dfh
