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

constexpr int WBM = 64;     // m rows per pipeline stage (contraction chunk)
constexpr int WBN = 128;    // n (output channel) tile
constexpr int WBK = 64;     // kcol tile = one 64-wide slice of one K segment
constexpr int Y_BYTES = WBM * WBN * 2, A_BYTES = WBM * WBK * 2, WSTAGE = Y_BYTES + A_BYTES;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

DFH_DEVICE bf16x8_t tr_frag(const unsigned char* tile, int row_bytes, int m0, int col0, int L) {
  // 8 consecutive m (rows m0..m0+7) of column col0 + L, as one MFMA operand fragment
  const unsigned char* p = tile + (m0 + (L >> 2)) * row_bytes + (col0 + (L & 3) * 4) * 2;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 4 * row_bytes));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = lane & 15, fg = lane >> 4;

  const int ntn = (a.N + WBN - 1) / WBN;
  const int n0 = (blockIdx.x % ntn) * WBN;
  const int chunk = blockIdx.x / ntn;
  // locate the chunk: (segment, channel offset, packed column)
  int seg = 0, base = 0, kc = chunk, seglen = 0;
  const int nseg = a.ntaps + a.nplain;
  for (;;) {
    seglen = seg < a.ntaps ? a.conv_c : a.p_c[seg - a.ntaps];
    const int n = (seglen + WBK - 1) / WBK;
    if (kc < n || seg == nseg - 1) break;
    kc -= n; base += seglen; ++seg;
  }
  const int c0 = kc * WBK, wcol = base + c0;
  const bool conv = seg < a.ntaps;
  const int ky = conv ? seg / 3 : 0, kx = conv ? seg - ky * 3 : 0;

  const int m_per = (((a.M + a.msplit - 1) / a.msplit) + WBM - 1) / WBM * WBM;
  const int m_begin = blockIdx.z * m_per, m_end = min(a.M, m_begin + m_per);
  const int nsteps = m_end > m_begin ? (m_end - m_begin + WBM - 1) / WBM : 0;

  const int HWo = a.Hout * a.Wout;
  const int Hv = a.ups ? a.Hin * 2 : a.Hin, Wv = a.ups ? a.Win * 2 : a.Win;
  // staging roles: dY pieces = 4 rows x 256 B (lane -> row lane/16, slot lane%16), 16 pieces per stage;
  //                A  pieces = 8 rows x 128 B (lane -> row lane/8,  slot lane%8),   8 pieces per stage
  auto glds = [&](const bf16_t* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue = [&](int step, int buf) {
    unsigned char* Ys = smem + buf * WSTAGE;
    unsigned char* As = Ys + Y_BYTES;
    const int mb = m_begin + step * WBM;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = i * 4 + wave;
      const int m = mb + p * 4 + (lane >> 4);
      const int n = n0 + (lane & 15) * 8;
      const bool ok = (m < m_end) & (n < a.N);
      glds(ok ? a.dY + ((long)m * a.ldy + n) : a.zero, Ys + p * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = i * 4 + wave;
      const int m = mb + p * 8 + (lane >> 3);
      const int ch = c0 + (lane & 7) * 8;
      bool ok = (m < m_end) & (ch < seglen);
      const bf16_t* src = a.zero;
      if (conv) {
        const int mm = min(m, a.M - 1);
        const int b = mm / HWo, rem = mm - b * HWo;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        const int yy = oy * a.stride - 1 + ky, xx = ox * a.stride - 1 + kx;
        ok = ok & ((unsigned)yy < (unsigned)Hv) & ((unsigned)xx < (unsigned)Wv);
        const int sy = a.ups ? (yy >> 1) : yy, sx = a.ups ? (xx >> 1) : xx;
        if (ok) src = a.conv_src + ((long)((b * a.Hin + sy) * a.Win + sx) * a.conv_c + ch);
      } else {
        const int ps = seg - a.ntaps;
        if (ok) src = a.p_src[ps] + ((long)m * a.p_c[ps] + ch);
      }
      glds(src, As + p * 1024);
    }
  };

  f32x4_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  if (nsteps > 0) {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
      const int cur = t & 1;
      if (t + 1 < nsteps) issue(t + 1, cur ^ 1);
      const unsigned char* Ys = smem + cur * WSTAGE;
      const unsigned char* As = Ys + Y_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int m0 = ks * 32 + fg * 8;
        bf16x8_t yf[2], af[4];
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) yf[nf] = tr_frag(Ys, WBN * 2, m0, wave * 32 + nf * 16, L);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) af[kf] = tr_frag(As, WBK * 2, m0, kf * 16, L);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
            // D^T[row = kcol (fg*4+r)][col = n (L)]
            acc[kf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kf], yf[nf], acc[kf][nf], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    const int n = n0 + wave * 32 + nf * 16 + L;
    if (n >= a.N) continue;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      const int kcol = kf * 16 + fg * 4;
      float* dst = a.dW + (long)n * a.ldw + wcol + kcol;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (c0 + kcol + r < seglen) atomicAdd(dst + r, acc[kf][nf][r]);
    }
  }
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
  for (int r = r0 + (threadIdx.x >> 5); r < r1; r += 8) {
    float f[8];
    unpack8(*(const uint4*)(Y + ((long)g * rows_per_group + r) * ldy + col), f);
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

int wgrad_launch(WgradArgs a, hipStream_t s) {
  DFH_REQUIRE(a.M > 0 && a.N > 0 && a.N % 4 == 0 && a.ldy % 8 == 0 && a.ldy >= ((a.N + 7) & ~7),
              "wgrad: N must be a positive multiple of 4, dY rows padded to a multiple of 8 columns");
  DFH_REQUIRE(a.ntaps == 0 || a.ntaps == 9, "ntaps must be 0 or 9");
  DFH_REQUIRE(a.ntaps + a.nplain >= 1 && a.zero && a.dY && a.dW, "wgrad: missing operand");
  int chunks = a.ntaps * ((a.conv_c + WBK - 1) / WBK);
  for (int i = 0; i < a.nplain; ++i) chunks += (a.p_c[i] + WBK - 1) / WBK;
  const int ntn = (a.N + WBN - 1) / WBN;
  if (a.msplit <= 0) {
    const int blocks = ntn * chunks;
    a.msplit = std::max(1, std::min((512 + blocks - 1) / blocks, (a.M + 4 * WBM - 1) / (4 * WBM)));
  }
  constexpr int lds = 2 * WSTAGE;
  double kreal = (double)a.ntaps * a.conv_c;
  for (int i = 0; i < a.nplain; ++i) kreal += a.p_c[i];
  ProfScope ps(PC_WGRAD, 2.0 * a.M * a.N * kreal, 2.0 * a.M * (a.N + kreal) + 4.0 * a.N * kreal, s);
  hipLaunchKernelGGL(gemm_wgrad_kernel, dim3(ntn * chunks, 1, a.msplit), dim3(256), lds, s, a);
  return check_launch("gemm_wgrad_kernel");
}

int colsum_launch(const bf16_t* Y, int ldy, int N, int groups, int rows_per_group, float* out, int ld_out, hipStream_t s) {
  DFH_REQUIRE(N % 8 == 0 && ldy % 8 == 0, "colsum: N and ldy must be multiples of 8");
  int rpb = std::max(64, (rows_per_group + 63) / 64);     // <= 64 row blocks per group
  rpb = (rpb + 7) / 8 * 8;
  const dim3 grid((N / 8 + 31) / 32, (rows_per_group + rpb - 1) / rpb, groups);
  ProfScope ps(PC_OTHER, 0.0, 2.0 * groups * rows_per_group * N, s);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, Y, ldy, N, rows_per_group, rpb, out, ld_out);
  return check_launch("colsum_kernel");
}

}  // namespace dfh
