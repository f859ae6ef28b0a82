// Persistent GEGLU projection (ff.net.0 of every BasicTransformerBlock: diffusers GEGLU.proj + gelu gate, reached from
// DiFashion/models/difashion.py:249-253,518-523): the 256 x 256 eight-wave tile of gemm.hip (`gemm_bf16_kernel<256,256,2,4,2,LEAN,WEPI>`,
// same staging, same fragment reads, same MFMA order, same in-register GEGLU arithmetic -> bit-identical results) as ONE workgroup per CU
// that walks its tiles and pipelines ACROSS them:
//
//   * K is short here (320 / 640 / 1280 = 5 / 10 / 20 k-steps) and every tile starts with a cold fetch of its first stage, then ends with
//     an epilogue during which nothing is in flight.  The persistent workgroup issues the NEXT tile's first stage (and requests that tile's
//     LayerNorm row statistics) at the top of the current tile's LAST k-step, into the ring buffer that k-step no longer reads; the
//     epilogue then runs while those lines arrive, and the next tile's k-loop starts on resident data.
//   * LDS: [stage A 64 KB][aux 4 KB][stage B 64 KB].  The epilogue stages the bf16 result through the buffer the last k-step read (two
//     passes of 128 rows: 35 KB), its bias / folded-LayerNorm vectors and row statistics through aux -- never through the buffer that is
//     receiving the next tile.
//   * vmcnt discipline: loads and stores retire out of order with respect to each other, so a wave never waits for LDS-DMA pieces with
//     stores behind them in its queue: the cross-tile stage is issued by waves 4-7 only (16 pieces each), the epilogue's global stores by
//     waves 0-3 only.  At the next tile's first k-step waves 4-7 wait for vmcnt(0) (nothing but that stage is outstanding for them), the
//     barrier publishes it to waves 0-3.
//   * tile order: virtual block ids blockIdx.x, + gridDim.x, ... through the same tile_coords() as the one-tile-per-workgroup launch, so an
//     XCD keeps the working set the launcher's tile order was chosen for.
#include "gemm.h"
#include "gemm_kiter.h"

#include <cstdio>
#include <cstdlib>

namespace {

template <int N> DFH_DEVICE void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NWV = 8;
constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
constexpr int AUX = 4096, AUX_BASE = STAGE, LDS_BYTES = 2 * STAGE + AUX;
constexpr int RSG = BN + 16;                       // bf16 row stride of the staged [128][BN / 2] half tile (bytes)
static_assert(128 * RSG <= STAGE && FN % 2 == 0 && BM <= NWV * 64, "geometry");

DFH_DEVICE int buf_base(int buf) { return buf ? STAGE + AUX : 0; }

// prof: diagnosis only (DFH_GEGLU_PROF=1): s_memtime stamps of workgroup 0 / thread 0, 8 per tile
__global__ __launch_bounds__(NWV * 64, 1) void gemm_geglu_rows_kernel(const GemmArgs a, unsigned long long* prof) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fg = lane >> 4;
  const int ntm = a.M / BM, ntn = a.N / BN, T = ntm * ntn;
  const int K = a.p_c[0], nk = K / BK;
  const bool lnf = a.ln_stat != nullptr;
  const int srow = lane >> 3;
  const int sslot = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));      // 16-byte slot swizzled on the source side (gemm.hip)
  const bf16_t* psrc0 = a.p_src[0];
  const bf16_t* Wb = a.W;
  asm volatile("" : "+s"(psrc0), "+s"(Wb));
  auto glds = [&](const bf16_t* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // staging pointers of the tile being walked: piece p = i * NWV + wave covers tile rows p * 8 + lane / 8
  const bf16_t* lp_a[4]; const bf16_t* lp_w[4];
  auto set_ptrs = [&](int m0, int n0, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (i * NWV + wave) * 8 + srow;
      lp_a[i] = psrc0 + ((size_t)(unsigned)(m0 + r) * (unsigned)K + (unsigned)(k0 + sslot * 8));
      lp_w[i] = Wb + ((size_t)(unsigned)(n0 + r) * (unsigned)a.ldw + (unsigned)(k0 + sslot * 8));
    }
  };
  auto issue = [&](int buf) {
    unsigned char* As = smem + buf_base(buf) + wave * 1024;
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) { glds(lp_a[i], As + i * NWV * 1024); lp_a[i] += BK; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { glds(lp_w[i], Bs + i * NWV * 1024); lp_w[i] += BK; }
  };
  // first stage of ANOTHER tile, by waves 4-7 only: piece p = 4 q + (wave - 4), q = 0 .. 7 (p & 1 == wave & 1: same source swizzle)
  auto issue_cross = [&](int buf, int m0, int n0) {
    unsigned char* As = smem + buf_base(buf);
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int p = 4 * q + (wave - 4), r = p * 8 + srow;
      glds(psrc0 + ((size_t)(unsigned)(m0 + r) * (unsigned)K + (unsigned)(sslot * 8)), As + p * 1024);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int p = 4 * q + (wave - 4), r = p * 8 + srow;
      glds(Wb + ((size_t)(unsigned)(n0 + r) * (unsigned)a.ldw + (unsigned)(sslot * 8)), Bs + p * 1024);
    }
  };

  int vb = blockIdx.x;
  if (vb >= T) return;
  int mt, nt;
  tile_coords(vb, ntm, ntn, a.n_major, a.tm_xm, a.tm_gm, mt, nt);
  int m0 = mt * BM, n0 = nt * BN;
  float2 lnmr = float2{0.f, 1.f};
  if (lnf && tid < BM) lnmr = ln_row_stats(a, m0 + tid);
  set_ptrs(m0, n0, 0);
  issue(0);
  int cb = 0;                                        // ring buffer holding stage 0 of the current tile
  bool first = true;

  int tile_no = 0;
  auto mark = [&](int i) {
    if (prof) {
      __builtin_amdgcn_sched_barrier(0);
      if (blockIdx.x == 0 && tid == 0 && tile_no < 16) prof[tile_no * 8 + i] = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (;;) {
    mark(0);
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int vbn = vb + gridDim.x;
    const bool has_next = vbn < T;                   // workgroup-uniform
    int m0n = 0, n0n = 0;
    float2 lnmr_n = float2{0.f, 1.f};
    float4 strip_v = float4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nk; ++t) {
      if (t == 0 && !first) { if (wave >= 4) wait_vmcnt<0>(); }      // waves 0-3 hold no LDS-DMA pieces of this stage (their stores may still fly)
      else wait_vmcnt<0>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // stage t visible to all waves; all waves done with k-step t - 1
      asm volatile("" ::: "memory");
      if (t == 0) mark(1);
      const int buf = (cb + t) & 1;
      if (t + 1 < nk) issue(buf ^ 1);
      else {
        // last k-step: this tile's epilogue vectors first (they must retire before anything that can miss), then the next tile
        if (tid < 64) { if (a.bias) strip_v = *(const float4*)(a.bias + n0 + tid * 4); }
        else if (tid < 128) { if (lnf) strip_v = *(const float4*)(a.ln_s + n0 + (tid - 64) * 4); }
        if (has_next) {
          tile_coords(vbn, ntm, ntn, a.n_major, a.tm_xm, a.tm_gm, mt, nt);
          m0n = mt * BM; n0n = nt * BN;
          if (lnf && tid < BM) lnmr_n = ln_row_stats(a, m0n + tid);
          __builtin_amdgcn_sched_barrier(0);         // the register loads above stay OLDER than the LDS-DMA pieces (they retire first)
          if (wave >= 4) issue_cross(buf ^ 1, m0n, n0n);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      const unsigned char* As = smem + buf_base(buf);
      const unsigned char* Bs = As + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm * TM + i * 16 + fr;
          af[i] = *(const bf16x8_t*)(As + row * 128 + (((ks * 4 + fg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wn * TN + j * 16 + fr;
          bfr[j] = *(const bf16x8_t*)(Bs + row * 128 + (((ks * 4 + fg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);     // weights as MFMA-A (gemm.hip)
      }
    }

    // ---------------------------------------------------------------- epilogue (arithmetic of gemm.hip's in-register GEGLU branch)
    mark(2);
    const int L = (cb + nk - 1) & 1;                 // the buffer the last k-step read: free from here on; L ^ 1 is receiving the next tile
    unsigned char* stg = smem + buf_base(L);
    __syncthreads();                                 // every wave is done reading the last stage
    mark(3);
    if (tid < 128) *(float4*)(smem + AUX_BASE + tid * 16) = strip_v;             // bias [256] | ln_s [256]
    if (lnf && tid < BM) *(float2*)(smem + AUX_BASE + 2048 + tid * 8) = lnmr;
    __syncthreads();
    const GeluK gk = gelu_consts();
    float4 bv[FN / 2], bg[FN / 2], sv[FN / 2], sg[FN / 2];
#pragma unroll
    for (int jj = 0; jj < FN / 2; ++jj) {
      const int col = wn * TN + jj * 32 + fg * 4;
      bv[jj] = *(const float4*)(smem + AUX_BASE + col * 4); bg[jj] = *(const float4*)(smem + AUX_BASE + (col + 16) * 4);
      sv[jj] = *(const float4*)(smem + AUX_BASE + 1024 + col * 4); sg[jj] = *(const float4*)(smem + AUX_BASE + 1024 + (col + 16) * 4);
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {                 // rows {wm * 128 + ph * 64 .. + 63}: every wave works in both passes
#pragma unroll
      for (int ii = 0; ii < FM / 2; ++ii) {
        const int i = ph * (FM / 2) + ii;
        const int row = wm * TM + i * 16 + fr, rr = wm * 64 + ii * 16 + fr;
        float2 mr = float2{0.f, 1.f};
        if (lnf) mr = *(const float2*)(smem + AUX_BASE + 2048 + row * 8);
        const float ms = -mr.x * mr.y;               // rstd * (acc - mean * s) + b' == rstd * acc + (b' - rstd * mean * s)
#pragma unroll
        for (int jj = 0; jj < FN / 2; ++jj) {
          const uint2 o = geglu4(acc[i][2 * jj], acc[i][2 * jj + 1], bv[jj], bg[jj], sv[jj], sg[jj], lnf, mr.y, ms, gk);
          const int ocl = ((wn * TN) >> 1) + jj * 16 + fg * 4;
          *(uint2*)(stg + rr * RSG + ocl * 2) = o;
        }
      }
      __syncthreads();
      mark(4 + ph);
      if (wave < 4) {                                // global stores by waves 0-3 only (see the vmcnt note at the top)
        constexpr int CPRG = BN / 16;                // 16-byte chunks per output row of the tile
        for (int c = tid; c < 128 * CPRG; c += 256) {
          const int rr = c / CPRG, cc = c - rr * CPRG;
          const int row = (rr >> 6) * TM + ph * 64 + (rr & 63);
          *(uint4*)((bf16_t*)a.out + (long)(m0 + row) * a.ld_out + (n0 >> 1) + cc * 8) = *(const uint4*)(stg + rr * RSG + cc * 16);
        }
      }
      if (ph == 0) __syncthreads();                  // the half tile is out of LDS before the second pass overwrites it
    }
    mark(6);
    ++tile_no;
    if (!has_next) break;
    vb = vbn; m0 = m0n; n0 = n0n; lnmr = lnmr_n; cb = L ^ 1; first = false;
    set_ptrs(m0, n0, BK);                            // stage 0 of this tile is already on its way
  }
}

}  // namespace

namespace dfh {

bool gemm_geglu_rows_ok(const GemmArgs& a) {
  // OPT-IN (DFH_GEGLU_ROWS=1): measured, not faster inside the step -- 171.8 / 141.1 / 135.8 us (one tile per workgroup) against 172.7 / 145.1 /
  // 144.0 us on the ff.net.0 launches of the 64x64 / 32x32 / 16x16 levels, step 16.60 vs 16.65 ms (profiles/r03/geglu_phases.txt: a tile
  // is 52 % k-loop at 3560 cycles per k-step -- the MFMA / ds_read co-issue bound, 64 MFMAs + 24 fragment reads per wave -- and 43 %
  // epilogue; the cross-tile fetch removes neither).  Kept as the measured answer to "pipeline across tiles", with its parity test.
  const char* e = getenv("DFH_GEGLU_ROWS");          // read per call (15 launches per step): tests switch it on for one case
  const bool off = !(e && e[0] == '1');
  if (off || a.act != ACT_GEGLU || a.out_mode != OUT_BF16 || a.resid || a.rowvec || a.out2 || a.nbatch > 1 || a.w_blocked) return false;
  if (a.ntaps != 0 || a.nplain != 1 || a.p_c[0] <= 0 || a.p_c[0] % BK != 0) return false;
  if (a.M % BM != 0 || a.N % BN != 0 || (a.ld_out & 7) != 0) return false;
  const long tiles = (long)(a.M / BM) * (a.N / BN);
  return tiles >= 512;                               // at least two tiles per workgroup: below that there is nothing to pipeline across
}

int gemm_geglu_rows_launch(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_geglu_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  static int ncu = 0;
  if (!ncu) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    ncu &= ~7;                                       // whole XCD rows: virtual block ids keep their XCD (tile_coords)
    if (ncu <= 0) ncu = 8;
  }
  census(CK_GEMM_ROWS_GEGLU);
  static const bool prof_on = [] { const char* e = getenv("DFH_GEGLU_PROF"); return e && e[0] == '1'; }();
  if (prof_on) {          // diagnosis: one launch with the phase stamps, printed as per-phase cycle averages over the tiles of workgroup 0
    static unsigned long long* buf = nullptr;
    if (!buf && hipMalloc((void**)&buf, 16 * 8 * 8) != hipSuccess) return -1;
    (void)hipMemsetAsync(buf, 0, 16 * 8 * 8, stream);
    hipLaunchKernelGGL(gemm_geglu_rows_kernel, dim3(ncu), dim3(NWV * 64), LDS_BYTES, stream, a, buf);
    unsigned long long h[16 * 8];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
    double ph[7] = {0, 0, 0, 0, 0, 0, 0}; int n = 0;
    for (int t = 1; t < 15; ++t) {
      if (!h[t * 8] || !h[t * 8 + 6] || !h[(t + 1) * 8]) continue;
      for (int i = 0; i < 6; ++i) ph[i] += (double)(h[t * 8 + i + 1] - h[t * 8 + i]);
      ph[6] += (double)(h[(t + 1) * 8] - h[t * 8 + 6]); ++n;
    }
    if (n) fprintf(stderr, "[geglu prof] M=%d N=%d K=%d, cycles per tile (thread 0 of workgroup 0, %d tiles): wait stage 0 %.0f | k-loop %.0f | "
                           "barrier %.0f | vectors + GELU + stage pass 0 %.0f | stores 0 + pass 1 %.0f | stores 1 %.0f | loop-back %.0f\n",
                   a.M, a.N, a.p_c[0], n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n, ph[6] / n);
    return check_launch("gemm_geglu_rows_kernel");
  }
  hipLaunchKernelGGL(gemm_geglu_rows_kernel, dim3(ncu), dim3(NWV * 64), LDS_BYTES, stream, a, (unsigned long long*)nullptr);
  return check_launch("gemm_geglu_rows_kernel");
}

}  // namespace dfh
