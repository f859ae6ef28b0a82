// K-segment iterator of the 64-deep implicit-GEMM kernels (gemm.hip, gemm_ws.hip): which 64-element slice of which K segment
// (conv tap / plain operand) a k-step covers.  See gemm.h for the segment model.
#pragma once
#include "gemm.h"

namespace {

constexpr int BK = 64;            // k-step depth (bf16 elements) = one 128-byte LDS row

struct KIter {                    // which 64-deep slice of which K segment a k-step covers
  int seg;                        // 0..ntaps-1 conv taps, then plain segments
  int c0;                         // channel offset inside the segment
  int wcol;                       // column of W where this slice starts
  int seglen;                     // channels in the current segment
};

DFH_DEVICE int cdiv64(int x) { return (x + BK - 1) / BK; }

DFH_DEVICE int seg_len(const GemmArgs& a, int seg) {
  // selects, not a[] indexing: a runtime index into the kernel arguments becomes an s_load + lgkmcnt(0) stall in the k-loop
  return seg < a.ntaps ? a.conv_c : (seg == a.ntaps ? a.p_c[0] : a.p_c[1]);
}

// K is walked CHANNEL-CHUNK-major over the conv taps: the nine taps of one 64-channel slice are consecutive k-steps, so
// the shifted re-reads of a pixel tile (eight of nine taps touch rows the previous taps already fetched) come back within
// ~1 MB of L2 traffic per XCD instead of after the whole tile x all channels (10 MB at the 64x64 level: they missed).
// W columns stay tap-major (column = tap * Cin + c): any k order works as long as A slice and W column agree.
DFH_DEVICE KIter kiter_at(const GemmArgs& a, int kstep) {
  KIter it;
  const int conv_steps = a.ntaps * cdiv64(a.conv_c);
  if (kstep < conv_steps) {
    const int cc = kstep / a.ntaps;
    it.seg = kstep - cc * a.ntaps; it.c0 = cc * BK; it.seglen = a.conv_c; it.wcol = it.seg * a.conv_c + it.c0;
    return it;
  }
  kstep -= conv_steps;
  int seg = a.ntaps, base = a.ntaps * a.conv_c;
  const int nseg = a.ntaps + a.nplain;
  for (;;) {
    const int len = seg_len(a, seg);
    const int n = cdiv64(len);
    if (kstep < n || seg == nseg - 1) { it.seglen = len; break; }
    kstep -= n; base += len; ++seg;
  }
  it.seg = seg; it.c0 = kstep * BK; it.wcol = base + it.c0;
  return it;
}

DFH_DEVICE void kiter_next(const GemmArgs& a, KIter& it) {
  if (it.seg < a.ntaps) {                  // inside the conv part: next tap of the same channel slice
    ++it.seg; it.wcol += a.conv_c;
    if (it.seg < a.ntaps) return;
    it.seg = 0; it.c0 += BK; it.wcol = it.c0;
    if (it.c0 < a.conv_c) return;
    it.seg = a.ntaps; it.c0 = 0; it.wcol = a.ntaps * a.conv_c;     // conv part done: first plain segment
    if (a.nplain > 0) it.seglen = seg_len(a, it.seg);
    return;
  }
  it.c0 += BK; it.wcol += BK;
  if (it.c0 >= it.seglen) {
    it.wcol -= it.c0 - it.seglen;        // next segment starts right after this one's real length
    it.c0 = 0; ++it.seg;
    if (it.seg < a.ntaps + a.nplain) it.seglen = seg_len(a, it.seg);
  }
}

}  // namespace
