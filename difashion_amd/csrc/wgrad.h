#pragma once
#include "dfh_common.h"
#include <algorithm>

struct WgradArgs {
  // forward A operand, same K-segment description as GemmArgs
  const bf16_t* conv_src; int conv_c; int ntaps;
  int Hin, Win, Hout, Wout, stride, ups;
  const bf16_t* p_src[2]; int p_c[2]; int nplain;
  const bf16_t* dY; int ldy;     // output gradient rows [M][ldy] (N columns used)
  const bf16_t* zero;
  int M, N;
  float* dW; int ldw;            // fp32 [N][ldw] in the PACKED weight layout, accumulated (+=)
  float* partial; size_t partial_cap;   // fp32 slab slots of 160 x 160 floats, one per partial piece (wgrad_partial_floats)
  int ktot;                      // filled by the launcher: total K of the call
  int overwrite;                 // 1: dW = ... instead of += (the caller guarantees this launch is the only writer)
  float* dbias;                  // optional fp32 [N]: += column sums of dY (bias gradient), fused into the same pass
  int msplit;                    // 0 = heuristic; n > 0: n equal pixel slices of every tile; -n: whole tiles in full rounds of 512 blocks, the rest in n slices
  int whole;                     // filled by the launcher: the first `whole` tiles are not sliced (wgrad.hip)
  int xblocks;                   // filled by the launcher: n-tiles x kcol-chunks
  // one PHASE PLANE of a nearest-2x upsample + 3x3 conv (gemm.h GemmArgs::phase2x): ntaps = 4, stride 1 over the SOURCE image, segment s =
  // the 3x3-tap position (tap_py + (s >> 1), tap_px + (s & 1)); dY = the plane's rows of the output gradient (gathered phase-major);
  // dW = the gradient of the plane's SUMMED weights [N][4 * conv_c] (un-folded onto the 3x3 taps by ups_phase_unfold)
  int tap2, tap_py, tap_px;
};

namespace dfh {
int wgrad_launch(WgradArgs a, hipStream_t s);
void wgrad_plan_only(WgradArgs& a);           // fills xblocks / ktot / whole / msplit as wgrad_launch would
size_t wgrad_partial_floats(WgradArgs a);   // slab floats the heuristic (or a.msplit) needs; 0 = none
// out[g][n] += sum_{m in group g} Y[m][n]   (bias gradient: groups = 1; time-embedding gradient: groups = batch)
int colsum_launch(const bf16_t* Y, int ldy, int N, int groups, int rows_per_group, float* out, int ld_out, hipStream_t s);
}  // namespace dfh
