// Depthwise backward, stride 1, one pass: BN2-backward affine of (dz2, a2) on load, ReLU6 mask of a1 + BN1-backward
// partial sums on the way out, and the backward-weight partial sums from the same window walk.
#include "depthwise_common.h"
bool dw_launch_bwd_fused(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) { return dw_launch_specialised<DW_IN_AFFINE2, DW_OUT_MASK_STATS, true>(p, st, tl); }
