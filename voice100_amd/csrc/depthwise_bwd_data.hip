// Depthwise backward-data: BN2-backward affine of (dz2, a2) on load, flipped taps, ReLU6 mask from a1,
// BN1-backward partial sums.
#include "depthwise_common.h"
bool dw_launch_bwd_data(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) { return dw_launch_specialised<DW_IN_AFFINE2, DW_OUT_MASK_STATS>(p, st, tl); }
