// Depthwise forward, eval mode: plain input, folded BN2 + ReLU6 epilogue.
#include "depthwise_common.h"
bool dw_launch_fwd_eval(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) { return dw_launch_specialised<DW_IN_NONE, DW_OUT_AFFINE_RELU6>(p, st, tl); }
