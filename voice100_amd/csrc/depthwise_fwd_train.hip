// Depthwise forward, training mode: BN1 affine + ReLU6 on load, raw output + BN2 partial sums.
#include "depthwise_common.h"
bool dw_launch_fwd_train(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) { return dw_launch_specialised<DW_IN_AFFINE_RELU6, DW_OUT_RAW_STATS>(p, st, tl); }
