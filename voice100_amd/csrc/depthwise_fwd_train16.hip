// Depthwise forward, training mode, hidden activations stored as bf16 ("act16"): a1 in (BN1 affine + ReLU6 on load),
// a2 out, BN2 partial sums from the fp32 accumulators.
#include "depthwise_common.h"
bool dw_launch_fwd_train16(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) {
    return dw_launch_specialised<DW_IN_AFFINE_RELU6, DW_OUT_RAW_STATS, false, DW_IO_X | DW_IO_Y>(p, st, tl);
}
