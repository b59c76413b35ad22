// Fused depthwise backward with the saved activations a2 (second input stream) and a1 (mask / xin) stored as bf16.
#include "depthwise_common.h"
bool dw_launch_bwd_fused16(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) {
    return dw_launch_specialised<DW_IN_AFFINE2, DW_OUT_MASK_STATS, true, DW_IO_X2 | DW_IO_AUX>(p, st, tl);
}
