// Fused depthwise backward, every hidden tensor stored as bf16: dz2 and a2 in, a1 for the mask / xin, dz1 out.
#include "depthwise_common.h"
bool dw_launch_bwd_fused16g(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) {
    return dw_launch_specialised<DW_IN_AFFINE2, DW_OUT_MASK_STATS, true, DW_IO_X | DW_IO_X2 | DW_IO_AUX | DW_IO_Y>(p, st, tl);
}
