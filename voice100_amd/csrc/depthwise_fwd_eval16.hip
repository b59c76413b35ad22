// Depthwise forward, EVAL mode (inference), hidden activations stored as bf16: h1 in (already activated), folded BatchNorm-2 +
// ReLU6 on the way out (asr.py:27-37, 49 with frozen statistics).  Rows of up to 768 outputs: the streaming kernel; longer: the
// general MFMA kernel.
#include "depthwise_common.h"
#include "depthwise_stream16.h"

#ifndef DWS_DEPTH
#define DWS_DEPTH 1
#endif
#ifndef DWS_NT
#define DWS_NT 1
#endif

// f16: the tensors hold fp16 (streaming kernel only: rows of up to 768 outputs); p.cm: channel-major storage (streaming kernel only)
bool dw_launch_fwd_eval16(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl, bool f16) {
    if (f16) {
        if (!(p.stride == 1 && p.upsample == 1 && !p.flip && p.Tin == p.Tout && p.Tin <= 768 && p.pad == (p.K - 1) / 2)) return false;
        dim3 grid(p.C, p.G);
        // two fp16 digits per tap: 22 mantissa bits, far below the fp16 rounding of the stored activations
#define X(KK)                                                                                                                           \
    if (p.K == KK) {                                                                                                                    \
        if (p.Tin <= 512) V100_LAUNCH(tl, (dwconv_fwd16_stream_kernel<KK, DW_DIGITS16, DWS_DEPTH, DWS_NT * 2, 2, true, true>), grid, dim3(256), 0, st, p);  \
        else V100_LAUNCH(tl, (dwconv_fwd16_stream_kernel<KK, DW_DIGITS16, DWS_DEPTH, DWS_NT * 2, 3, true, true>), grid, dim3(256), 0, st, p);         \
        return true;                                                                                                                    \
    }
        V100_DW_SPECIALISED(X)
#undef X
        return false;
    }
    if (p.stride == 1 && p.upsample == 1 && !p.flip && p.Tin == p.Tout && p.Tin <= 768 && p.pad == (p.K - 1) / 2) {
        const DwPathConfig cfg = dw_path_config();
        dim3 grid(p.C, p.G);
#define GO(KK, NTT)                                                                                                                     \
    do {                                                                                                                                \
        if (p.Tin <= 512) V100_LAUNCH(tl, (dwconv_fwd16_stream_kernel<KK, NTT, DWS_DEPTH, DWS_NT * 2, 2, true>), grid, dim3(256), 0, st, p);  \
        else V100_LAUNCH(tl, (dwconv_fwd16_stream_kernel<KK, NTT, DWS_DEPTH, DWS_NT * 2, 3, true>), grid, dim3(256), 0, st, p);         \
    } while (0)
#define X(KK)                                                                                                                           \
    if (p.K == KK) {                                                                                                                    \
        if (cfg.digits3) GO(KK, 3); else GO(KK, DW_DIGITS16);                                                                                     \
        return true;                                                                                                                    \
    }
        V100_DW_SPECIALISED(X)
#undef X
#undef GO
    }
    if (p.cm) return false;               // the general kernel addresses [B][C][P] only
    return dw_launch_specialised<DW_IN_NONE, DW_OUT_AFFINE_RELU6, false, DW_IO_X | DW_IO_Y>(p, st, tl);
}
