// Depthwise forward, training mode, hidden activations stored as bf16 ("act16"): a1 in (BN1 affine + ReLU6 on load),
// a2 out, BN2 partial sums from the fp32 accumulators.
#include "depthwise_common.h"
#include "depthwise_stream16.h"

// rows in flight beyond the current one / nontemporal row traffic: the A/B matrix is profiles/r03_dw_stream_ab.txt (D = 1 with NT wins
// in every regime; D = 2 ties on a rotating working set and loses behind the producing GEMM; D = 4 loses everywhere)
#ifndef DWS_DEPTH
#define DWS_DEPTH 1
#endif
#ifndef DWS_DEPTH3
#define DWS_DEPTH3 DWS_DEPTH      /* rows of loads in flight per wave in the 768-position form (time-stretched rows of 513 .. 768 outputs) */
#endif
#ifndef DWS_NT
#define DWS_NT 1
#endif

// rows of up to 768 outputs take the streaming kernels (V100_DW_STREAM=0: the general kernel, for A/B runs)
static bool dws_enabled() {
    static const bool on = [] { const char* e = getenv("V100_DW_STREAM"); return !(e && e[0] == '0'); }();
    return on;
}

bool dw_launch_fwd_train16(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) {
    if (dws_enabled() && p.stride == 1 && p.upsample == 1 && !p.flip && p.Tin == p.Tout && p.Tin <= 768 && p.pad == (p.K - 1) / 2) {
        const DwPathConfig cfg = dw_path_config();
        dim3 grid(p.C, p.G);
#define GO(KK, NTT)                                                                                                               \
    do {                                                                                                                          \
        if (p.Tin <= 512) V100_LAUNCH(tl, (dwconv_fwd16_stream_kernel<KK, NTT, DWS_DEPTH, DWS_NT * 2, 2>), grid, dim3(256), 0, st, p);  \
        else V100_LAUNCH(tl, (dwconv_fwd16_stream_kernel<KK, NTT, DWS_DEPTH3, DWS_NT * 2, 3>), grid, dim3(256), 0, st, p);         \
    } while (0)
#define X(KK)                                                                                                                     \
    if (p.K == KK) {                                                                                                              \
        if (cfg.digits3) GO(KK, 3); else GO(KK, DW_DIGITS16);                                                                               \
        return true;                                                                                                              \
    }
        V100_DW_SPECIALISED(X)
#undef X
#undef GO
    }
    if (p.cm) return false;               // the general kernel addresses [B][C][P] only
    return dw_launch_specialised<DW_IN_AFFINE_RELU6, DW_OUT_RAW_STATS, false, DW_IO_X | DW_IO_Y>(p, st, tl);
}
