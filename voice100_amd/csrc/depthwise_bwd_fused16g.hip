// Fused depthwise backward, every hidden tensor stored as bf16: dz2 and a2 in, a1 for the mask / xin, dz1 out.
#include "depthwise_common.h"
#include "depthwise_stream16.h"

#ifndef DWS_BWD_DEPTH
#define DWS_BWD_DEPTH 1          /* rows (of three input streams) a wave keeps in flight */
#endif
#ifndef DWS_NT
#define DWS_NT 1
#endif

static bool dws_bwd_enabled() {
    static const bool on = [] { const char* e = getenv("V100_DW_STREAM_BWD"); return !(e && e[0] == '0'); }();
    return on;
}

bool dw_launch_bwd_fused16g(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) {
    // rows that fit one tile: the streaming kernel (V100_DW_STREAM_BWD=0: the general kernel, for A/B runs)
    if (dws_bwd_enabled() && p.stride == 1 && p.upsample == 1 && p.flip && p.Tin == p.Tout && p.Tin <= 512 &&
        p.pad == p.K - 1 - (p.K - 1) / 2) {
        const DwPathConfig cfg = dw_path_config();
        dim3 grid(p.C, p.G);
        static const int depth = [] { const char* e = getenv("V100_DW_STREAM_BWD_D"); return e ? atoi(e) : DWS_BWD_DEPTH; }();
        static const int ntp = [] { const char* e = getenv("V100_DW_STREAM_NT"); return e ? atoi(e) : DWS_NT; }();
#define GO(KK, DD, CPP) V100_LAUNCH(tl, (dwconv_bwd16_stream_kernel<KK, 2, DD, CPP>), grid, dim3(256), 0, st, p)
#define X(KK)                                                                                                                  \
    if (p.K == KK) {                                                                                                           \
        if (cfg.digits3) V100_LAUNCH(tl, (dwconv_bwd16_stream_kernel<KK, 3, DWS_BWD_DEPTH, 0>), grid, dim3(256), 0, st, p);    \
        else if (depth == 1) { if (ntp) GO(KK, 1, 2); else GO(KK, 1, 0); }                                                     \
        else if (depth == 2) { if (ntp) GO(KK, 2, 2); else GO(KK, 2, 0); }                                                     \
        else { if (ntp) GO(KK, 3, 2); else GO(KK, 3, 0); }                                                                     \
        return true;                                                                                                           \
    }
        V100_DW_SPECIALISED(X)
#undef X
#undef GO
    }
    return dw_launch_specialised<DW_IN_AFFINE2, DW_OUT_MASK_STATS, true, DW_IO_X | DW_IO_X2 | DW_IO_AUX | DW_IO_Y>(p, st, tl);
}
