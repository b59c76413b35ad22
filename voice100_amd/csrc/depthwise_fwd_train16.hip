// Depthwise forward, training mode, hidden activations stored as bf16 ("act16"): a1 in (BN1 affine + ReLU6 on load),
// a2 out, BN2 partial sums from the fp32 accumulators.
#include "depthwise_common.h"
#include "depthwise_stream16.h"

#ifndef DWS_DEPTH
#define DWS_DEPTH 1          /* rows of loads a wave keeps in flight beyond the current one: see profiles/r03_dw_stream_ab.txt */
#endif
#ifndef DWS_NT
#define DWS_NT 1
#endif

// rows that fit one tile take the streaming kernel (V100_DW_STREAM=0: the general kernel, for A/B runs)
static bool dws_enabled() {
    static const bool on = [] { const char* e = getenv("V100_DW_STREAM"); return !(e && e[0] == '0'); }();
    return on;
}

bool dw_launch_fwd_train16(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) {
    if (dws_enabled() && p.stride == 1 && p.upsample == 1 && !p.flip && p.Tin == p.Tout && p.Tin <= 512 &&
        p.pad == (p.K - 1) / 2) {
        const DwPathConfig cfg = dw_path_config();
        dim3 grid(p.C, p.G);
        // tuning switches (A/B runs only): V100_DW_STREAM_D = rows in flight (1, 2, 4), V100_DW_STREAM_NT = nontemporal rows
        static const int depth = [] { const char* e = getenv("V100_DW_STREAM_D"); return e ? atoi(e) : DWS_DEPTH; }();
        static const int ntp = [] { const char* e = getenv("V100_DW_STREAM_NT"); return e ? atoi(e) : DWS_NT; }();
#define GO(KK, DD, CPP) V100_LAUNCH(tl, (dwconv_fwd16_stream_kernel<KK, 2, DD, CPP>), grid, dim3(256), 0, st, p)
#define X(KK)                                                                                                             \
    if (p.K == KK) {                                                                                                      \
        if (cfg.digits3) V100_LAUNCH(tl, (dwconv_fwd16_stream_kernel<KK, 3, DWS_DEPTH, 0>), grid, dim3(256), 0, st, p);   \
        else if (depth == 1) { if (ntp) GO(KK, 1, 2); else GO(KK, 1, 0); }                                                \
        else if (depth == 2) { if (ntp) GO(KK, 2, 2); else GO(KK, 2, 0); }                                                \
        else { if (ntp) GO(KK, 4, 2); else GO(KK, 4, 0); }                                                                \
        return true;                                                                                                      \
    }
        V100_DW_SPECIALISED(X)
#undef X
#undef GO
    }
    return dw_launch_specialised<DW_IN_AFFINE_RELU6, DW_OUT_RAW_STATS, false, DW_IO_X | DW_IO_Y>(p, st, tl);
}
