// Fused depthwise backward, every hidden tensor stored as bf16: dz2 and a2 in, a1 for the mask / xin, dz1 out.
#include "depthwise_common.h"
#include "depthwise_stream16.h"

// (see depthwise_fwd_train16.hip; same A/B table)
#ifndef DWS_BWD_DEPTH
#define DWS_BWD_DEPTH 1
#endif
#ifndef DWS_NT
#define DWS_NT 1
#endif
#ifndef DWS_BWD_DEPTH3
#define DWS_BWD_DEPTH3 DWS_BWD_DEPTH     /* ... in the 768-position form (time-stretched rows) */
#endif
// rows of loads in flight per wave in the kept-rows (DA1) form.  Round 6, 8 layers on a rotating working set (profiles/r06_dw_ab.txt):
// 1 -> 2: 407 -> 366 us.  That form holds two workgroups per CU (its kept rows fill the register file), and unlike the plain form --
// which a depth of 2 never helped -- it has too few bytes in flight at one row per wave; the extra row costs no occupancy here.
#ifndef DWS_KEEP_DEPTH
#define DWS_KEEP_DEPTH 2
#endif

static bool dws_bwd_enabled() {
    static const bool on = [] { const char* e = getenv("V100_DW_STREAM_BWD"); return !(e && e[0] == '0'); }();
    return on;
}

// the kept-rows form of the streaming kernel (DA1, depthwise_stream16.h): one group, a wave's rows fit its 8 register slots
bool dw_bwd_da1_supported(int B, int C, int T, int K, int G) {
    static const bool on = [] { const char* e = getenv("V100_IR_DA1"); return !(e && e[0] == '0'); }();     // A/B switch
    // Rows of 513 .. 768 outputs (time-stretched steps) cannot keep their rows (12 registers a row: the file is full); their read-back
    // form exists (depthwise_stream16.h, V100_IR_DA1_TMAX=768) and was measured in round 6 on a step with every batch stretched to 110 %:
    // the two expand GEMMs gain 0.19 ms (1.704 + 0.822 -> 1.580 + 0.756) and this kernel loses 0.19 (0.416 -> 0.610): no net gain, so
    // those rows keep dz1 + the consumers' transform on load.
    static const int tmax = [] { const char* e = getenv("V100_IR_DA1_TMAX"); return e ? atoi(e) : 512; }();
    if (!on || !dws_bwd_enabled() || G != 1 || B > 32 || T > tmax || T > 768 || T < 1 || C < 1) return false;
#define X(KK) if (K == KK) return true;
    V100_DW_SPECIALISED(X)
#undef X
    return false;
}

bool dw_launch_bwd_fused16g(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) {
    // rows of up to 768 outputs: the streaming kernel (V100_DW_STREAM_BWD=0: the general kernel, for A/B runs)
    if (dws_bwd_enabled() && p.stride == 1 && p.upsample == 1 && p.flip && p.Tin == p.Tout && p.Tin <= 768 &&
        p.pad == p.K - 1 - (p.K - 1) / 2) {
        const DwPathConfig cfg = dw_path_config();
        dim3 grid(p.C, p.G);
#define GO(KK, NTT)                                                                                                               \
    do {                                                                                                                          \
        if (p.da1) {                                                                                                              \
            if (p.fin.mode != 2 || p.G != 1 || p.B > 32) return false;                                                            \
            if (p.Tin > 512) V100_LAUNCH(tl, (dwconv_bwd16_stream_kernel<KK, NTT, DWS_BWD_DEPTH3, DWS_NT * 2, 3, true>), grid, dim3(256), 0, st, p);  \
            else V100_LAUNCH(tl, (dwconv_bwd16_stream_kernel<KK, NTT, DWS_KEEP_DEPTH, DWS_NT * 2, 2, true>), grid, dim3(256), 0, st, p);          \
        } else if (p.Tin <= 512) V100_LAUNCH(tl, (dwconv_bwd16_stream_kernel<KK, NTT, DWS_BWD_DEPTH, DWS_NT * 2, 2>), grid, dim3(256), 0, st, p);  \
        else V100_LAUNCH(tl, (dwconv_bwd16_stream_kernel<KK, NTT, DWS_BWD_DEPTH3, DWS_NT * 2, 3>), grid, dim3(256), 0, st, p);     \
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
    if (p.cm || p.da1) return false;      // the general kernel addresses [B][C][P] only (and has no kept-rows form)
    return dw_launch_specialised<DW_IN_AFFINE2, DW_OUT_MASK_STATS, true, DW_IO_X | DW_IO_X2 | DW_IO_AUX | DW_IO_Y>(p, st, tl);
}
