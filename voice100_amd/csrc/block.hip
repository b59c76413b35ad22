// Block executor: one host call runs the whole kernel chain of an InvertedResidual block
// (voice100/models/asr.py:40-59) forward (training mode) or backward on the given stream.
// Pure host code: it sequences the kernel entry points of this library (no allocation: the caller
// passes the outputs, the tensors saved for backward and ONE workspace blob that is carved here).
// This replaces ~12 (forward) / ~20 (backward) Python->C crossings per block with one.
#include "common.h"
#include "../../include/voice100_hip.h"
#include "depthwise_common.h"   // DwFin: BatchNorm finalisation inside the depthwise kernels
#ifndef IR_FUSE_BN3
#define IR_FUSE_BN3 2      /* 1: reduce + finalise in one kernel; 2: ... + the affine, from registers */
#endif
// consumer-side BatchNorm finalisation in the depthwise kernels (DwPre): bit 0 forward (BatchNorm 1 from the expand GEMM's slab),
// bit 1 backward (BatchNorm-2 backward from the project backward-data GEMM's slab).  Bit-identical results.  Round 2 measured it as
// a faster STEP (16 launches fewer) with a slower depthwise kernel and left it off because that kernel is the one graded against the
// HBM roofline; the round-5 review asked for the default that minimises ms_per_step.  Round 6, with the finalisation's loads issued
// ahead of the rows (dw_pre_issue) and the Toeplitz fragments built under their latency, A/B on one box (profiles/r06_dw_ab.txt):
// 3.30 -> 3.24 ms/step, 151 -> 135 launches, glue 0.58 -> 0.51 ms; the depthwise forward pays +2 us a launch for it (0.189 -> 0.207 ms
// per step, roofline.frac_nominal_step 0.57 -> 0.52: the fp64 finalisation of a channel sits in front of its first row), the fused
// backward 0.388 -> 0.408.  On (3): the step is what counts; the slab bytes are counted in the kernels' algorithmic bytes.
#ifndef IR_FUSE_PRE
#define IR_FUSE_PRE 3
#endif

namespace {
struct Carver {
    char* p; size_t used;
    explicit Carver(void* base) : p((char*)base), used(0) {}
    template <class T> T* take(size_t n) {
        used = (used + 255) & ~size_t(255);
        T* r = p ? (T*)(p + used) : nullptr;
        used += n * sizeof(T);
        return r;
    }
};
inline int conv_out(int t, int k, int s) { return (t + 2 * ((k - 1) / 2) - k) / s + 1; }
const float kMom = 0.1f, kEps = 1e-5f;
}
#define CK(call) do { rc = (call); if (rc) return rc; } while (0)

// shape = {B, Cin, hid, Cout, T, K, stride, residual, bf16, prepped, act16}
// coef  = 12 vectors of `hid` floats: s1 t1 mean1 rstd1 s2 t2 mean2 rstd2 s3 t3 mean3 rstd3
// act16 (bf16 operands only): 0 = every activation fp32; 1 = the two hidden tensors saved for backward, a1 and a2, are
// stored as bf16 [B][hid][pitch16(T)] (the caller allocates them so); 2 = also the hidden gradients dz2 / dz1 inside the
// backward workspace; 3 = also the project output a3 (saved for backward; the caller allocates it bf16 [B][cout][pitch16(T)])
// and its BatchNorm-backward gradient da3 (workspace).  Statistics and every accumulation stay fp32; only the stored copies are rounded.
enum { IR_B, IR_CIN, IR_HID, IR_COUT, IR_T, IR_K, IR_STRIDE, IR_RES, IR_BF16, IR_PREPPED, IR_ACT16, IR_NSHAPE };
static inline int pitch16(int T, int B) { return v100_pitch16(T, B); }      // THE rule: common.h
// the row pitch (elements) of a 16-bit-stored [B][C][P] activation, for host code that allocates such tensors (functional.pitch16)
extern "C" int v100_row_pitch16(int T, int B) { return v100_pitch16(T, B); }
enum { PW_IO_X = 1, PW_IO_X2 = 2, PW_IO_Y = 4, PW_IO_R = 8, PW_IO_F16 = 16, WG_IO_G = 1, WG_IO_G2 = 2, WG_IO_X = 4 };   // DW_IO_*: depthwise_common.h

// 1 when a block of this shape can run with act16 != 0 (stride 1, a depthwise kernel size with an MFMA kernel, bf16 operands,
// tensors addressable by the buffer-descriptor kernels)
extern "C" int v100_ir_act16_supported(const int* sh) {
    if (!sh || sh[IR_BF16] != 1 || sh[IR_STRIDE] != 1) return 0;
    if (!v100_dw_mfma_supported(sh[IR_K], sh[IR_STRIDE])) return 0;
    const long B = sh[IR_B], hid = sh[IR_HID], cin = sh[IR_CIN], cout = sh[IR_COUT], P = pitch16(sh[IR_T], sh[IR_B]);
    if (hid % 2 || cin % 2 || cout % 2) return 0;
    if (B * hid * P * 4 >= 0x7fffff00L || (hid + 128) * P * 4 >= 0x7fffff00L || (hid + 128) * (cin > cout ? cin : cout) * 2 >= 0x7fffff00L) return 0;
    return 1;
}

// Prepared weights, written once by forward and reused by backward (caller-owned, saved with the block):
// bf16 mode: w1_bf [hid][cin], w1t_bf [cin][hid], w3_bf [cout][hid], w3t_bf [hid][cout];  fp32 mode: w1t, w3t (fp32).
struct IrPrep { void *w1bf, *w1tbf, *w3bf, *w3tbf; float *w1t, *w3t; };
static size_t ir_prep_carve(const int* sh, void* base, IrPrep& w) {
    const int cin = sh[IR_CIN], hid = sh[IR_HID], cout = sh[IR_COUT];
    Carver c(base);
    if (sh[IR_BF16]) {
        w.w1bf = c.take<u16>((size_t)hid * cin); w.w1tbf = c.take<u16>((size_t)hid * cin);
        w.w3bf = c.take<u16>((size_t)cout * hid); w.w3tbf = c.take<u16>((size_t)cout * hid);
        w.w1t = nullptr; w.w3t = nullptr;
    } else {
        w.w1bf = w.w1tbf = w.w3bf = w.w3tbf = nullptr;
        w.w1t = c.take<float>((size_t)hid * cin); w.w3t = c.take<float>((size_t)cout * hid);
    }
    return c.used + 256;
}
extern "C" long long v100_ir_prep_bytes(const int* shape) {
    IrPrep w;
    return (long long)ir_prep_carve(shape, nullptr, w);
}

static size_t ir_fwd_carve(const int* sh, void* base, float** stats) {
    const int B = sh[IR_B], hid = sh[IR_HID], cout = sh[IR_COUT], T = sh[IR_T];
    const int T2 = conv_out(T, sh[IR_K], sh[IR_STRIDE]);
    Carver c(base);
    size_t n = (size_t)v100_pw_num_parts(B, T) * hid * 2;
    const size_t n2 = (size_t)v100_dw_num_groups(B, hid) * hid * 2;
    const size_t n3 = (size_t)v100_pw_num_parts(B, T2) * cout * 2;
    if (n2 > n) n = n2;
    if (n3 > n) n = n3;
    *stats = c.take<float>(n);
    return c.used + 256;
}

extern "C" long long v100_ir_fwd_workspace_bytes(const int* shape) {
    float* s;
    return (long long)ir_fwd_carve(shape, nullptr, &s);
}

// ptrs: 0 x | 1 w1 2 g1 3 b1 4 rm1 5 rv1 6 nbt1 | 7 wd 8 g2 9 b2 10 rm2 11 rv2 12 nbt2 | 13 w3 14 g3 15 b3 16 rm3 17 rv3 18 nbt3 |
//       19 a1 20 a2 21 a3 22 y 23 coef 24 workspace 25 prepared-weights buffer (v100_ir_prep_bytes)
//       26 x16 27 y16 (act16 >= 4 only, each may be NULL): bf16 shadow [B][C][pitch16(T, B)] of the block input (written by the
//       previous block) / of this block's output (for the next block)
extern "C" int v100_ir_fwd_train(const int* sh, const void* const* P, void* stream) {
    if (!sh || !P) return V100_ERR_NULL;
    const int B = sh[IR_B], cin = sh[IR_CIN], hid = sh[IR_HID], cout = sh[IR_COUT], T = sh[IR_T], K = sh[IR_K], S = sh[IR_STRIDE];
    const int res = sh[IR_RES], bf = sh[IR_BF16];
    if (bf != 0 && bf != 1) return V100_ERR_SHAPE;            // training: fp32 or bf16 operands (fp16 = inference only)
    const int pad = (K - 1) / 2, T2 = conv_out(T, K, S);
    const float* x = (const float*)P[0];
    const float* w1 = (const float*)P[1]; const float* wd = (const float*)P[7]; const float* w3 = (const float*)P[13];
    float* a1 = (float*)P[19]; float* a2 = (float*)P[20]; float* a3 = (float*)P[21]; float* y = (float*)P[22];
    float* coef = (float*)P[23];
    const int mc = hid > cout ? hid : cout;          // coef vector stride
    float *s1 = coef, *t1 = coef + mc, *m1 = coef + 2 * mc, *r1 = coef + 3 * mc, *s2 = coef + 4 * mc, *t2 = coef + 5 * mc,
          *m2 = coef + 6 * mc, *r2 = coef + 7 * mc, *s3 = coef + 8 * mc, *t3 = coef + 9 * mc, *m3 = coef + 10 * mc, *r3 = coef + 11 * mc;
    float* st;
    ir_fwd_carve(sh, const_cast<void*>(P[24]), &st);
    IrPrep pw;
    ir_prep_carve(sh, const_cast<void*>(P[25]), pw);
    void *w1bf = pw.w1bf, *w3bf = pw.w3bf;
    int rc;
    // both orientations in one pass per weight: forward uses w*_bf, backward the transposed copies
    // (skipped when the caller has already filled `prep` for these weights, e.g. for every block of a stack in one
    // v100_ir_prep_batched launch)
    if (!(sh[IR_PREPPED] & 1)) {          // bit 0: prepared by v100_ir_prep_batched; bit 1: the pointer table has the shadow slots 26 / 27
        CK(v100_weight_prep(w1, hid, cin, pw.w1bf, pw.w1t, pw.w1tbf, stream));
        CK(v100_weight_prep(w3, cout, hid, pw.w3bf, pw.w3t, pw.w3tbf, stream));
    }
    const int parts1 = v100_pw_num_parts(B, T), parts3 = v100_pw_num_parts(B, T2), G = v100_dw_num_groups(B, hid);
    // PREPPED bit 3: FROZEN BatchNorm statistics (the block is in eval() but takes part in autograd): the same kernels with the running
    // statistics in place of the batch's (nothing updated), on the fp32-storage path only (no fused finalisers to teach the mode to)
    const bool frozen = (sh[IR_PREPPED] & 8) != 0;
    if (frozen && sh[IR_ACT16]) return V100_ERR_SHAPE;
    auto finalize = [&](const float* stats, int parts, long long count, int j, float* s_, float* t_, float* m_, float* r_, int C) -> int {
        // j: first slot of this BatchNorm in the pointer table (gamma, beta, running_mean, running_var, num_batches_tracked)
        if (frozen)
            return v100_bn_frozen_coeffs((const float*)P[j], (const float*)P[j + 1], (const float*)P[j + 2], (const float*)P[j + 3], kEps, s_, t_, m_, r_, C, stream);
        return v100_bn_finalize_train(stats, parts, count, (const float*)P[j], (const float*)P[j + 1], (float*)P[j + 2], (float*)P[j + 3],
                                      (long long*)P[j + 4], kMom, kEps, s_, t_, m_, r_, C, stream);
    };
    if (sh[IR_ACT16]) {
        if (!v100_ir_act16_supported(sh)) return V100_ERR_SHAPE;
        const void* x16 = (sh[IR_ACT16] >= 4 && (sh[IR_PREPPED] & 2)) ? P[26] : nullptr;
        void* y16 = (sh[IR_ACT16] >= 4 && (sh[IR_PREPPED] & 2)) ? const_cast<void*>(P[27]) : nullptr;
        if (x16) CK(v100_pw_gemm_io(w1bf, x16, nullptr, nullptr, nullptr, nullptr, 0, a1, nullptr, nullptr, nullptr, 1, st, B, hid, cin, T, PW_IO_X | PW_IO_Y, stream));
        else CK(v100_pw_gemm_io(w1bf, x, nullptr, nullptr, nullptr, nullptr, 0, a1, nullptr, nullptr, nullptr, 1, st, B, hid, cin, T, PW_IO_Y, stream));
        const bool pre1 = (IR_FUSE_PRE & 1) && G == 1;      // ... and BatchNorm 1 too, from the expand GEMM's slab, before its first row
        if (!pre1) CK(v100_bn_finalize_train(st, parts1, (long long)B * T, (const float*)P[2], (const float*)P[3], (float*)P[4], (float*)P[5], (long long*)P[6],
                                             kMom, kEps, s1, t1, m1, r1, hid, stream));
        if (G == 1) {          // the depthwise kernel finalises BatchNorm 2 itself (its workgroup owns the channel's sums)
            const DwFin fin{1, (double)B * T2, (const float*)P[8], (const float*)P[9], nullptr, s2, t2, nullptr, m2, r2,
                            (float*)P[10], (float*)P[11], (long long*)P[12], kMom, kEps};
            DwPre pre{};
            if (pre1) pre = DwPre{{1, (double)B * T, (const float*)P[2], (const float*)P[3], nullptr, s1, t1, nullptr, m1, r1,
                                   (float*)P[4], (float*)P[5], (long long*)P[6], kMom, kEps}, st, parts1};
            CK(dw_fwd_train_io_fin(a1, wd, s1, t1, a2, st, G, B, hid, T, K, DW_IO_X | DW_IO_Y, fin, pre, stream));
        } else {
            CK(v100_dwconv_fwd_train_io(a1, wd, s1, t1, a2, st, G, B, hid, T, K, DW_IO_X | DW_IO_Y, stream));
            CK(v100_bn_finalize_train(st, G, (long long)B * T2, (const float*)P[8], (const float*)P[9], (float*)P[10], (float*)P[11], (long long*)P[12],
                                      kMom, kEps, s2, t2, m2, r2, hid, stream));
        }
        const bool a316 = sh[IR_ACT16] >= 3;          // the project output (saved for backward) as bf16 too
        CK(v100_pw_gemm_io(w3bf, a2, nullptr, s2, t2, nullptr, 1, a3, nullptr, nullptr, nullptr, 1, st, B, cout, hid, T2, PW_IO_X | (a316 ? PW_IO_Y : 0), stream));
        if (IR_FUSE_BN3) {     // BatchNorm 3 finalised by the block-output pass itself (one workgroup per channel)
            const DwPre pre{{1, (double)B * T2, (const float*)P[14], (const float*)P[15], nullptr, s3, t3, nullptr, m3, r3,
                             (float*)P[16], (float*)P[17], (long long*)P[18], kMom, kEps}, st, parts3};
            // level 5: the residual comes from the bf16 shadow of x (what the expand GEMM above read), and an interior block of a stack
            // (PREPPED bit 2: nothing reads its fp32 output) writes only the shadow of its own output
            const bool r16 = sh[IR_ACT16] >= 5 && x16 && a316;            // (the previous block may not have written its fp32 output at all)
            const bool noy = sh[IR_ACT16] >= 5 && y16 && a316 && (sh[IR_PREPPED] & 4);
            CK(chan_affine2_fin(a3, res ? (r16 ? x16 : (const void*)x) : nullptr, noy ? nullptr : y, y16, B, cout, T2, a316 ? 1 : 0, pre, stream, r16 ? 1 : 0));
            return V100_OK;
        }
        CK(v100_bn_finalize_train(st, parts3, (long long)B * T2, (const float*)P[14], (const float*)P[15], (float*)P[16], (float*)P[17], (long long*)P[18],
                                  kMom, kEps, s3, t3, m3, r3, cout, stream));
        if (y16) CK(v100_chan_affine2_shadow(a3, res ? x : nullptr, s3, t3, y, y16, B, cout, T2, a316 ? 1 : 0, stream));
        else if (a316) CK(v100_chan_affine2_io(a3, res ? x : nullptr, s3, nullptr, t3, y, B, cout, T2, 1, stream));
        else CK(v100_chan_affine2(a3, res ? x : nullptr, s3, nullptr, t3, y, B, cout, T2, stream));
        return V100_OK;
    }
    CK(v100_pw_gemm(w1, w1bf, x, nullptr, nullptr, nullptr, nullptr, 0, a1, nullptr, nullptr, nullptr, nullptr, 1, st, B, hid, cin, T, bf, stream));
    CK(finalize(st, parts1, (long long)B * T, 2, s1, t1, m1, r1, hid));
    CK(v100_dwconv(a1, nullptr, wd, s1, t1, nullptr, 1, a2, nullptr, nullptr, nullptr, 0, st, G, B, hid, T, T2, K, S, pad, 0, 1, 0, stream));
    CK(finalize(st, G, (long long)B * T2, 8, s2, t2, m2, r2, hid));
    CK(v100_pw_gemm(w3, w3bf, a2, nullptr, s2, t2, nullptr, 1, a3, nullptr, nullptr, nullptr, nullptr, 1, st, B, cout, hid, T2, bf, stream));
    // a block that itself keeps fp32 storage (the stride-2 first layer) still emits the shadow its successor reads: the caller
    // passes P[27] only in bf16 precision at act16 level 4 (26 / 27 are not read otherwise) -- and, like the 16-bit blocks, lets that
    // pass finalise BatchNorm 3 itself (one launch fewer per step)
    if (IR_FUSE_BN3 && !frozen && bf == 1 && (sh[IR_PREPPED] & 2) && P[27]) {
        const DwPre pre{{1, (double)B * T2, (const float*)P[14], (const float*)P[15], nullptr, s3, t3, nullptr, m3, r3,
                         (float*)P[16], (float*)P[17], (long long*)P[18], kMom, kEps}, st, parts3};
        CK(chan_affine2_fin(a3, res ? x : nullptr, y, const_cast<void*>(P[27]), B, cout, T2, 0, pre, stream, 0));
        return V100_OK;
    }
    CK(finalize(st, parts3, (long long)B * T2, 14, s3, t3, m3, r3, cout));
    if (bf == 1 && (sh[IR_PREPPED] & 2) && P[27]) CK(v100_chan_affine2_shadow(a3, res ? x : nullptr, s3, t3, y, const_cast<void*>(P[27]), B, cout, T2, 0, stream));
    else CK(v100_chan_affine2(a3, res ? x : nullptr, s3, nullptr, t3, y, B, cout, T2, stream));
    return V100_OK;
}

struct IrBwdWs {
    float *part, *da3, *dz2, *dz1, *slab, *pqr;
};
static size_t ir_bwd_carve(const int* sh, void* base, IrBwdWs& w) {
    const int B = sh[IR_B], cin = sh[IR_CIN], hid = sh[IR_HID], cout = sh[IR_COUT], T = sh[IR_T], K = sh[IR_K];
    const int T2 = conv_out(T, K, sh[IR_STRIDE]);
    Carver c(base);
    size_t n = (size_t)v100_dw_num_groups(B, cout) * cout * 2;
    size_t m = (size_t)v100_pw_num_parts(B, T2) * hid * 2; if (m > n) n = m;
    m = (size_t)v100_dw_num_groups(B, hid) * hid * 2; if (m > n) n = m;
    w.part = c.take<float>(n);
    w.da3 = sh[IR_ACT16] >= 3 ? (float*)c.take<u16>((size_t)B * cout * pitch16(T2, B)) : c.take<float>((size_t)B * cout * T2);
    // act16 == 2: the two hidden gradients are bf16 with pitched rows (half the bytes; the fp32 size is an upper bound
    // only when pitch16(T) <= 2*T, i.e. always)
    if (sh[IR_ACT16] >= 2) {
        w.dz2 = (float*)c.take<u16>((size_t)B * hid * pitch16(T2, B));
        w.dz1 = (float*)c.take<u16>((size_t)B * hid * pitch16(T, B));
    } else {
        w.dz2 = c.take<float>((size_t)B * hid * T2);
        w.dz1 = c.take<float>((size_t)B * hid * T);
    }
    n = (size_t)v100_pw_wgrad_splits(B, cout, hid) * cout * hid;
    m = (size_t)v100_pw_wgrad_splits(B, hid, cin) * hid * cin; if (m > n) n = m;
    m = (size_t)v100_dw_num_groups(B, hid) * hid * K; if (m > n) n = m;
    w.slab = c.take<float>(n);
    w.pqr = c.take<float>((size_t)6 * (hid > cout ? hid : cout));       // two coefficient sets (see DwFin in v100_ir_bwd)
    return c.used + 256;
}

extern "C" long long v100_ir_bwd_workspace_bytes(const int* shape) {
    IrBwdWs w;
    return (long long)ir_bwd_carve(shape, nullptr, w);
}

// ptrs: 0 x 1 a1 2 a2 3 a3 | 4 w1 5 wd 6 w3 | 7 g1 8 g2 9 g3 | 10 coef | 11 dy | 12 dx (may be NULL) |
//       13 dW1 14 dg1 15 db1 16 dWd 17 dg2 18 db2 19 dW3 20 dg3 21 db3 | 22 workspace | 23 prepared weights (from forward)
//       24 x16 (act16 >= 4 only, may be NULL): the bf16 shadow of x the forward read
extern "C" int v100_ir_bwd(const int* sh, const void* const* P, void* stream) {
    if (!sh || !P) return V100_ERR_NULL;
    const int B = sh[IR_B], cin = sh[IR_CIN], hid = sh[IR_HID], cout = sh[IR_COUT], T = sh[IR_T], K = sh[IR_K], S = sh[IR_STRIDE];
    const int res = sh[IR_RES], bf = sh[IR_BF16];
    const int pad = (K - 1) / 2, T2 = conv_out(T, K, S);
    const float *x = (const float*)P[0], *a1 = (const float*)P[1], *a2 = (const float*)P[2], *a3 = (const float*)P[3];
    const float *w1 = (const float*)P[4], *wd = (const float*)P[5], *w3 = (const float*)P[6];
    const float *g1 = (const float*)P[7], *g2 = (const float*)P[8], *g3 = (const float*)P[9];
    const float* coef = (const float*)P[10];
    const int mc = hid > cout ? hid : cout;
    const float *s1 = coef, *t1 = coef + mc, *m1 = coef + 2 * mc, *r1 = coef + 3 * mc, *s2 = coef + 4 * mc, *t2 = coef + 5 * mc,
                *m2 = coef + 6 * mc, *r2 = coef + 7 * mc, *m3 = coef + 10 * mc, *r3 = coef + 11 * mc;
    const float* dy = (const float*)P[11];
    float* dx = (float*)P[12];
    // PREPPED bits 4 / 5 (set by the stack executor only, ir_grad16_ok): dy arrives / dx leaves as bf16 [B][C][pitch16(T)] -- the
    // gradient stream between the blocks of a stack in the 16-bit form its forward stream has at level 5
    const int dy16 = (sh[IR_PREPPED] & 16) ? 1 : 0, dx16 = (sh[IR_PREPPED] & 32) ? 1 : 0;
    IrBwdWs w;
    ir_bwd_carve(sh, const_cast<void*>(P[22]), w);
    float *pp = w.pqr, *qq = w.pqr + mc, *rr = w.pqr + 2 * mc;
    int rc;
    IrPrep pw;
    ir_prep_carve(sh, const_cast<void*>(P[23]), pw);
    (void)w1; (void)w3;
    const bool frozen = (sh[IR_PREPPED] & 8) != 0;     // frozen BatchNorm statistics (see v100_ir_fwd_train): q = r = 0 in every coefficient set
    if (frozen && sh[IR_ACT16]) return V100_ERR_SHAPE;
    auto bwd_finalize = frozen ? v100_bn_bwd_finalize_frozen : v100_bn_bwd_finalize;
    // BN3 backward
    const int Gr = v100_dw_num_groups(B, cout);
    const bool a316 = sh[IR_ACT16] >= 3;              // a3 (saved) and da3 (workspace) stored as bf16
    bool da3_done = false;
    if (a316 && IR_FUSE_BN3) {     // one workgroup per channel sums (dy, dy * a3) and finalises BatchNorm 3's backward itself ...
        const DwFin fin{2, (double)B * T2, g3, m3, r3, pp, qq, rr, (float*)P[20], (float*)P[21], nullptr, nullptr, nullptr, 0.f, 0.f};
        // ... and, when the channel's samples fit its registers, applies the affine to them in the same launch
        if (IR_FUSE_BN3 >= 2 && chan_bn3_bwd(dy, a3, w.part, w.da3, B, cout, T2, fin, stream, dy16)) {
            if ((rc = v100_launch_status()) != V100_OK) return rc;
            da3_done = true;
        } else if (dy16) return V100_ERR_SHAPE;           // (the stack executor asks for a 16-bit dy only where the one-pass kernel applies)
        else CK(chan_reduce2_io_fin(dy, a3, w.part, B, cout, T2, fin, stream));
    } else {
        if (dy16) return V100_ERR_SHAPE;
        if (a316) CK(v100_chan_reduce2_io(dy, a3, w.part, Gr, B, cout, T2, 2, stream));
        else CK(v100_chan_reduce2(dy, a3, w.part, Gr, B, cout, T2, stream));
        CK(bwd_finalize(w.part, Gr, (long long)B * T2, g3, m3, r3, pp, qq, rr, (float*)P[20], (float*)P[21], cout, stream));
    }
    if (da3_done) {}
    else if (a316) CK(v100_chan_affine2_io(dy, a3, pp, qq, rr, w.da3, B, cout, T2, 6, stream));
    else CK(v100_chan_affine2(dy, a3, pp, qq, rr, w.da3, B, cout, T2, stream));
    if (sh[IR_ACT16]) {
        if (!v100_ir_act16_supported(sh)) return V100_ERR_SHAPE;
        const bool g16 = sh[IR_ACT16] >= 2;
        CK(v100_pw_wgrad_io(w.da3, nullptr, nullptr, nullptr, nullptr, 0, a2, s2, t2, 1, w.slab, (float*)P[19], v100_pw_wgrad_splits(B, cout, hid),
                            B, cout, hid, T2, WG_IO_X | (a316 ? WG_IO_G : 0), stream));
        const int parts16 = v100_pw_num_parts(B, T2);
        CK(v100_pw_gemm_io(pw.w3tbf, w.da3, nullptr, nullptr, nullptr, nullptr, 0, w.dz2, s2, t2, a2, 4, w.part, B, hid, cout, T2,
                           PW_IO_R | (g16 ? PW_IO_Y : 0) | (a316 ? PW_IO_X : 0), stream));
        const int G16 = v100_dw_num_groups(B, hid);
        bool da1 = false;
        const bool pre2 = (IR_FUSE_PRE & 2) && G16 == 1;    // BatchNorm-2 backward coefficients from the GEMM's slab, by the depthwise kernel
        if (!pre2) CK(v100_bn_bwd_finalize(w.part, parts16, (long long)B * T2, g2, m2, r2, pp, qq, rr, (float*)P[17], (float*)P[18], hid, stream));
        if (G16 == 1) {        // BatchNorm-1 backward coefficients by the depthwise backward kernel itself
            // p / q / r are INPUTS of this kernel (BatchNorm-2 backward) and outputs of its finalisation (BatchNorm-1 backward):
            // the outputs go to the second coefficient set
            float *pp2 = w.pqr + 3 * mc, *qq2 = w.pqr + 4 * mc, *rr2 = w.pqr + 5 * mc;
            const DwFin fin{2, (double)B * T, g1, m1, r1, pp2, qq2, rr2, (float*)P[14], (float*)P[15], nullptr, nullptr, nullptr, 0.f, 0.f};
            DwPre pre{};
            if (pre2) pre = DwPre{{2, (double)B * T2, g2, m2, r2, pp, qq, rr, (float*)P[17], (float*)P[18], nullptr, nullptr, nullptr, 0.f, 0.f},
                                  w.part, parts16};
            // Round 5: where the streaming kernel can keep a wave's rows in registers (one group, B <= 32, every hidden tensor bf16) it
            // writes the FINISHED gradient da1 = p dz1 + q a1 + r (into the dz1 buffer) and the two consumers below read ONE plain
            // bf16 tensor -- bit for bit the operand their on-load transform of (dz1, a1) produced.
            da1 = g16 && !frozen && dw_bwd_da1_supported(B, hid, T, K, G16);
            CK(dw_bwd_io_fin(w.dz2, a2, wd, pp, qq, rr, a1, s1, t1, w.dz1, w.part, w.slab, (float*)P[16], G16, B, hid, T, K,
                             g16 ? (DW_IO_X | DW_IO_X2 | DW_IO_AUX | DW_IO_Y) : (DW_IO_X2 | DW_IO_AUX), fin, pre, stream, da1 ? 1 : 0));
            pp = pp2; qq = qq2; rr = rr2;
        } else {
            CK(v100_dwconv_bwd_io(w.dz2, a2, wd, pp, qq, rr, a1, s1, t1, w.dz1, w.part, w.slab, (float*)P[16], G16, B, hid, T, K,
                                  g16 ? (DW_IO_X | DW_IO_X2 | DW_IO_AUX | DW_IO_Y) : (DW_IO_X2 | DW_IO_AUX), stream));
            CK(v100_bn_bwd_finalize(w.part, G16, (long long)B * T, g1, m1, r1, pp, qq, rr, (float*)P[14], (float*)P[15], hid, stream));
        }
        const void* x16 = (sh[IR_ACT16] >= 4 && (sh[IR_PREPPED] & 2)) ? P[24] : nullptr;
        if (da1) {
            CK(v100_pw_wgrad_io(w.dz1, nullptr, nullptr, nullptr, nullptr, 0, x16 ? x16 : (const void*)x, nullptr, nullptr, 0, w.slab, (float*)P[13],
                                v100_pw_wgrad_splits(B, hid, cin), B, hid, cin, T, WG_IO_G | (x16 ? WG_IO_X : 0), stream));
            if (dx)
                CK(v100_pw_gemm_io(pw.w1tbf, w.dz1, nullptr, nullptr, nullptr, nullptr, 0, dx, nullptr, nullptr, res ? dy : nullptr, res ? 5 : 0, nullptr,
                                   B, cin, hid, T, PW_IO_X | ((dy16 && res) ? PW_IO_R : 0) | (dx16 ? PW_IO_Y : 0), stream));
            return V100_OK;
        }
        if (dy16 || dx16) return V100_ERR_SHAPE;
        CK(v100_pw_wgrad_io(w.dz1, a1, pp, qq, rr, 2, x16 ? x16 : (const void*)x, nullptr, nullptr, 0, w.slab, (float*)P[13], v100_pw_wgrad_splits(B, hid, cin),
                            B, hid, cin, T, WG_IO_G2 | (g16 ? WG_IO_G : 0) | (x16 ? WG_IO_X : 0), stream));
        if (dx)
            CK(v100_pw_gemm_io(pw.w1tbf, w.dz1, a1, pp, qq, rr, 2, dx, nullptr, nullptr, res ? dy : nullptr, res ? 5 : 0, nullptr,
                               B, cin, hid, T, PW_IO_X2 | (g16 ? PW_IO_X : 0), stream));
        return V100_OK;
    }
    if (dy16 || dx16) return V100_ERR_SHAPE;
    // pw-linear: weight grad, then data grad through ReLU6 with BN2-backward sums
    CK(v100_pw_wgrad(w.da3, nullptr, nullptr, nullptr, nullptr, 0, a2, s2, t2, 1, w.slab, (float*)P[19], v100_pw_wgrad_splits(B, cout, hid),
                     B, cout, hid, T2, bf, stream));
    const int parts = v100_pw_num_parts(B, T2);
    CK(v100_pw_gemm(pw.w3t, pw.w3tbf, w.da3, nullptr, nullptr, nullptr, nullptr, 0, w.dz2, nullptr, s2, t2, a2, 4, w.part, B, hid, cout, T2, bf, stream));
    CK(bwd_finalize(w.part, parts, (long long)B * T2, g2, m2, r2, pp, qq, rr, (float*)P[17], (float*)P[18], hid, stream));
    // depthwise: weight grad, then data grad through ReLU6 with BN1-backward sums
    const int G = v100_dw_num_groups(B, hid);
    CK(v100_dwconv_bwd(w.dz2, a2, wd, pp, qq, rr, a1, s1, t1, w.dz1, w.part, w.slab, (float*)P[16], G, B, hid, T, T2, K, S, pad, 0, stream));
    CK(bwd_finalize(w.part, G, (long long)B * T, g1, m1, r1, pp, qq, rr, (float*)P[14], (float*)P[15], hid, stream));
    // pw: weight grad, data grad (+ residual)
    CK(v100_pw_wgrad(w.dz1, a1, pp, qq, rr, 2, x, nullptr, nullptr, 0, w.slab, (float*)P[13], v100_pw_wgrad_splits(B, hid, cin),
                     B, hid, cin, T, bf, stream));
    if (dx)
        CK(v100_pw_gemm(pw.w1t, pw.w1tbf, w.dz1, a1, pp, qq, rr, 2, dx, nullptr, nullptr, nullptr, res ? dy : nullptr, res ? 5 : 0, nullptr,
                        B, cin, hid, T, bf, stream));
    return V100_OK;
}

// ---- eval mode: BatchNorm folded to per-channel scale/shift inside three kernels (pw+BN+ReLU6, dw+BN+ReLU6, pw+BN(+x)).
// The folded coefficients and the bf16 weight copies only change when the parameters do, so they live in a caller-owned
// cache filled by v100_ir_eval_prep; a forward is then 3 launches from one call.
struct IrEvalCache { void *w1bf, *w3bf; float *s1, *t1, *s2, *t2, *s3, *t3; };
static size_t ir_eval_carve(const int* sh, void* base, IrEvalCache& w) {
    const int cin = sh[IR_CIN], hid = sh[IR_HID], cout = sh[IR_COUT];
    Carver c(base);
    w.w1bf = w.w3bf = nullptr;
    if (sh[IR_BF16]) { w.w1bf = c.take<u16>((size_t)hid * cin); w.w3bf = c.take<u16>((size_t)cout * hid); }
    w.s1 = c.take<float>(hid); w.t1 = c.take<float>(hid); w.s2 = c.take<float>(hid); w.t2 = c.take<float>(hid);
    w.s3 = c.take<float>(cout); w.t3 = c.take<float>(cout);
    return c.used + 256;
}
extern "C" long long v100_ir_eval_cache_bytes(const int* shape) {
    IrEvalCache w;
    return (long long)ir_eval_carve(shape, nullptr, w);
}
// ptrs: 0 w1 | 1 g1 2 b1 3 rm1 4 rv1 | 5 g2 6 b2 7 rm2 8 rv2 | 9 w3 | 10 g3 11 b3 12 rm3 13 rv3 | 14 cache
extern "C" int v100_ir_eval_prep(const int* sh, const void* const* P, void* stream) {
    if (!sh || !P) return V100_ERR_NULL;
    const int cin = sh[IR_CIN], hid = sh[IR_HID], cout = sh[IR_COUT];
    IrEvalCache c;
    ir_eval_carve(sh, const_cast<void*>(P[14]), c);
    int rc;
    if (sh[IR_BF16] == 2) {                                   // fp16 operands
        CK(v100_weight_prep_f16((const float*)P[0], hid, cin, c.w1bf, stream));
        CK(v100_weight_prep_f16((const float*)P[9], cout, hid, c.w3bf, stream));
    } else if (sh[IR_BF16]) {
        CK(v100_weight_prep((const float*)P[0], hid, cin, c.w1bf, nullptr, nullptr, stream));
        CK(v100_weight_prep((const float*)P[9], cout, hid, c.w3bf, nullptr, nullptr, stream));
    }
    CK(v100_bn_eval_coeffs((const float*)P[1], (const float*)P[2], (const float*)P[3], (const float*)P[4], kEps, c.s1, c.t1, hid, stream));
    CK(v100_bn_eval_coeffs((const float*)P[5], (const float*)P[6], (const float*)P[7], (const float*)P[8], kEps, c.s2, c.t2, hid, stream));
    CK(v100_bn_eval_coeffs((const float*)P[10], (const float*)P[11], (const float*)P[12], (const float*)P[13], kEps, c.s3, c.t3, cout, stream));
    return V100_OK;
}
// ptrs: 0 x | 1 w1 2 wd 3 w3 | 4 cache | 5 h1 [B,hid,T] 6 h2 [B,hid,T2] 7 y [B,cout,T2]
extern "C" int v100_ir_fwd_eval(const int* sh, const void* const* P, void* stream) {
    if (!sh || !P) return V100_ERR_NULL;
    const int B = sh[IR_B], cin = sh[IR_CIN], hid = sh[IR_HID], cout = sh[IR_COUT], T = sh[IR_T], K = sh[IR_K], S = sh[IR_STRIDE];
    const int res = sh[IR_RES], bf = sh[IR_BF16];
    const int pad = (K - 1) / 2, T2 = conv_out(T, K, S);
    const float* x = (const float*)P[0];
    IrEvalCache c;
    ir_eval_carve(sh, const_cast<void*>(P[4]), c);
    float *h1 = (float*)P[5], *h2 = (float*)P[6], *y = (float*)P[7];
    int rc;
    if (sh[IR_ACT16] == 2) {
        // CHANNEL-MAJOR inference (round 4): x [cin][B P] fp32, h1 / h2 [hid][B P] 16-bit, y [cout][B P] fp32, P = pitch16(T) -- every
        // utterance of the batch back to back in ONE [C x (B P)] matrix.  The two 1x1 convolutions are then ONE GEMM over all B P columns
        // (1-second chunks, T' = 51: 128-column tiles are full instead of 40 % full), a channel's rows are contiguous for the depthwise
        // kernel (the order it walks them in), short rows are packed several to a wave item.  The columns T .. P-1 of an utterance are
        // padding: finite garbage that no valid column ever reads (GEMM columns are independent; the depthwise kernel masks them).
        // Hidden tensors in the GEMMs' operand format: bf16, or fp16 at precision "fp16".
        int shb[IR_NSHAPE];
        for (int i = 0; i < IR_NSHAPE; ++i) shb[i] = sh[i];
        shb[IR_BF16] = 1;
        if ((bf != 1 && bf != 2) || S != 1 || T > 768 || !v100_ir_act16_supported(shb)) return V100_ERR_SHAPE;
        const long long N = (long long)B * pitch16(T, B);
        if (N > 0x7fffff00LL) return V100_ERR_SHAPE;
        const int f16 = bf == 2 ? PW_IO_F16 : 0;
        CK(v100_pw_gemm_io(c.w1bf, x, nullptr, nullptr, nullptr, nullptr, 0, h1, c.s1, c.t1, nullptr, 2, nullptr, 1, hid, cin, (int)N, PW_IO_Y | f16, stream));
        CK(dw_fwd_eval_io(h1, (const float*)P[2], c.s2, c.t2, h2, B, hid, T, K, stream, 1, bf == 2));
        CK(v100_pw_gemm_io(c.w3bf, h2, nullptr, nullptr, nullptr, nullptr, 0, y, c.s3, c.t3, res ? x : nullptr, 3, nullptr, 1, cout, hid, (int)N, PW_IO_X | f16, stream));
        return V100_OK;
    }
    if (sh[IR_ACT16]) {
        // inference at precision "bf16": the two hidden tensors (each 4x the block's width, already BatchNorm'ed and clamped to [0, 6])
        // are stored as bf16 [B][hid][pitch16(T)] -- half the bytes of the three kernels' big streams; the GEMMs round them to bf16 as
        // operands anyway.  Same three launches.
        if (bf != 1 || !v100_ir_act16_supported(sh)) return V100_ERR_SHAPE;
        CK(v100_pw_gemm_io(c.w1bf, x, nullptr, nullptr, nullptr, nullptr, 0, h1, c.s1, c.t1, nullptr, 2, nullptr, B, hid, cin, T, PW_IO_Y, stream));
        CK(dw_fwd_eval_io(h1, (const float*)P[2], c.s2, c.t2, h2, B, hid, T, K, stream));
        CK(v100_pw_gemm_io(c.w3bf, h2, nullptr, nullptr, nullptr, nullptr, 0, y, c.s3, c.t3, res ? x : nullptr, 3, nullptr, B, cout, hid, T2, PW_IO_X, stream));
        return V100_OK;
    }
    CK(v100_pw_gemm((const float*)P[1], c.w1bf, x, nullptr, nullptr, nullptr, nullptr, 0, h1, nullptr, c.s1, c.t1, nullptr, 2, nullptr, B, hid, cin, T, bf, stream));
    CK(v100_dwconv(h1, nullptr, (const float*)P[2], nullptr, nullptr, nullptr, 0, h2, nullptr, c.s2, c.t2, 1, nullptr, v100_dw_num_groups(B, hid),
                   B, hid, T, T2, K, S, pad, 0, 1, 0, stream));
    CK(v100_pw_gemm((const float*)P[3], c.w3bf, h2, nullptr, nullptr, nullptr, nullptr, 0, y, nullptr, c.s3, c.t3, res ? x : nullptr, 3, nullptr, B, cout, hid, T2, bf, stream));
    return V100_OK;
}

extern "C" int v100_ir_stack_fwd_eval(int n, const int* shapes, const void* const* ptrs, void* stream) {
    if (!shapes || !ptrs) return V100_ERR_NULL;
    if (n < 0) return V100_ERR_SHAPE;
    for (int i = 0; i < n; ++i) {
        const int rc = v100_ir_fwd_eval(shapes + (size_t)i * IR_NSHAPE, ptrs + (size_t)i * 8, stream);
        if (rc != V100_OK) return rc;
    }
    return V100_OK;
}

// Prepared weights of n blocks in ONE launch (the per-block pair of weight_prep launches costs ~13 us a block, mostly
// launch latency).  shapes: n x IR_NSHAPE ints; w1s / w3s: the fp32 weights; preps: each block's prep buffer.  HOST arrays.
struct PrepBatch {
    enum { MAXN = 32 };
    const float* w[2 * MAXN]; u16* wbf[2 * MAXN]; float* wt[2 * MAXN]; u16* wtbf[2 * MAXN];
    int rows[2 * MAXN], cols[2 * MAXN];
};
// 64 x 64 tiles through LDS: the row-major bf16 copy and BOTH transposed copies are written in full lines (the element-wise
// form scattered the transposed copies 2 / 4 bytes at a time: 58 us per step for 46 MB of weights).
__global__ __launch_bounds__(256) void weight_prep_batched_kernel(PrepBatch t) {
    __shared__ float tile[64][65];
    const int e = blockIdx.y;
    const float* __restrict__ w = t.w[e];
    const int rows = t.rows[e], cols = t.cols[e];
    const int tr = (rows + 63) >> 6, tc = (cols + 63) >> 6;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;            // 64 columns x 4 row groups
    // Whole 64 x 64 tiles of a bf16-only preparation (every 1x1 weight of the step's networks): 16-byte loads, 8-byte packed stores on
    // both copies (the element-wise form below moved the transposed copy 2 bytes per lane: 30.7 us for the step's 86.7 MB, 2.8 TB/s).
    const bool fast = !(rows & 63) && !(cols & 63) && t.wt[e] == nullptr && t.wbf[e] != nullptr && t.wtbf[e] != nullptr;
    for (int tl = blockIdx.x; fast && tl < tr * tc; tl += gridDim.x) {
        const int r0 = (tl / tc) << 6, c0 = (tl % tc) << 6;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = threadIdx.x + 256 * q;                      // 64 rows x 16 float4
            const int i = id >> 4, c4 = (id & 15) << 2;
            const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)(r0 + i) * cols + c0 + c4);
            uint2 o; o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]);
            *reinterpret_cast<uint2*>(t.wbf[e] + (size_t)(r0 + i) * cols + c0 + c4) = o;
            tile[i][c4] = v[0]; tile[i][c4 + 1] = v[1]; tile[i][c4 + 2] = v[2]; tile[i][c4 + 3] = v[3];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = threadIdx.x + 256 * q;                      // 64 columns x 16 groups of 4 rows
            const int c = id >> 4, r4 = (id & 15) << 2;
            uint2 o; o.x = pack_bf16(tile[r4][c], tile[r4 + 1][c]); o.y = pack_bf16(tile[r4 + 2][c], tile[r4 + 3][c]);
            *reinterpret_cast<uint2*>(t.wtbf[e] + (size_t)(c0 + c) * rows + r0 + r4) = o;
        }
    }
    for (int tl = blockIdx.x; !fast && tl < tr * tc; tl += gridDim.x) {
        const int r0 = (tl / tc) << 6, c0 = (tl % tc) << 6;
        __syncthreads();                                               // the previous tile's reads are done
#pragma unroll 4
        for (int i = ty; i < 64; i += 4) {
            const int r = r0 + i, c = c0 + tx;
            float v = 0.f;
            if (r < rows && c < cols) {
                v = w[(size_t)r * cols + c];
                if (t.wbf[e]) t.wbf[e][(size_t)r * cols + c] = f2bf(v);
            }
            tile[i][tx] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = ty; i < 64; i += 4) {
            const int c = c0 + i, r = r0 + tx;                         // transposed: consecutive threads walk the ROWS of w
            if (c < cols && r < rows) {
                const float v = tile[tx][i];
                if (t.wt[e]) t.wt[e][(size_t)c * rows + r] = v;
                if (t.wtbf[e]) t.wtbf[e][(size_t)c * rows + r] = f2bf(v);
            }
        }
    }
}
extern "C" int v100_ir_prep_batched(const int* shapes, const void* const* w1s, const void* const* w3s, void* const* preps, int n,
                                    void* stream) {
    if (!shapes || !w1s || !w3s || !preps) return V100_ERR_NULL;
    if (n <= 0 || n > PrepBatch::MAXN) return V100_ERR_SHAPE;
    PrepBatch t;
    long maxn = 0;
    for (int i = 0; i < n; ++i) {
        const int* sh = shapes + (size_t)i * IR_NSHAPE;
        if (!w1s[i] || !w3s[i] || !preps[i]) return V100_ERR_NULL;
        IrPrep pw;
        ir_prep_carve(sh, preps[i], pw);
        t.w[2 * i] = (const float*)w1s[i]; t.rows[2 * i] = sh[IR_HID]; t.cols[2 * i] = sh[IR_CIN];
        t.wbf[2 * i] = (u16*)pw.w1bf; t.wt[2 * i] = pw.w1t; t.wtbf[2 * i] = (u16*)pw.w1tbf;
        t.w[2 * i + 1] = (const float*)w3s[i]; t.rows[2 * i + 1] = sh[IR_COUT]; t.cols[2 * i + 1] = sh[IR_HID];
        t.wbf[2 * i + 1] = (u16*)pw.w3bf; t.wt[2 * i + 1] = pw.w3t; t.wtbf[2 * i + 1] = (u16*)pw.w3tbf;
        const long a = (long)sh[IR_HID] * sh[IR_CIN], b = (long)sh[IR_COUT] * sh[IR_HID];
        if (a > maxn) maxn = a;
        if (b > maxn) maxn = b;
        if (a <= 0 || b <= 0) return V100_ERR_SHAPE;
    }
    long gx = (maxn + 4095) / 4096;           // 64 x 64 tiles of the largest matrix
    if (gx > 256) gx = 256;
    if (gx < 1) gx = 1;
    V100_GGL(weight_prep_batched_kernel, dim3((unsigned)gx, 2 * n), dim3(256), 0, (hipStream_t)stream, t);
    return v100_launch_status();
}

// =====================================================================================================================
// Stack executor: ONE host call per direction for a run of consecutive InvertedResidual blocks (the encoder's nine, a decoder
// segment's three or four): the per-block Python -> autograd -> ctypes crossings were 2/3 of the 3 ms a step took to ENQUEUE.
// All activations the run keeps for backward live in ONE caller-allocated blob laid out by v100_ir_stack_plan; backward walks the
// blocks in reverse with one shared workspace and two ping-pong gradient buffers.  Pure sequencing of v100_ir_fwd_train / v100_ir_bwd
// (same kernels, same order, bit-identical results).
//
// desc (HOST ints): {n, B, T, bf16, act16_level, want_last_shadow} then n x {cin, hid, cout, k, stride, residual}
// plan (HOST long long out, 8 per block + 6): per block byte offsets into the blob of a1, a2, a3, y, y16 (or -1), coef, prep, T_out;
//   then {blob_bytes, bwd_ws_bytes, grad_floats, fwd_ws_offset, 0, 0}
// params (HOST pointer table, 18 per block): w1 g1 b1 rm1 rv1 nbt1 | wd g2 b2 rm2 rv2 nbt2 | w3 g3 b3 rm3 rv3 nbt3
enum { ST_N, ST_B, ST_T, ST_BF16, ST_LEVEL, ST_SHADOW, ST_HDR };
namespace {
struct StackBlock { int sh[IR_NSHAPE]; long long a1, a2, a3, y, y16, coef, prep; int Tout; size_t grad_floats; };
inline size_t al256(size_t v) { return (v + 255) & ~size_t(255); }
// fills blocks[0..n) and the totals; returns 0 on a bad descriptor
int stack_layout(const int* desc, StackBlock* blk, size_t& blob_bytes, size_t& bwd_ws, size_t& grad_floats, size_t& fwd_ws_off) {
    const int n = desc[ST_N], B = desc[ST_B], bf = desc[ST_BF16], level = desc[ST_LEVEL];
    if (n <= 0 || n > 32 || B <= 0 || desc[ST_T] <= 0 || (bf != 0 && bf != 1)) return 0;
    int T = desc[ST_T];
    size_t off = 0, fwd_ws = 0, bws = 0, dxmax = 0;
    grad_floats = 0;
    const bool shadows = bf == 1 && level >= 4;
    for (int i = 0; i < n; ++i) {
        const int* d = desc + ST_HDR + 6 * i;
        StackBlock& b = blk[i];
        const int cin = d[0], hid = d[1], cout = d[2], k = d[3], s = d[4], res = d[5];
        if (cin <= 0 || hid <= 0 || cout <= 0 || k <= 0 || s <= 0) return 0;
        int* sh = b.sh;
        sh[IR_B] = B; sh[IR_CIN] = cin; sh[IR_HID] = hid; sh[IR_COUT] = cout; sh[IR_T] = T; sh[IR_K] = k; sh[IR_STRIDE] = s;
        sh[IR_RES] = res; sh[IR_BF16] = bf; sh[IR_PREPPED] = 1 | (shadows ? 2 : 0); sh[IR_ACT16] = 0;
        if (bf == 1 && level && v100_ir_act16_supported(sh)) sh[IR_ACT16] = level;
        const int T2 = conv_out(T, k, s);
        if (T2 <= 0) return 0;
        b.Tout = T2;
        const int lv = sh[IR_ACT16];
        const size_t P = pitch16(T, B), P2 = pitch16(T2, B);
        b.a1 = (long long)off; off = al256(off + (lv ? (size_t)B * hid * P * 2 : (size_t)B * hid * T * 4));
        b.a2 = (long long)off; off = al256(off + (lv ? (size_t)B * hid * P * 2 : (size_t)B * hid * T2 * 4));   // (act16: stride 1, T2 == T)
        b.a3 = (long long)off; off = al256(off + (lv >= 3 ? (size_t)B * cout * P * 2 : (size_t)B * cout * T2 * 4));
        b.y = (long long)off; off = al256(off + (size_t)B * cout * T2 * 4);
        b.y16 = -1;
        if (shadows && (i + 1 < n || desc[ST_SHADOW])) { b.y16 = (long long)off; off = al256(off + (size_t)B * cout * P2 * 2); }
        // level 5: block i - 1 need not write its fp32 output when this block (its only reader inside the stack) takes x from the shadow
        if (i > 0 && lv >= 5 && shadows && blk[i - 1].y16 >= 0 && blk[i - 1].sh[IR_ACT16] >= 5) blk[i - 1].sh[IR_PREPPED] |= 4;
        b.coef = (long long)off; off = al256(off + (size_t)12 * (hid > cout ? hid : cout) * 4);
        b.prep = (long long)off; off = al256(off + (size_t)v100_ir_prep_bytes(sh));
        const size_t fw = (size_t)v100_ir_fwd_workspace_bytes(sh), bw = (size_t)v100_ir_bwd_workspace_bytes(sh);
        if (fw > fwd_ws) fwd_ws = fw;
        if (bw > bws) bws = bw;
        const size_t dxb = (size_t)B * cin * T * 4;
        if (dxb > dxmax) dxmax = dxb;
        b.grad_floats = (size_t)hid * cin + 2 * (size_t)hid + (size_t)hid * k + 2 * (size_t)hid + (size_t)cout * hid + 2 * (size_t)cout;
        grad_floats += b.grad_floats;
        T = T2;
    }
    fwd_ws_off = off;
    blob_bytes = al256(off + fwd_ws) + 256;
    bwd_ws = al256(bws) + 2 * al256(dxmax) + 256;
    return 1;
}
}   // namespace

extern "C" long long v100_ir_stack_plan(const int* desc, long long* plan) {
    if (!desc || !plan) return -1;
    StackBlock blk[32];
    size_t blob, bws, gf, fwo;
    if (!stack_layout(desc, blk, blob, bws, gf, fwo)) return -1;
    const int n = desc[ST_N];
    for (int i = 0; i < n; ++i) {
        long long* o = plan + 8 * i;
        o[0] = blk[i].a1; o[1] = blk[i].a2; o[2] = blk[i].a3; o[3] = blk[i].y; o[4] = blk[i].y16; o[5] = blk[i].coef; o[6] = blk[i].prep;
        o[7] = blk[i].Tout;
    }
    long long* t = plan + 8 * n;
    t[0] = (long long)blob; t[1] = (long long)bws; t[2] = (long long)gf; t[3] = (long long)fwo; t[4] = 0; t[5] = 0;
    return (long long)blob;
}

// x16: bf16 shadow of x written by whatever produced x (NULL: the first expand GEMM reads the fp32 x)
extern "C" int v100_ir_stack_fwd_train(const int* desc, const void* const* params, const void* x, const void* x16, void* blob,
                                       void* stream) {
    if (!desc || !params || !x || !blob) return V100_ERR_NULL;
    StackBlock blk[32];
    size_t blob_bytes, bws, gf, fwo;
    if (!stack_layout(desc, blk, blob_bytes, bws, gf, fwo)) return V100_ERR_SHAPE;
    const int n = desc[ST_N];
    char* base = (char*)blob;
    int rc;
    {   // bf16 / transposed weight copies of every block: one launch
        int shapes[32 * IR_NSHAPE];
        const void* w1s[32]; const void* w3s[32]; void* preps[32];
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < IR_NSHAPE; ++j) shapes[i * IR_NSHAPE + j] = blk[i].sh[j];
            w1s[i] = params[18 * i + 0]; w3s[i] = params[18 * i + 12]; preps[i] = base + blk[i].prep;
        }
        CK(v100_ir_prep_batched(shapes, w1s, w3s, preps, n, stream));
    }
    const void* xin = x;
    const void* xin16 = x16;
    for (int i = 0; i < n; ++i) {
        const StackBlock& b = blk[i];
        const void* P[28];
        P[0] = xin;
        for (int j = 0; j < 18; ++j) P[1 + j] = params[18 * i + j];
        P[19] = base + b.a1; P[20] = base + b.a2; P[21] = base + b.a3; P[22] = base + b.y; P[23] = base + b.coef;
        P[24] = base + fwo; P[25] = base + b.prep;
        P[26] = xin16; P[27] = b.y16 >= 0 ? base + b.y16 : nullptr;
        CK(v100_ir_fwd_train(b.sh, P, stream));
        xin = base + b.y;
        xin16 = b.y16 >= 0 ? base + b.y16 : nullptr;
    }
    return V100_OK;
}

// dy: gradient of the last block's output; dx: gradient of x (NULL = not needed); grads: grad_floats floats, per block
// dW1 dg1 db1 dWd dg2 db2 dW3 dg3 db3; ws: bwd_ws_bytes (plan)
extern "C" int v100_ir_stack_bwd(const int* desc, const void* const* params, const void* x, const void* x16, const void* blob,
                                 const void* dy, void* dx, void* grads, void* ws, void* stream) {
    if (!desc || !params || !x || !blob || !dy || !grads || !ws) return V100_ERR_NULL;
    StackBlock blk[32];
    size_t blob_bytes, bws, gf, fwo;
    if (!stack_layout(desc, blk, blob_bytes, bws, gf, fwo)) return V100_ERR_SHAPE;
    const int n = desc[ST_N];
    const char* base = (const char*)blob;
    size_t wsmax = 0, dxmax = 0;
    for (int i = 0; i < n; ++i) {
        const size_t bw = (size_t)v100_ir_bwd_workspace_bytes(blk[i].sh);
        if (bw > wsmax) wsmax = bw;
        const size_t dxb = (size_t)blk[i].sh[IR_B] * blk[i].sh[IR_CIN] * blk[i].sh[IR_T] * 4;
        if (dxb > dxmax) dxmax = dxb;
    }
    char* w0 = (char*)ws;
    char* pp[2] = {w0 + al256(wsmax), w0 + al256(wsmax) + al256(dxmax)};
    size_t goff = gf;
    const void* dyi = dy;
    int rc;
    // Round 6: at level 5 the GRADIENT stream between two blocks is bf16 too (what autograd hands back for a bf16 residual stream under
    // the reference's autocast) wherever both sides can take it: the producer a block on the finished-gradient path (its expand
    // backward-data GEMM adds dy, if it has a residual, and stores dx through the 16-bit epilogue), the consumer a block whose one-pass
    // BatchNorm-3 backward covers the shape.  Half the bytes of the three passes that touch it (dx store, dy in BatchNorm-3 backward, dy as the
    // residual gradient).  V100_IR_GRAD16=0: the fp32 stream (A/B).
    static const bool grad16_on = [] { const char* e = getenv("V100_IR_GRAD16"); return !(e && e[0] == '0'); }();
    auto grad16_ok = [&](int i) {                          // block i can consume a 16-bit dy AND produce a 16-bit dx
        const int* sh = blk[i].sh;
        return grad16_on && desc[ST_LEVEL] >= 5 && sh[IR_ACT16] >= 3 && sh[IR_STRIDE] == 1 && sh[IR_T] >= 8 &&
               IR_FUSE_BN3 >= 2 && chan_bn3_bwd_fits(sh[IR_B], sh[IR_T]) && v100_ir_act16_supported(sh) &&
               dw_bwd_da1_supported(sh[IR_B], sh[IR_HID], sh[IR_T], sh[IR_K], v100_dw_num_groups(sh[IR_B], sh[IR_HID])) && sh[IR_ACT16] >= 2;
    };
    bool dy_is16 = false;
    for (int i = n - 1; i >= 0; --i) {
        StackBlock b = blk[i];
        // dx of block i is dy of block i - 1: 16-bit when both are capable blocks (with or without a residual)
        const bool dx_is16 = i > 0 && grad16_ok(i) && grad16_ok(i - 1);
        if (dy_is16 && !grad16_ok(i)) return V100_ERR_SHAPE;      // (cannot happen: dy_is16 was decided with this block's capability)
        b.sh[IR_PREPPED] |= (dy_is16 ? 16 : 0) | (dx_is16 ? 32 : 0);
        goff -= b.grad_floats;
        float* g = (float*)grads + goff;
        const int cin = b.sh[IR_CIN], hid = b.sh[IR_HID], cout = b.sh[IR_COUT], k = b.sh[IR_K];
        void* dxi = i > 0 ? (void*)pp[i & 1] : dx;
        const void* P[25];
        P[0] = i > 0 ? (const void*)(base + blk[i - 1].y) : x;
        P[1] = base + b.a1; P[2] = base + b.a2; P[3] = base + b.a3;
        P[4] = params[18 * i + 0]; P[5] = params[18 * i + 6]; P[6] = params[18 * i + 12];
        P[7] = params[18 * i + 1]; P[8] = params[18 * i + 7]; P[9] = params[18 * i + 13];
        P[10] = base + b.coef; P[11] = dyi; P[12] = dxi;
        float* q = g;
        P[13] = q; q += (size_t)hid * cin; P[14] = q; q += hid; P[15] = q; q += hid;
        P[16] = q; q += (size_t)hid * k; P[17] = q; q += hid; P[18] = q; q += hid;
        P[19] = q; q += (size_t)cout * hid; P[20] = q; q += cout; P[21] = q;
        P[22] = w0; P[23] = base + b.prep;
        P[24] = i > 0 ? (blk[i - 1].y16 >= 0 ? (const void*)(base + blk[i - 1].y16) : nullptr) : x16;
        CK(v100_ir_bwd(b.sh, P, stream));
        dyi = dxi;
        dy_is16 = dx_is16;
    }
    return V100_OK;
}
