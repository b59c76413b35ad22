// K2: depthwise Conv1d (groups == channels) for gfx950, fp32 activations in [B, C, T].
//
// Replaces nn.Conv1d(groups=hidden) + nn.BatchNorm1d + nn.ReLU6 of the reference's
// ConvBNActivate "dw" stage (voice100/models/asr.py:27-37, 49) and, with flipped taps,
// its backward-data; dwconv_wgrad is its backward-weight.
//
// Design (DESIGN.md "K2"): HBM-bound for k <~ 51, at the fp32-VALU/HBM balance point for
// k >= 59.  One workgroup = one channel x a group of batch rows, so the taps are wave-uniform.
// Each wave streams (row, tile) items: a tile is 64 lanes x R consecutive outputs; the input
// span (+halo) is staged once through a per-wave LDS row with the producer's BatchNorm affine +
// ReLU6 (or the BN-backward affine of two tensors) applied on the way in, then every lane slides
// a register window over it: R*K v_fmac per lane against ~(R*S+K)/4 window + K/4 tap
// ds_read_b128 (taps are broadcast reads streamed next to the window, so the register footprint
// does not grow with K).  The next tile's global loads are in flight while the current tile
// computes.  Per-channel sums for the consumer BatchNorm (training statistics, or the two
// BN-backward reductions) are accumulated in registers and written once per workgroup to a
// [G][C][2] slab -- deterministic, no atomics.
#pragma once
#include "common.h"
#include "timing.h"
#include <stdlib.h>

enum { DW_IN_NONE = 0, DW_IN_AFFINE_RELU6 = 1, DW_IN_AFFINE2 = 2 };
enum { DW_OUT_RAW_STATS = 0, DW_OUT_AFFINE_RELU6 = 1, DW_OUT_MASK_STATS = 2, DW_OUT_RAW = 3 };

// BatchNorm finalisation by the PRODUCING depthwise kernel (G == 1 only: the workgroup of channel c then owns c's complete sums,
// so the one-wave-per-channel finaliser launch -- ~5 us of latency between two dependent kernels, 54 times a step -- is not needed).
// Same arithmetic as bn_finalize_train_kernel / bn_bwd_finalize_kernel (double precision on the one float partial): identical results.
struct DwFin {
    int mode;               // 0 none; 1 training statistics -> scale / shift (+ saved mean / rstd, running stats); 2 backward -> p, q, r (+ dgamma, dbeta)
    double count;           // B * T
    const float* gamma;
    const float* a;         // mode 1: beta;      mode 2: saved mean
    const float* b;         // mode 2: saved rstd
    float* o0; float* o1; float* o2;    // mode 1: scale, shift, -;          mode 2: p, q, r
    float* o3; float* o4;               // mode 1: save_mean, save_rstd;     mode 2: dgamma, dbeta
    float* running_mean; float* running_var; long long* num_batches_tracked;
    float momentum, eps;
};

// ... or by the CONSUMING kernel, from a producer's slab of partial sums (dw_finalize_parts below)
struct DwPre {
    DwFin f;                // mode 0: none
    const float* stats;     // [parts][C][2]
    int parts;
};

struct DwParams {
    const float* x;      // [B,C,Tin]
    const float* x2;     // [B,C,Tin]   second stream for DW_IN_AFFINE2
    const float* w;      // [C,K]
    const float* in_a;   // [C] scale (AFFINE_RELU6) / p (AFFINE2)
    const float* in_b;   // [C] shift            / q
    const float* in_c;   // [C]                  / r
    float* y;            // [B,C,Tout]
    const float* aux;    // [B,C,Tout]  pre-activation tensor for DW_OUT_MASK_STATS
    const float* out_a;  // [C]
    const float* out_b;  // [C]
    float* stats;        // [G][C][2]
    int B, C, Tin, Tout, K, stride, pad, flip, upsample, G, in_mode, out_mode;
    float* wpartial;     // [G][C][K]  backward-weight partial sums of the fused backward kernel (else null)
    // 16-bit storage of the big hidden tensors ("act16", bf16 mode): DW_IO_* mask -- those tensors are bf16 [B][C][P], row
    // pitch P = dw_pitch16(T) (multiple of 8 elements); x / x2 / aux / y then point at bf16 data.  MFMA kernels only.
    int io16;
    DwFin fin;           // mode 0 unless the caller asked for in-kernel finalisation (G == 1)
    DwPre pre;           // f.mode 0 unless the kernel also finalises the BatchNorm whose coefficients it applies on load (G == 1): then
                         // in_a / in_b / in_c are ignored and the coefficients come from pre (and are written to pre.f.o0 .. o2)
    // Channel-major storage ("cm", streaming kernels only): the tensors are [C][B][P] -- a channel's B rows contiguous, which is the
    // order this kernel walks them in (one workgroup per channel) and, seen from the 1x1 GEMMs, ONE [C x (B P)] matrix whose columns
    // are all utterances back to back -- instead of [B][C][P].  Row (b, c) then starts at (c B + b) P.
    int cm;
    // Segment packing (eval-mode streaming kernel only): rows much shorter than a wave item's 256 NS positions (1-second chunks: 51
    // outputs) are laid side by side in ONE LDS image, `segn` utterances per item, `segs` image positions apart (a multiple of 16, >= P +
    // pad so that the zeros between two rows are each row's own zero padding).  0 / 1: one row per item.
    int segn, segs;
    // fused backward, streaming kernel, one group, <= 32 batch rows (dw_bwd_da1_supported): the kernel writes the FINISHED BatchNorm-1
    // backward gradient da1 = p dz1 + q a1 + r (coefficients it finalises itself: fin.mode == 2) where it otherwise writes dz1
    int da1;
};
__device__ __forceinline__ unsigned dw_row_index(const DwParams& p, int b, int c) {
    return p.cm ? (unsigned)(c * p.B + b) : (unsigned)(b * p.C + c);
}
enum { DW_IO_X = 1, DW_IO_X2 = 2, DW_IO_AUX = 4, DW_IO_Y = 8 };

// one lane, with the channel's complete sums in double: the arithmetic of bn_finalize_train_kernel / bn_bwd_finalize_kernel.
// coef (may be null) receives the three per-channel coefficients the consumer applies: mode 1 (scale, shift, 0), mode 2 (p, q, r).
// BatchNorm backward (mode 2) from the channel's sums s0 = sum(dz), s1 = sum(dz * x) with the three per-channel inputs already in hand
// (saved mean, rstd, gamma): the ONE place that arithmetic lives -- dw_finalize_d loads them and calls this; a kernel that has to store
// data behind the result (the fused depthwise backward's finished gradient) fetches them at its start instead of at its end.
__device__ __forceinline__ void dw_finalize_bwd_pre(const DwFin& f, int c, double s0, double s1, float mu_f, float rs_f, float ga_f, float* coef) {
    const double mu = mu_f, rs = rs_f, ga = ga_f;
    const double dg = rs * (s1 - mu * s0);
    const double pp = ga * rs;
    const double qq = -ga * rs * rs * dg / f.count;
    const double rr = -pp * s0 / f.count - qq * mu;
    f.o0[c] = (float)pp;
    f.o1[c] = (float)qq;
    f.o2[c] = (float)rr;
    if (f.o3) f.o3[c] = (float)dg;
    if (f.o4) f.o4[c] = (float)s0;
    if (coef) { coef[0] = (float)pp; coef[1] = (float)qq; coef[2] = (float)rr; }
}

// BatchNorm training statistics (mode 1) from the channel's sums with gamma, beta and the running statistics already in hand (see
// dw_finalize_bwd_pre: same reason -- a consumer that finalises in front of its first row requests them at kernel entry)
__device__ __forceinline__ void dw_finalize_fwd_pre(const DwFin& f, int c, double s0, double s1, float ga_f, float be_f, float rm_f, float rv_f,
                                                    float* coef) {
    if (c == 0 && f.num_batches_tracked) *f.num_batches_tracked += 1;
    const double mean = s0 / f.count;
    double var = s1 / f.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float sc = ga_f * rstd;
    const float sh = be_f - (float)mean * sc;
    f.o0[c] = sc;
    f.o1[c] = sh;
    if (f.o3) f.o3[c] = (float)mean;
    if (f.o4) f.o4[c] = rstd;
    if (f.running_mean) {
        const double unbiased = f.count > 1.0 ? var * (f.count / (f.count - 1.0)) : var;
        f.running_mean[c] = (1.f - f.momentum) * rm_f + f.momentum * (float)mean;
        f.running_var[c] = (1.f - f.momentum) * rv_f + f.momentum * (float)unbiased;
    }
    if (coef) { coef[0] = sc; coef[1] = sh; coef[2] = 0.f; }
}

__device__ __forceinline__ void dw_finalize_d(const DwFin& f, int c, double s0, double s1, float* coef) {
    if (f.mode == 1) {
        dw_finalize_fwd_pre(f, c, s0, s1, f.gamma[c], f.a[c], f.running_mean ? f.running_mean[c] : 0.f, f.running_mean ? f.running_var[c] : 0.f, coef);
    } else if (f.mode == 2) {
        dw_finalize_bwd_pre(f, c, s0, s1, f.a[c], f.b[c], f.gamma[c], coef);
    }
}
// thread 0 of the workgroup of channel c, with the channel's complete sums (G == 1)
__device__ __forceinline__ void dw_finalize(const DwFin& f, int c, float sum0, float sum1) { dw_finalize_d(f, c, (double)sum0, (double)sum1, nullptr); }

// BatchNorm finalisation by the CONSUMING kernel from a producer's slab of partial sums [parts][C][2] (a GEMM epilogue's): for
// kernels that run ONE workgroup per channel.  Executed by one whole wave (EXEC all ones) before the workgroup's first barrier:
// lanes stride over the slab rows, butterfly in double -- the finaliser kernels' own order, so the results are identical.
__device__ __forceinline__ void dw_finalize_parts(const DwPre& pre, int C, int c, int lane, float* coef) {
    double s0 = 0.0, s1 = 0.0;
    for (int g = lane; g < pre.parts; g += 64) {
        s0 += (double)pre.stats[((size_t)g * C + c) * 2 + 0];
        s1 += (double)pre.stats[((size_t)g * C + c) * 2 + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if (lane == 0) dw_finalize_d(pre.f, c, s0, s1, coef);
}

// blockIdx -> channel for kernels that run ONE workgroup per channel over [B][C][T] rows: workgroups are dealt to the 8 XCDs round-robin;
// groups of GR consecutive channels go to one XCD (the groups round-robin), so an XCD sweeps GR-row runs of every utterance and the
// channels that share a 128-byte line of a per-channel array ([parts][C][2] slabs, BatchNorm parameters) share an L2.  Identity when
// C is not a multiple of 8 GR.  (Round 6: depthwise_stream16.h dws_chan, measured there; bn.hip's block-boundary passes.)
template <int GR>
__device__ __forceinline__ int v100_chan_of_block(int bid, int C) {
    if (C & (8 * GR - 1)) return bid;
    const int xcd = bid & 7, idx = bid >> 3;
    return ((idx / GR) * 8 + xcd) * GR + (idx & (GR - 1));
}

// dw_finalize_parts with its loads issued AHEAD (streaming depthwise kernels, round 6).  Inside the kernel the finalisation used to sit
// behind the row requests: its slab reads and per-channel parameter reads were YOUNGER than the wave's first rows, vmcnt retires in
// order, so wave 0 waited a full HBM round trip for rows it did not need yet and three waves waited for wave 0 at the barrier
// (IR_FUSE_PRE cost the forward kernel 10 %).  dw_pre_issue requests everything at kernel entry, in front of the rows, with no branch
// (inactive waves / modes load through an out-of-range buffer offset -- no bytes move -- or from a harmless valid address);
// dw_pre_finish is pure arithmetic on those registers: same values, same order of summation as dw_finalize_parts (parts past the slab
// read as 0.0 and add nothing), so the coefficients are bit-identical.  Up to DW_PRE_MAXPARTS partial sums (4 per lane).
#define DW_PRE_MAXPARTS 256
struct DwPreRegs { float s[4][2]; float ga, a, b, rm, rv; };
__device__ __forceinline__ void dw_pre_issue(const DwPre& pre, int C, int c, int lane, bool active, const float* safe, DwPreRegs& r) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pre.stats), 0,
                                                                        active ? (int)((size_t)pre.parts * C * 8) : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = lane + 64 * i;
        const unsigned off = (unsigned)(((size_t)g * C + c) * 8);
        const f32x2_ v = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(rs, (active && g < pre.parts) ? (int)off : 0x7ffffff0, 0, 0));
        r.s[i][0] = v[0]; r.s[i][1] = v[1];
    }
    const DwFin& f = pre.f;
    r.ga = *((active && f.gamma) ? f.gamma + c : safe);
    r.a = *((active && f.a) ? f.a + c : safe);
    r.b = *((active && f.b) ? f.b + c : safe);
    r.rm = *((active && f.running_mean) ? f.running_mean + c : safe);
    r.rv = *((active && f.running_var) ? f.running_var + c : safe);
}
__device__ __forceinline__ void dw_pre_finish(const DwPre& pre, int c, int lane, const DwPreRegs& r, float* coef) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s0 += (double)r.s[i][0]; s1 += (double)r.s[i][1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if (lane == 0) {
        if (pre.f.mode == 1) dw_finalize_fwd_pre(pre.f, c, s0, s1, r.ga, r.a, r.rm, r.rv, coef);
        else dw_finalize_bwd_pre(pre.f, c, s0, s1, r.a, r.b, r.ga, coef);
    }
}

int chan_affine2_fin(const void* u, const void* v, float* out, void* shadow, int B, int C, int T, int u_bf16, const DwPre& pre, void* stream,
                     int v_bf16 = 0);   // bn.hip
int chan_bn3_bwd(const void* u, const void* v, float* partial, void* out, int B, int C, int T, const DwFin& fin, void* stream, int u_bf16 = 0);   // bn.hip; 0 = not covered
int chan_bn3_bwd_fits(int B, int T);     // 1 when chan_bn3_bwd covers the shape
int chan_reduce2_io_fin(const void* u, const void* v, float* partial, int B, int C, int T, const DwFin& fin, void* stream);   // bn.hip
// the act16 entry points with in-kernel BatchNorm finalisation (block executor; fin.mode != 0 requires G == 1)
int dw_fwd_train_io_fin(const void* a1, const float* w, const float* in_a, const float* in_b, void* a2, float* stats, int G, int B, int C,
                        int T, int K, int io16, const DwFin& fin, const DwPre& pre, void* stream);
int dw_fwd_eval_io(const void* h1, const float* w, const float* out_a, const float* out_b, void* h2, int B, int C, int T, int K, void* stream,
                   int cm = 0, int f16 = 0);
int dw_bwd_io_fin(const void* g, const void* g2, const float* w, const float* ga, const float* gb, const float* gc, const void* xpre,
                  const float* xa, const float* xb, void* dxin, float* stats, float* wpartial, float* dw, int G, int B, int C, int T, int K,
                  int io16, const DwFin& fin, const DwPre& pre, void* stream, int da1 = 0);
bool dw_bwd_da1_supported(int B, int C, int T, int K, int G);      // depthwise_bwd_fused16g.hip
__host__ __device__ __forceinline__ int dw_pitch16(int T, int B) { return v100_pitch16(T, B); }

struct DwWgradParams {
    const float* g;      // [B,C,Tout] upstream gradient stream 1
    const float* g2;     // [B,C,Tout] stream 2 for AFFINE2
    const float* ga; const float* gb; const float* gc;   // [C] each
    const float* x;      // [B,C,Tin]  conv input (pre-activation when x_mode = AFFINE_RELU6)
    const float* xa; const float* xb;                     // [C] each
    float* partial;      // [G][C][K]
    int B, C, Tin, Tout, K, stride, pad, G, g_mode, x_mode;
};

__device__ __forceinline__ float dw_in_transform(int mode, float v, float v2, float a, float b, float c) {
    if (mode == DW_IN_AFFINE_RELU6) return relu6f(fmaf(v, a, b));
    if (mode == DW_IN_AFFINE2) return fmaf(v, a, fmaf(v2, b, c));
    return v;
}

// Branch-free loads: a load inside a conditional makes hipcc wait for it at the join, serialising a
// tile's loads one latency after another.  Every load below is unconditional (address clamped to
// the row start when out of range), the value is selected afterwards.

// R consecutive floats ptr[0..R-1] (positions t0..t0+R-1 of a row of length T), zero past the end.
template <int R, bool AL>
__device__ __forceinline__ void dw_load_run(float (&out)[R], const float* __restrict__ ptr, int t0, int T) {
    if constexpr (AL) {
#pragma unroll
        for (int q = 0; q < R / 4; ++q) {
            const bool ok = t0 + 4 * q < T;                  // first element in range: the 16-byte load is safe to issue
            const f32x4 v = *reinterpret_cast<const f32x4u*>(ok ? ptr + 4 * q : ptr - t0);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[4 * q + e] = (t0 + 4 * q + e < T) ? v[e] : 0.f;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = t0 + r < T;
            const float v = ptr[ok ? r : -t0];
            out[r] = ok ? v : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Staging: one wave copies the input span of one tile into its LDS row.
//   LDS index i  <->  input position in0 + i,   in0 = out0*S - pad (may be negative)
// Global reads are 16-byte aligned float4 when the row length is a multiple of 4 (AL).
template <int NV, bool TWO>
struct DwRaw {
    f32x4 v[NV];
    f32x4 v2[TWO ? NV : 1];
};

typedef unsigned int dw_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dw_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// Buffer loads over the whole [B,C,T] tensor: the row offset is a scalar (soffset), the in-row offset a
// per-lane voffset; positions outside the row's [0, Tin) window get an out-of-range voffset and come back
// as zero from the hardware bounds check (no select, no fault at the tensor's end, any row alignment).
template <int NV, int SPAN, bool TWO, bool AL>
__device__ __forceinline__ void dw_issue_loads(DwRaw<NV, TWO>& raw, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rx2,
                                               unsigned row_bytes, int in0, int Tin, int lane) {
    const int in0a = in0 & ~3;            // floor to a multiple of 4 (two's complement: also for negatives)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int ia = in0a + 4 * (lane + 64 * v);
        const bool ok = ia >= 0 && ia < Tin && ia < in0 + SPAN;
        const int vo = ok ? ia * 4 : 0x7ffffff0;
        raw.v[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, (int)row_bytes, 0));
        if constexpr (TWO) raw.v2[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx2, vo, (int)row_bytes, 0));
    }
}

template <int NV, int SPAN, int MODE, bool TWO>
__device__ __forceinline__ void dw_stage_to_lds(const DwRaw<NV, TWO>& raw, float* lds, int in0, int Tin,
                                                float ca, float cb, float cc, int lane) {
    const int in0a = in0 & ~3;
    const int off = in0 - in0a;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int ia = in0a + 4 * (lane + 64 * v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = ia + e;
            const int i = 4 * (lane + 64 * v) + e - off;
            float val;
            if constexpr (MODE == DW_IN_AFFINE_RELU6) val = relu6f(fmaf(raw.v[v][e], ca, cb));
            else if constexpr (MODE == DW_IN_AFFINE2) val = fmaf(raw.v[v][e], ca, fmaf(raw.v2[v][e], cb, cc));
            else val = raw.v[v][e];
            val = (t >= 0 && t < Tin) ? val : 0.f;      // zero padding applies to the TRANSFORMED tensor
            if (i >= 0 && i < SPAN) lds[i] = val;
        }
    }
}

template <int K, int S, int R>
struct DwGeom {
    static_assert((R * S) % 4 == 0, "lane window must start on a 16-byte LDS boundary");
    static constexpr int TILE = 64 * R;
    static constexpr int WIN = (R - 1) * S + K;               // inputs one lane touches
    static constexpr int SPAN = (TILE - 1) * S + K;           // inputs one tile touches
    static constexpr int SPAN4 = (SPAN + 3) & ~3;
    static constexpr int NV = (SPAN + 3 + 3) / 4 / 64 + 1;    // float4 loads per lane covering [in0a, in0+SPAN)
    static constexpr int NCH = (WIN + 3) / 4;                 // window chunks
    static constexpr int NTC = (K + 3) / 4;                   // tap chunks
    static constexpr int PD = 2;                              // LDS prefetch distance (chunks)
};


// ---------------------------------------------------------------------------------------------
// Forward / backward-data kernel.  K, stride S, outputs-per-lane R, the input/output modes and the
// alignment class (AL: Tin and Tout multiples of 4 -> float4 global accesses) are compile time, so
// the window walk is fully unrolled with static register indices and the staging is branch-free.
//
// WG (fused backward, stride 1): the same pass also accumulates the backward-weight of the forward conv.  In the
// backward-data call the window holds the upstream gradient g' and each lane already loads the forward conv's
// pre-activation input a1 at its R positions (for the ReLU6 mask); with xin = relu6(bn1(a1)),
//     dxin[u] += wf[j'] * g'[u - pad' + j']      and      dWf[j'] += xin[u] * g'[u - pad' + j']
// run over the SAME (position, tap, window element) triples (wf = taps flipped, dW[j] = dWf[K-1-j]), so the second
// product costs one more v_fmac per triple and no memory traffic at all -- the stand-alone backward-weight kernel
// re-reads all three tensors.  K more accumulators per lane, reduced once per workgroup.
template <int K, int S, int R, int IM, int OM, bool AL, bool WG = false>
__global__ __launch_bounds__(256) void dwconv_kernel(DwParams p) {
    static_assert(!WG || (S == 1 && OM == DW_OUT_MASK_STATS), "fused backward-weight rides on the stride-1 backward-data pass");
    using G_ = DwGeom<K, S, R>;
    constexpr int TILE = G_::TILE, WIN = G_::WIN, SPAN = G_::SPAN, NV = G_::NV, NCH = G_::NCH, NTC = G_::NTC, PD = G_::PD;
    constexpr bool TWO = IM == DW_IN_AFFINE2;
    constexpr bool STATS = (OM == DW_OUT_RAW_STATS || OM == DW_OUT_MASK_STATS);

    __shared__ __attribute__((aligned(16))) float lds_all[4][G_::SPAN4 + 8];
    __shared__ __attribute__((aligned(16))) float lds_w[NTC * 4];
    __shared__ float lds_red[4][2];
    __shared__ float lds_wred[WG ? 4 : 1][WG ? K : 1];

    const int c = blockIdx.x;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* lds = lds_all[wave];

    // taps -> LDS (zero padded to a multiple of 4); streamed through registers next to the window
    for (int j = threadIdx.x; j < NTC * 4; j += 256)
        lds_w[j] = j < K ? p.w[(size_t)c * K + (p.flip ? (K - 1 - j) : j)] : 0.f;
    __syncthreads();

    float ca = 1.f, cb = 0.f, cc = 0.f, oa = 1.f, ob = 0.f;
    if constexpr (IM != DW_IN_NONE) { ca = p.in_a[c]; cb = p.in_b[c]; }
    if constexpr (IM == DW_IN_AFFINE2) cc = p.in_c[c];
    if constexpr (OM == DW_OUT_AFFINE_RELU6 || OM == DW_OUT_MASK_STATS) { oa = p.out_a[c]; ob = p.out_b[c]; }

    const int Tin = p.Tin, Tout = p.Tout;
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int ntiles = (Tout + TILE - 1) / TILE;

    float s0 = 0.f, s1 = 0.f;
    float accw[WG ? K : 1];
#pragma unroll
    for (int j = 0; j < (WG ? K : 1); ++j) accw[j] = 0.f;
    DwRaw<NV, TWO> raw;
    const unsigned xbytes = (unsigned)((size_t)p.B * p.C * Tin * 4);
    const __amdgpu_buffer_rsrc_t rx = dw_make_rsrc(p.x, xbytes);
    const __amdgpu_buffer_rsrc_t rx2 = dw_make_rsrc(TWO ? p.x2 : p.x, xbytes);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // provably wave-uniform row index
    // Tile outer, batch row inner: everything that depends only on the tile (bounds masks, LDS
    // indices, in-row offsets) is loop-invariant for the inner loop and hoisted out of it; the
    // per-row work is loads + transform + select + ds_write.
    for (int tile = 0; tile < ntiles; ++tile) {
    const int out0 = tile * TILE;
    const int in0 = out0 * S - p.pad;
    int bi = wave_u;
    if (bi < nb) {
        const unsigned rb = (unsigned)(((size_t)(b0 + bi) * p.C + c) * Tin * 4);
        dw_issue_loads<NV, SPAN, TWO, AL>(raw, rx, rx2, rb, in0, Tin, lane);
    }
    for (; bi < nb; bi += 4) {
        const int b = b0 + bi;
        dw_stage_to_lds<NV, SPAN, IM, TWO>(raw, lds, in0, Tin, ca, cb, cc, lane);
        // Cross-lane hand-off through LDS inside one wave: the hardware runs a wave's LDS operations in order, but the
        // compiler sees one thread, finds no alias between this lane's stores and its window reads, and may swap
        // them (it did in dwconv_up2_bwd_kernel).  A compiler-only barrier keeps program order; it emits nothing.
        asm volatile("" ::: "memory");

        // prefetch the next row's input while this one computes
        if (bi + 4 < nb) {
            const unsigned rb = (unsigned)(((size_t)(b + 4) * p.C + c) * Tin * 4);
            dw_issue_loads<NV, SPAN, TWO, AL>(raw, rx, rx2, rb, in0, Tin, lane);
        }

        const int t0 = out0 + lane * R;
        const size_t oo = ((size_t)b * p.C + c) * Tout + t0;
        float auxv[R];
        if constexpr (OM == DW_OUT_MASK_STATS) dw_load_run<R, AL>(auxv, p.aux + oo, t0, Tout);

        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        float xin[WG ? R : 1];
        if constexpr (WG) {
#pragma unroll
            for (int r = 0; r < R; ++r) xin[r] = (t0 + r < Tout) ? relu6f(fmaf(auxv[r], oa, ob)) : 0.f;
        }
        const float* win = lds + lane * (R * S);
        // The tap reads are loop-invariant; hide that from LICM (an opaque zero offset per item) or
        // hipcc hoists all K taps into registers for the whole kernel.
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const float* wl = lds_w + opaque;
        // Software-pipelined walk: chunk ch of the window and of the taps are fetched PD steps ahead;
        // the sched_barrier stops hipcc from hoisting every ds_read to the top of the unrolled body
        // (WIN + K live registers, which halves occupancy for K >= 51).
        f32x4 inc[NCH];
        f32x4 tapc[NTC];
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            if (q < NCH) inc[q] = *reinterpret_cast<const f32x4*>(win + 4 * q);
            if (q < NTC) tapc[q] = *reinterpret_cast<const f32x4*>(wl + 4 * q);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch + PD < NCH) inc[ch + PD] = *reinterpret_cast<const f32x4*>(win + 4 * (ch + PD));
            if (ch + PD < NTC) tapc[ch + PD] = *reinterpret_cast<const f32x4*>(wl + 4 * (ch + PD));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * ch + e;
                if (i < WIN) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int j = i - r * S;
                        if (j >= 0 && j < K) {
                            acc[r] = fmaf(tapc[j >> 2][j & 3], inc[ch][e], acc[r]);
                            if constexpr (WG) accw[j] = fmaf(xin[r], inc[ch][e], accw[j]);
                        }
                    }
                }
            }
            // pin this chunk's FMAs in front of the barrier (pure ops otherwise sink below it)
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[r]));
            if constexpr (WG) {
#pragma unroll
                for (int d = 0; d < (R - 1) * S + 4; ++d) {
                    const int j = 4 * ch + 3 - d;
                    if (j >= 0 && j < K) asm volatile("" :: "v"(accw[j]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // epilogue
        float outv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool valid = t0 + r < Tout;
            float yv = acc[r];
            if constexpr (OM == DW_OUT_RAW_STATS) {
                if (valid) { s0 += yv; s1 = fmaf(yv, yv, s1); }
            } else if constexpr (OM == DW_OUT_AFFINE_RELU6) {
                yv = relu6f(fmaf(yv, oa, ob));
            } else if constexpr (OM == DW_OUT_MASK_STATS) {
                const float pre = fmaf(auxv[r], oa, ob);
                yv = (pre > 0.f && pre < 6.f) ? yv : 0.f;
                if (valid) { s0 += yv; s1 = fmaf(yv, auxv[r], s1); }
            }
            outv[r] = yv;
        }
        if constexpr (AL) {
#pragma unroll
            for (int q = 0; q < R / 4; ++q) {
                if (t0 + 4 * q + 3 < Tout) {
                    f32x4 o = {outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]};
                    *reinterpret_cast<f32x4u*>(p.y + oo + 4 * q) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (t0 + 4 * q + e < Tout) p.y[oo + 4 * q + e] = outv[4 * q + e];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (t0 + r < Tout) p.y[oo + r] = outv[r];
        }
        asm volatile("" ::: "memory");     // the next row's LDS stores stay behind this row's window reads
    }
    }   // tile

    if constexpr (WG) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float s = wave_sum_dpp_hi(accw[j]);
            if (lane == 63) lds_wred[wave][j] = s;
        }
        __syncthreads();
        // flipped taps here = forward taps K-1-j
        for (int j = threadIdx.x; j < K; j += 256)
            p.wpartial[((size_t)g * p.C + c) * K + (K - 1 - j)] = (lds_wred[0][j] + lds_wred[1][j]) + (lds_wred[2][j] + lds_wred[3][j]);
    }
    if constexpr (STATS) {
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) { lds_red[wave][0] = s0; lds_red[wave][1] = s1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float a = (lds_red[0][0] + lds_red[1][0]) + (lds_red[2][0] + lds_red[3][0]);
            const float b = (lds_red[0][1] + lds_red[1][1]) + (lds_red[2][1] + lds_red[3][1]);
            p.stats[((size_t)g * p.C + c) * 2 + 0] = a;
            p.stats[((size_t)g * p.C + c) * 2 + 1] = b;
        }
    }
}

#include "depthwise_mfma.h"

// Stride-1 layers with a specialised kernel size run on the Toeplitz-MFMA kernels (round 3: the register-window VALU kernels they
// replaced are no longer instantiated for those sizes -- they remain for the stride-2 opener and, as dwconv_generic_kernel, for
// everything else).  Precision knob, read once: V100_DW_DIGITS = 2 | 3 bf16 digits per fp32 tap / sample (default 3: fp32-exact
// products).  With 16-bit activation storage the taps default to DW_DIGITS16 (below).
// Round 6: with 16-bit activation storage the taps are ONE digit by default -- rounded to bf16 (fp16 at precision "fp16") like the data
// operand, which is what F.conv1d computes under the reference's 16-bit autocast (its weights are cast with its input); the second
// digit bought nothing the rounded data could show and cost a second pass over the matrix pipe per Toeplitz block (step -0.9 %, the
// depthwise family +0.02 of 8 TB/s, and power: profiles/r06_dw_digits_ab.txt).  V100_DW_DIGITS=3 restores fp32-exact taps.
#ifndef DW_DIGITS16
#define DW_DIGITS16 1
#endif
struct DwPathConfig { int digits; bool digits3; };
static inline DwPathConfig dw_path_config() {
    static const DwPathConfig cfg = [] {
        DwPathConfig c{3, false};
        const char* d = getenv("V100_DW_DIGITS");
        if (d && d[0] == '2') c.digits = 2;
        if (d && d[0] == '3') c.digits3 = true;
        return c;
    }();
    return cfg;
}

// kernel sizes used by the reference's networks: asr.py:68-76, tts.py:18-25, 73-76
#define V100_DW_SPECIALISED(X) X(5) X(7) X(11) X(17) X(19) X(27) X(29) X(33) X(35) X(51) X(59) X(65) X(67) X(75) X(83)

// One launcher per (input mode, output mode) pair, each in its own translation unit (they compile
// in parallel).  Returns false when (K, stride) has no specialisation.
#ifndef DW_FUSED_R
#define DW_FUSED_R 4
#endif
template <int IM, int OM, bool WG = false, int IO = 0>
static bool dw_launch_specialised(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl) {
    dim3 grid(p.C, p.G);
    {
        const DwPathConfig cfg = dw_path_config();
        if (p.stride == 1 && p.upsample == 1) {
#define X(KK)                                                                                                           \
    if (p.K == KK) {                                                                                                    \
        if constexpr (IO != 0) {                                                                                        \
            if (!cfg.digits3) V100_LAUNCH(tl, (dwconv_mfma_kernel<KK, IM, OM, DW_DIGITS16, WG, IO>), grid, dim3(256), 0, st, p); \
            else V100_LAUNCH(tl, (dwconv_mfma_kernel<KK, IM, OM, 3, WG, IO>), grid, dim3(256), 0, st, p);              \
        } else {                                                                                                        \
            if (cfg.digits == 2) V100_LAUNCH(tl, (dwconv_mfma_kernel<KK, IM, OM, 2, WG, IO>), grid, dim3(256), 0, st, p); \
            else V100_LAUNCH(tl, (dwconv_mfma_kernel<KK, IM, OM, 3, WG, IO>), grid, dim3(256), 0, st, p);              \
        }                                                                                                               \
        return true;                                                                                                    \
    }
            V100_DW_SPECIALISED(X)
#undef X
        }
    }
    if constexpr (IO != 0) return false;      // 16-bit storage exists on the MFMA kernels only
    else {
    const bool big = p.Tout > 256 && !(WG && DW_FUSED_R == 4);
    // rows of any length take the 16-byte (dword-aligned) global path; tails are masked per element
#define DW_GO(KK, SS)                                                                                             \
    do {                                                                                                          \
        if (big) V100_LAUNCH(tl, (dwconv_kernel<KK, SS, 8, IM, OM, true, WG>), grid, dim3(256), 0, st, p);        \
        else V100_LAUNCH(tl, (dwconv_kernel<KK, SS, 4, IM, OM, true, WG>), grid, dim3(256), 0, st, p);            \
        return true;                                                                                              \
    } while (0)
    if constexpr (!WG) {
        if (p.K == 11 && p.stride == 2) DW_GO(11, 2);
    }
#undef DW_GO
    return false;
    }
}

bool dw_launch_fwd_train16(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl);     // a1 in, a2 out stored as bf16
bool dw_launch_fwd_eval16(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl, bool f16 = false);      // eval: h1 in, h2 = relu6(bn2(conv)) out, bf16
bool dw_launch_bwd_fused16(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl);     // a2 (x2) and a1 (aux) stored as bf16
bool dw_launch_bwd_fused16g(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl);    // ... and dz2 in, dz1 out too
bool dw_launch_fwd_train(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl);   // in AFFINE_RELU6, out RAW_STATS
bool dw_launch_fwd_eval(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl);    // in NONE,         out AFFINE_RELU6
bool dw_launch_bwd_data(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl);    // in AFFINE2,      out MASK_STATS
bool dw_launch_bwd_fused(const DwParams& p, hipStream_t st, const V100TimedLaunch& tl);   // the same + backward-weight partial sums (stride 1)
