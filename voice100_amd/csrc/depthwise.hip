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
#include "common.h"

enum { DW_IN_NONE = 0, DW_IN_AFFINE_RELU6 = 1, DW_IN_AFFINE2 = 2 };
enum { DW_OUT_RAW_STATS = 0, DW_OUT_AFFINE_RELU6 = 1, DW_OUT_MASK_STATS = 2, DW_OUT_RAW = 3 };

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
};

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

// R consecutive floats starting at ptr (t0 .. t0+R-1 of a row of length T), zero past the end.
template <int R>
__device__ __forceinline__ void dw_load_run(float (&out)[R], const float* __restrict__ ptr, int t0, int T, bool vec) {
    if (vec) {
#pragma unroll
        for (int q = 0; q < R / 4; ++q) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t0 + 4 * q < T) v = *reinterpret_cast<const f32x4*>(ptr + 4 * q);
            out[4 * q] = v[0]; out[4 * q + 1] = v[1]; out[4 * q + 2] = v[2]; out[4 * q + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = (t0 + r < T) ? ptr[r] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// Staging: one wave copies the input span of one tile into its LDS row.
//   LDS index i  <->  input position in0 + i,   in0 = out0*S - pad (may be negative)
// Global reads are 16-byte aligned float4 when the row length is a multiple of 4.
template <int NV>
struct DwRaw {
    f32x4 v[NV];
    f32x4 v2[NV];
};

template <int NV, int SPAN>
__device__ __forceinline__ void dw_issue_loads(DwRaw<NV>& raw, const float* __restrict__ row, const float* __restrict__ row2,
                                               int in0, int Tin, bool two, int lane) {
    const int in0a = in0 & ~3;            // floor to a multiple of 4 (two's complement: also for negatives)
    const bool aligned = (Tin & 3) == 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int ia = in0a + 4 * (lane + 64 * v);
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (ia < in0 + SPAN) {
            if (aligned) {
                if (ia >= 0 && ia < Tin) {
                    a = *reinterpret_cast<const f32x4*>(row + ia);
                    if (two) b = *reinterpret_cast<const f32x4*>(row2 + ia);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = ia + e;
                    if (t >= 0 && t < Tin) {
                        a[e] = row[t];
                        if (two) b[e] = row2[t];
                    }
                }
            }
        }
        raw.v[v] = a;
        raw.v2[v] = b;
    }
}

template <int NV, int SPAN>
__device__ __forceinline__ void dw_stage_to_lds(const DwRaw<NV>& raw, float* lds, int in0, int Tin, int mode,
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
            float val = 0.f;
            if (t >= 0 && t < Tin) val = dw_in_transform(mode, raw.v[v][e], raw.v2[v][e], ca, cb, cc);
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
// Forward / backward-data kernel.  K, stride S and outputs-per-lane R are compile time so the
// window walk is fully unrolled with static register indices.
template <int K, int S, int R>
__global__ __launch_bounds__(256) void dwconv_kernel(DwParams p) {
    using G_ = DwGeom<K, S, R>;
    constexpr int TILE = G_::TILE, WIN = G_::WIN, SPAN = G_::SPAN, NV = G_::NV, NCH = G_::NCH, NTC = G_::NTC, PD = G_::PD;

    __shared__ __attribute__((aligned(16))) float lds_all[4][G_::SPAN4 + 8];
    __shared__ __attribute__((aligned(16))) float lds_w[NTC * 4];
    __shared__ float lds_red[4][2];

    const int c = blockIdx.x;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* lds = lds_all[wave];

    for (int j = threadIdx.x; j < NTC * 4; j += 256)
        lds_w[j] = j < K ? p.w[(size_t)c * K + (p.flip ? (K - 1 - j) : j)] : 0.f;
    __syncthreads();

    const int in_mode = p.in_mode, out_mode = p.out_mode;
    float ca = 1.f, cb = 0.f, cc = 0.f, oa = 1.f, ob = 0.f;
    if (in_mode != DW_IN_NONE) { ca = p.in_a[c]; cb = p.in_b[c]; }
    if (in_mode == DW_IN_AFFINE2) cc = p.in_c[c];
    if (out_mode == DW_OUT_AFFINE_RELU6 || out_mode == DW_OUT_MASK_STATS) { oa = p.out_a[c]; ob = p.out_b[c]; }

    const int Tin = p.Tin, Tout = p.Tout;
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int ntiles = (Tout + TILE - 1) / TILE;
    const int nitems = nb > 0 ? nb * ntiles : 0;
    const bool two = in_mode == DW_IN_AFFINE2;
    const bool out_vec = (Tout & 3) == 0;

    float s0 = 0.f, s1 = 0.f;
    DwRaw<NV> raw;
    int item = wave;
    if (item < nitems) {
        const int b = b0 + item / ntiles, tile = item % ntiles;
        const size_t ro = ((size_t)b * p.C + c) * Tin;
        dw_issue_loads<NV, SPAN>(raw, p.x + ro, two ? p.x2 + ro : p.x, tile * TILE * S - p.pad, Tin, two, lane);
    }
    for (; item < nitems; item += 4) {
        const int b = b0 + item / ntiles, tile = item % ntiles;
        const int out0 = tile * TILE;
        const int in0 = out0 * S - p.pad;
        dw_stage_to_lds<NV, SPAN>(raw, lds, in0, Tin, in_mode, ca, cb, cc, lane);

        // prefetch the next item's input while this one computes
        const int nitem = item + 4;
        if (nitem < nitems) {
            const int nb_ = b0 + nitem / ntiles, ntile = nitem % ntiles;
            const size_t ro = ((size_t)nb_ * p.C + c) * Tin;
            dw_issue_loads<NV, SPAN>(raw, p.x + ro, two ? p.x2 + ro : p.x, ntile * TILE * S - p.pad, Tin, two, lane);
        }

        const int t0 = out0 + lane * R;
        const size_t oo = ((size_t)b * p.C + c) * Tout + t0;
        float auxv[R];
        if (out_mode == DW_OUT_MASK_STATS) dw_load_run<R>(auxv, p.aux + oo, t0, Tout, out_vec);

        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
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
                        if (j >= 0 && j < K) acc[r] = fmaf(tapc[j >> 2][j & 3], inc[ch][e], acc[r]);
                    }
                }
            }
            // pin this chunk's FMAs in front of the barrier (pure ops otherwise sink below it)
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[r]));
            __builtin_amdgcn_sched_barrier(0);
        }

        // epilogue
        float outv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool valid = t0 + r < Tout;
            float yv = acc[r];
            if (out_mode == DW_OUT_RAW_STATS) {
                if (valid) { s0 += yv; s1 = fmaf(yv, yv, s1); }
            } else if (out_mode == DW_OUT_AFFINE_RELU6) {
                yv = relu6f(fmaf(yv, oa, ob));
            } else if (out_mode == DW_OUT_MASK_STATS) {
                const float pre = fmaf(auxv[r], oa, ob);
                yv = (pre > 0.f && pre < 6.f) ? yv : 0.f;
                if (valid) { s0 += yv; s1 = fmaf(yv, auxv[r], s1); }
            }
            outv[r] = yv;
        }
        if (out_vec) {
#pragma unroll
            for (int q = 0; q < R / 4; ++q) {
                if (t0 + 4 * q < Tout) {
                    f32x4 o = {outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(p.y + oo + 4 * q) = o;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (t0 + r < Tout) p.y[oo + r] = outv[r];
        }
    }

    if (out_mode == DW_OUT_RAW_STATS || out_mode == DW_OUT_MASK_STATS) {
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

// ---------------------------------------------------------------------------------------------
// Generic fallback: any K / stride / zero-upsampled input (backward-data of a strided conv).
// One thread per output element, taps in LDS, inputs straight from global/L2.  Correct, not fast;
// used only for shapes without a specialisation (e.g. L0's stride-2 backward-data, even K).
__global__ __launch_bounds__(256) void dwconv_generic_kernel(DwParams p) {
    extern __shared__ float gw[];
    __shared__ float red[4][2];
    const int c = blockIdx.x, g = blockIdx.y;
    const int K = p.K, S = p.stride, U = p.upsample;
    for (int j = threadIdx.x; j < K; j += 256) gw[j] = p.w[(size_t)c * K + (p.flip ? (K - 1 - j) : j)];
    __syncthreads();
    const int in_mode = p.in_mode, out_mode = p.out_mode;
    float ca = 1.f, cb = 0.f, cc = 0.f, oa = 1.f, ob = 0.f;
    if (in_mode != DW_IN_NONE) { ca = p.in_a[c]; cb = p.in_b[c]; }
    if (in_mode == DW_IN_AFFINE2) cc = p.in_c[c];
    if (out_mode == DW_OUT_AFFINE_RELU6 || out_mode == DW_OUT_MASK_STATS) { oa = p.out_a[c]; ob = p.out_b[c]; }
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int TinUp = (p.Tin - 1) * U + 1;
    float s0 = 0.f, s1 = 0.f;
    const long total = (long)(nb > 0 ? nb : 0) * p.Tout;
    for (long idx = threadIdx.x; idx < total; idx += 256) {
        const int b = b0 + (int)(idx / p.Tout), t = (int)(idx % p.Tout);
        const size_t ro = ((size_t)b * p.C + c) * p.Tin;
        float acc = 0.f;
        for (int j = 0; j < K; ++j) {
            const int u = t * S - p.pad + j;
            if (u < 0 || u >= TinUp || (u % U) != 0) continue;
            const int ti = u / U;
            const float v = dw_in_transform(in_mode, p.x[ro + ti], in_mode == DW_IN_AFFINE2 ? p.x2[ro + ti] : 0.f, ca, cb, cc);
            acc = fmaf(gw[j], v, acc);
        }
        const size_t oo = ((size_t)b * p.C + c) * p.Tout + t;
        if (out_mode == DW_OUT_RAW_STATS) { s0 += acc; s1 = fmaf(acc, acc, s1); }
        else if (out_mode == DW_OUT_AFFINE_RELU6) acc = relu6f(fmaf(acc, oa, ob));
        else if (out_mode == DW_OUT_MASK_STATS) {
            const float a = p.aux[oo];
            const float pre = fmaf(a, oa, ob);
            acc = (pre > 0.f && pre < 6.f) ? acc : 0.f;
            s0 += acc; s1 = fmaf(acc, a, s1);
        }
        p.y[oo] = acc;
    }
    if (out_mode == DW_OUT_RAW_STATS || out_mode == DW_OUT_MASK_STATS) {
        s0 = wave_sum(s0); s1 = wave_sum(s1);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            p.stats[((size_t)g * p.C + c) * 2 + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
            p.stats[((size_t)g * p.C + c) * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward-weight: dW[c][j] = sum_{b,t} g[b,c,t] * xin[b,c,t*S - pad + j].
// Same staging of xin (with its BN affine + ReLU6 recomputed on the way in); each lane keeps K
// partial sums in registers across all its tiles and the wave reduces them once at the end.
template <int K, int S, int R>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwWgradParams p) {
    using G_ = DwGeom<K, S, R>;
    constexpr int TILE = G_::TILE, WIN = G_::WIN, SPAN = G_::SPAN, NV = G_::NV, NCH = G_::NCH, PD = G_::PD;

    __shared__ __attribute__((aligned(16))) float lds_all[4][G_::SPAN4 + 8];
    __shared__ float lds_red[4][K];

    const int c = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* lds = lds_all[wave];
    const int x_mode = p.x_mode, g_mode = p.g_mode;
    float xa = 1.f, xb = 0.f, ga = 1.f, gb = 0.f, gc = 0.f;
    if (x_mode != DW_IN_NONE) { xa = p.xa[c]; xb = p.xb[c]; }
    if (g_mode != DW_IN_NONE) { ga = p.ga[c]; gb = p.gb[c]; }
    if (g_mode == DW_IN_AFFINE2) gc = p.gc[c];

    const int Tin = p.Tin, Tout = p.Tout;
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int ntiles = (Tout + TILE - 1) / TILE;
    const int nitems = nb > 0 ? nb * ntiles : 0;
    const bool g_vec = (Tout & 3) == 0;

    float accw[K];
#pragma unroll
    for (int j = 0; j < K; ++j) accw[j] = 0.f;

    DwRaw<NV> raw;
    int item = wave;
    if (item < nitems) {
        const int b = b0 + item / ntiles, tile = item % ntiles;
        const size_t ro = ((size_t)b * p.C + c) * Tin;
        dw_issue_loads<NV, SPAN>(raw, p.x + ro, p.x + ro, tile * TILE * S - p.pad, Tin, false, lane);
    }
    for (; item < nitems; item += 4) {
        const int b = b0 + item / ntiles, tile = item % ntiles;
        const int out0 = tile * TILE;
        const int in0 = out0 * S - p.pad;
        dw_stage_to_lds<NV, SPAN>(raw, lds, in0, Tin, x_mode, xa, xb, 0.f, lane);
        const int nitem = item + 4;
        if (nitem < nitems) {
            const int nb_ = b0 + nitem / ntiles, ntile = nitem % ntiles;
            const size_t ro = ((size_t)nb_ * p.C + c) * Tin;
            dw_issue_loads<NV, SPAN>(raw, p.x + ro, p.x + ro, ntile * TILE * S - p.pad, Tin, false, lane);
        }
        const int t0 = out0 + lane * R;
        const size_t oo = ((size_t)b * p.C + c) * Tout + t0;
        float gv[R], gv2[R];
        dw_load_run<R>(gv, p.g + oo, t0, Tout, g_vec);
        if (g_mode == DW_IN_AFFINE2) dw_load_run<R>(gv2, p.g2 + oo, t0, Tout, g_vec);
#pragma unroll
        for (int r = 0; r < R; ++r)
            gv[r] = (t0 + r < Tout) ? dw_in_transform(g_mode, gv[r], g_mode == DW_IN_AFFINE2 ? gv2[r] : 0.f, ga, gb, gc) : 0.f;

        const float* win = lds + lane * (R * S);
        f32x4 inc[NCH];
#pragma unroll
        for (int q = 0; q < PD; ++q)
            if (q < NCH) inc[q] = *reinterpret_cast<const f32x4*>(win + 4 * q);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch + PD < NCH) inc[ch + PD] = *reinterpret_cast<const f32x4*>(win + 4 * (ch + PD));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * ch + e;
                if (i < WIN) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int j = i - r * S;
                        if (j >= 0 && j < K) accw[j] = fmaf(gv[r], inc[ch][e], accw[j]);
                    }
                }
            }
            // pin this chunk's FMAs in front of the barrier (pure ops otherwise sink below it)
#pragma unroll
            for (int d = 0; d < (R - 1) * S + 4; ++d) {
                const int j = 4 * ch + 3 - d;
                if (j >= 0 && j < K) asm volatile("" :: "v"(accw[j]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const float s = wave_sum(accw[j]);
        if (lane == 0) lds_red[wave][j] = s;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += 256)
        p.partial[((size_t)g * p.C + c) * K + j] = (lds_red[0][j] + lds_red[1][j]) + (lds_red[2][j] + lds_red[3][j]);
}

// Generic backward-weight: one workgroup per (channel, group), threads stride over (b, t), taps looped.
__global__ __launch_bounds__(256) void dwconv_wgrad_generic_kernel(DwWgradParams p) {
    extern __shared__ float red[];     // [4][K]
    const int c = blockIdx.x, g = blockIdx.y;
    const int K = p.K, S = p.stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x_mode = p.x_mode, g_mode = p.g_mode;
    float xa = 1.f, xb = 0.f, ga = 1.f, gb = 0.f, gc = 0.f;
    if (x_mode != DW_IN_NONE) { xa = p.xa[c]; xb = p.xb[c]; }
    if (g_mode != DW_IN_NONE) { ga = p.ga[c]; gb = p.gb[c]; }
    if (g_mode == DW_IN_AFFINE2) gc = p.gc[c];
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const long total = (long)(nb > 0 ? nb : 0) * p.Tout;
    for (int j = 0; j < K; ++j) {
        float acc = 0.f;
        for (long idx = threadIdx.x; idx < total; idx += 256) {
            const int b = b0 + (int)(idx / p.Tout), t = (int)(idx % p.Tout);
            const int ti = t * S - p.pad + j;
            if (ti < 0 || ti >= p.Tin) continue;
            const size_t oo = ((size_t)b * p.C + c) * p.Tout + t;
            const float gvv = dw_in_transform(g_mode, p.g[oo], g_mode == DW_IN_AFFINE2 ? p.g2[oo] : 0.f, ga, gb, gc);
            const float xv = dw_in_transform(x_mode, p.x[((size_t)b * p.C + c) * p.Tin + ti], 0.f, xa, xb, 0.f);
            acc = fmaf(gvv, xv, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) red[wave * K + j] = acc;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += 256)
        p.partial[((size_t)g * p.C + c) * K + j] = (red[j] + red[K + j]) + (red[2 * K + j] + red[3 * K + j]);
}

// sum the [G][n] slabs into out[n]
__global__ void slab_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int G, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += partial[(size_t)g * n + i];
    out[i] = accumulate ? out[i] + s : s;
}

// ---------------------------------------------------------------------------------------------
template <int K, int S>
static void launch_dw(const DwParams& p, hipStream_t st) {
    dim3 grid(p.C, p.G);
    if (p.Tout > 256) hipLaunchKernelGGL((dwconv_kernel<K, S, 8>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((dwconv_kernel<K, S, 4>), grid, dim3(256), 0, st, p);
}
template <int K, int S>
static void launch_dw_wgrad(const DwWgradParams& p, hipStream_t st) {
    dim3 grid(p.C, p.G);
    // R = 8 only: the R = 4 instantiations of K >= 67 fall out of registers (hipcc 7.2 leaves the
    // accumulator array in scratch), and short rows are not the case this kernel is tuned for.
    hipLaunchKernelGGL((dwconv_wgrad_kernel<K, S, 8>), grid, dim3(256), 0, st, p);
}

// kernel sizes used by the reference's networks: asr.py:68-76, tts.py:18-25, 73-76
#define V100_DW_SPECIALISED(X) X(5) X(7) X(11) X(17) X(19) X(27) X(29) X(33) X(35) X(51) X(59) X(65) X(67) X(75) X(83)

extern "C" int v100_dw_num_groups(int B, int C) {
    // enough workgroups to fill 256 CUs several times over, but as few slabs as possible
    int G = ceil_div(2048, C > 0 ? C : 1);
    if (G > B) G = B;
    if (G < 1) G = 1;
    return G;
}

extern "C" int v100_dwconv(const float* x, const float* x2, const float* w, const float* in_a, const float* in_b,
                           const float* in_c, int in_mode, float* y, const float* aux, const float* out_a,
                           const float* out_b, int out_mode, float* stats, int G, int B, int C, int Tin, int Tout,
                           int K, int stride, int pad, int flip, int upsample, int force_generic, void* stream) {
    if (!x || !w || !y) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || Tin <= 0 || Tout <= 0 || K <= 0 || stride <= 0 || upsample <= 0 || G <= 0 || G > B)
        return V100_ERR_SHAPE;
    if (in_mode < 0 || in_mode > 2 || out_mode < 0 || out_mode > 3) return V100_ERR_SHAPE;
    if (in_mode != DW_IN_NONE && (!in_a || !in_b)) return V100_ERR_NULL;
    if (in_mode == DW_IN_AFFINE2 && (!x2 || !in_c)) return V100_ERR_NULL;
    if ((out_mode == DW_OUT_AFFINE_RELU6 || out_mode == DW_OUT_MASK_STATS) && (!out_a || !out_b)) return V100_ERR_NULL;
    if (out_mode == DW_OUT_MASK_STATS && !aux) return V100_ERR_NULL;
    if ((out_mode == DW_OUT_RAW_STATS || out_mode == DW_OUT_MASK_STATS) && !stats) return V100_ERR_NULL;
    DwParams p{x, x2, w, in_a, in_b, in_c, y, aux, out_a, out_b, stats,
               B, C, Tin, Tout, K, stride, pad, flip, upsample, G, in_mode, out_mode};
    hipStream_t st = (hipStream_t)stream;
    bool done = false;
    if (!force_generic && upsample == 1) {
#define X(KK) if (!done && K == KK && stride == 1) { launch_dw<KK, 1>(p, st); done = true; }
        V100_DW_SPECIALISED(X)
#undef X
        if (!done && K == 11 && stride == 2) { launch_dw<11, 2>(p, st); done = true; }
    }
    if (!done) hipLaunchKernelGGL(dwconv_generic_kernel, dim3(C, G), dim3(256), K * sizeof(float), st, p);
    return v100_launch_status();
}

extern "C" int v100_dwconv_wgrad(const float* g, const float* g2, const float* ga, const float* gb, const float* gc,
                                 int g_mode, const float* x, const float* xa, const float* xb, int x_mode,
                                 float* partial, float* dw, int G, int B, int C, int Tin, int Tout, int K, int stride,
                                 int pad, int force_generic, void* stream) {
    if (!g || !x || !partial || !dw) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || Tin <= 0 || Tout <= 0 || K <= 0 || stride <= 0 || G <= 0 || G > B) return V100_ERR_SHAPE;
    if (g_mode < 0 || g_mode > 2 || x_mode < 0 || x_mode > 1) return V100_ERR_SHAPE;
    if (g_mode != DW_IN_NONE && (!ga || !gb)) return V100_ERR_NULL;
    if (g_mode == DW_IN_AFFINE2 && (!g2 || !gc)) return V100_ERR_NULL;
    if (x_mode != DW_IN_NONE && (!xa || !xb)) return V100_ERR_NULL;
    DwWgradParams p{g, g2, ga, gb, gc, x, xa, xb, partial, B, C, Tin, Tout, K, stride, pad, G, g_mode, x_mode};
    hipStream_t st = (hipStream_t)stream;
    bool done = false;
    if (!force_generic) {
#define X(KK) if (!done && K == KK && stride == 1) { launch_dw_wgrad<KK, 1>(p, st); done = true; }
        V100_DW_SPECIALISED(X)
#undef X
        if (!done && K == 11 && stride == 2) { launch_dw_wgrad<11, 2>(p, st); done = true; }
    }
    if (!done) hipLaunchKernelGGL(dwconv_wgrad_generic_kernel, dim3(C, G), dim3(256), 4 * K * sizeof(float), st, p);
    const int n = C * K;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, partial, dw, G, n, 0);
    return v100_launch_status();
}
