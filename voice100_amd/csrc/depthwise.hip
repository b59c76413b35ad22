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
#include "depthwise_common.h"
#include "timing.h"

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
        // taps that land on a real (non-inserted) input sample: j = j0, j0+U, ... with u = t*S - pad + j a multiple of U;
        // consecutive ones read consecutive inputs, so no division or modulo inside the loop
        const int base = t * S - p.pad;
        int j0 = ((-base) % U + U) % U;                       // smallest j >= 0 with (base + j) % U == 0
        if (base + j0 < 0) j0 += ((-(base + j0)) + U - 1) / U * U;
        int ti = (base + j0) / U;
        for (int j = j0; j < K && ti < p.Tin; j += U, ++ti) {
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
// Backward-data of the stride-2 layer (L0 of the encoder, k = 11: asr.py:68): a stride-1 convolution with flipped
// taps over the upstream gradient zero-stuffed by 2.  Output t only meets the taps j of one parity (those with
// t - pad + j even), so a lane's 8 consecutive outputs need 12 consecutive upstream samples: three ds_read_b128 of
// the per-wave staged row (BN-backward affine of (dz2, a2) applied on the way in) and ~5.5 FMAs per output, all with
// compile-time tap / window indices.  Same epilogue as dwconv_kernel's DW_OUT_MASK_STATS (ReLU6 mask from a1,
// BN1-backward partial sums).  Memory-bound; replaces the one-thread-per-output generic kernel on the training path.
template <int K>
__global__ __launch_bounds__(256) void dwconv_up2_bwd_kernel(DwParams p) {
    constexpr int R = 8, TILE = 64 * R;              // outputs per wave pass
    constexpr int PADB = (K - 1) / 2;                // pad of the equivalent stride-1 conv (odd K, symmetric padding)
    constexpr int WOFF = 4;                          // the lane window starts WOFF samples before t0/2 (16-byte aligned)
    constexpr int SPAN = TILE / 2 + 12;              // staged samples per tile: [tile*256 - 4, tile*256 + 256 + 8)
    __shared__ __attribute__((aligned(16))) float lds_all[4][SPAN + 4];
    __shared__ float lds_w[K];
    __shared__ float lds_red[4][2];
    const int c = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* lds = lds_all[wave];
    for (int j = threadIdx.x; j < K; j += 256) lds_w[j] = p.w[(size_t)c * K + (K - 1 - j)];     // flipped taps
    __syncthreads();
    float wf[K];
#pragma unroll
    for (int j = 0; j < K; ++j) wf[j] = lds_w[j];
    const float ca = p.in_a[c], cb = p.in_b[c], cc = p.in_c[c], oa = p.out_a[c], ob = p.out_b[c];
    const int Tin = p.Tin, Tout = p.Tout;            // Tin: upstream length (forward output), Tout: forward input length
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int ntiles = (Tout + TILE - 1) / TILE;
    float s0 = 0.f, s1 = 0.f;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int in0 = tile * (TILE / 2) - WOFF;    // upstream index of LDS slot 0
        for (int bi = wave; bi < nb; bi += 4) {
            const int b = b0 + bi;
            const size_t ro = ((size_t)b * p.C + c) * Tin;
            // stage: SPAN samples, 4 per lane and pass (in0 is a multiple of 4: aligned float4 when Tin % 4 == 0)
#pragma unroll
            for (int v = 0; v < (SPAN + 255) / 256; ++v) {
                const int i = 4 * (lane + 64 * v);
                if (i < SPAN) {
                    const int ti0 = in0 + i;
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
                    if (ti0 >= 0 && ti0 + 3 < Tin) {               // whole quad in range: 16-byte (dword-aligned) loads
                        a = *reinterpret_cast<const f32x4u*>(p.x + ro + ti0);
                        a2 = *reinterpret_cast<const f32x4u*>(p.x2 + ro + ti0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ti0 + e >= 0 && ti0 + e < Tin) { a[e] = p.x[ro + ti0 + e]; a2[e] = p.x2[ro + ti0 + e]; }
                    }
                    f32x4 val;
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = (ti0 + e >= 0 && ti0 + e < Tin) ? fmaf(a[e], ca, fmaf(a2[e], cb, cc)) : 0.f;
                    *reinterpret_cast<f32x4*>(lds + i) = val;
                }
            }
            // other lanes' LDS stores must precede this lane's window reads: the hardware runs a wave's LDS operations
            // in order, the compiler (single-thread view: no alias between the two) must not swap them
            asm volatile("" ::: "memory");
            const int t0 = tile * TILE + lane * R;
            const size_t oo = ((size_t)b * p.C + c) * Tout + t0;
            float auxv[R];
            dw_load_run<R, true>(auxv, p.aux + oo, t0, Tout);
            const float* win = lds + lane * (R / 2);                 // = slot of upstream index t0/2 - WOFF
            f32x4 wv[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) wv[q] = *reinterpret_cast<const f32x4*>(win + 4 * q);
            float outv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    if (((r - PADB + j) & 1) == 0) {                  // t0 is even: parity of t - pad + j = parity of r - pad + j
                        const int wi = (r - PADB + j) / 2 + WOFF;      // (negative numerators are even here: exact)
                        if (wi >= 0 && wi < 12) acc = fmaf(wf[j], wv[wi >> 2][wi & 3], acc);
                    }
                }
                const float pre = fmaf(auxv[r], oa, ob);
                acc = (pre > 0.f && pre < 6.f) ? acc : 0.f;
                if (t0 + r < Tout) { s0 += acc; s1 = fmaf(acc, auxv[r], s1); }
                outv[r] = acc;
            }
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
            asm volatile("" ::: "memory");     // ... and the next row's stores must stay behind this row's reads
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) { lds_red[wave][0] = s0; lds_red[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        p.stats[((size_t)g * p.C + c) * 2 + 0] = (lds_red[0][0] + lds_red[1][0]) + (lds_red[2][0] + lds_red[3][0]);
        p.stats[((size_t)g * p.C + c) * 2 + 1] = (lds_red[0][1] + lds_red[1][1]) + (lds_red[2][1] + lds_red[3][1]);
    }
}

// ---------------------------------------------------------------------------------------------
// Backward-weight: dW[c][j] = sum_{b,t} g[b,c,t] * xin[b,c,t*S - pad + j].
// Same staging of xin (with its BN affine + ReLU6 recomputed on the way in); each lane keeps K
// partial sums in registers across all its tiles and the wave reduces them once at the end.
// Specialised for the training combination: g = BN-backward affine of (dz2, a2), xin = relu6(bn(a1)).
template <int K, int S, int R, bool AL>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwWgradParams p) {
    using G_ = DwGeom<K, S, R>;
    constexpr int TILE = G_::TILE, WIN = G_::WIN, SPAN = G_::SPAN, NV = G_::NV, NCH = G_::NCH, PD = G_::PD;

    __shared__ __attribute__((aligned(16))) float lds_all[4][G_::SPAN4 + 8];
    __shared__ float lds_red[4][K];

    const int c = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* lds = lds_all[wave];
    const float xa = p.xa[c], xb = p.xb[c], ga = p.ga[c], gb = p.gb[c], gc = p.gc[c];

    const int Tin = p.Tin, Tout = p.Tout;
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int ntiles = (Tout + TILE - 1) / TILE;

    float accw[K];
#pragma unroll
    for (int j = 0; j < K; ++j) accw[j] = 0.f;

    DwRaw<NV, false> raw;
    const __amdgpu_buffer_rsrc_t rx = dw_make_rsrc(p.x, (unsigned)((size_t)p.B * p.C * Tin * 4));
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int tile = 0; tile < ntiles; ++tile) {
    const int out0 = tile * TILE;
    const int in0 = out0 * S - p.pad;
    int bi = wave_u;
    if (bi < nb) {
        const unsigned rb = (unsigned)(((size_t)(b0 + bi) * p.C + c) * Tin * 4);
        dw_issue_loads<NV, SPAN, false, AL>(raw, rx, rx, rb, in0, Tin, lane);
    }
    for (; bi < nb; bi += 4) {
        const int b = b0 + bi;
        dw_stage_to_lds<NV, SPAN, DW_IN_AFFINE_RELU6, false>(raw, lds, in0, Tin, xa, xb, 0.f, lane);
        asm volatile("" ::: "memory");     // cross-lane LDS hand-off: keep stores before the window reads (see dwconv_kernel)
        if (bi + 4 < nb) {
            const unsigned rb = (unsigned)(((size_t)(b + 4) * p.C + c) * Tin * 4);
            dw_issue_loads<NV, SPAN, false, AL>(raw, rx, rx, rb, in0, Tin, lane);
        }
        const int t0 = out0 + lane * R;
        const size_t oo = ((size_t)b * p.C + c) * Tout + t0;
        float gv[R], gv2[R];
        dw_load_run<R, AL>(gv, p.g + oo, t0, Tout);
        dw_load_run<R, AL>(gv2, p.g2 + oo, t0, Tout);
#pragma unroll
        for (int r = 0; r < R; ++r)
            gv[r] = (t0 + r < Tout) ? fmaf(gv[r], ga, fmaf(gv2[r], gb, gc)) : 0.f;

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
        asm volatile("" ::: "memory");
    }
    }   // tile
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const float s = wave_sum_dpp_hi(accw[j]);
        if (lane == 63) lds_red[wave][j] = s;
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
static void launch_dw_wgrad(const DwWgradParams& p, hipStream_t st) {
    dim3 grid(p.C, p.G);
    // R = 8 only: the R = 4 instantiations of K >= 67 fall out of registers (hipcc 7.2 leaves the
    // accumulator array in scratch), and short rows are not the case this kernel is tuned for.
    V100_GGL((dwconv_wgrad_kernel<K, S, 8, true>), grid, dim3(256), 0, st, p);
}

extern "C" int v100_dw_num_groups(int B, int C) {
    // enough workgroups to fill 256 CUs several times over, but as few slabs as possible
    int G = ceil_div(2048, C > 0 ? C : 1);
    // 1024 channels are already 1024 workgroups (one round at 4 per CU): ONE group, so a workgroup owns its channel's complete
    // sums -- no slab to reduce for the weight gradient, BatchNorm finalised by the producing kernel (DwFin)
#ifndef DW_G1_MINC
#define DW_G1_MINC 1024
#endif
    if (C >= DW_G1_MINC) G = 1;
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
               B, C, Tin, Tout, K, stride, pad, flip, upsample, G, in_mode, out_mode, nullptr};
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes of this launch: fp32 input(s) + output (+ aux) + taps + per-channel coefficients
    const double nin = (in_mode == DW_IN_AFFINE2 ? 2.0 : 1.0) * B * C * (double)Tin, nout = (out_mode == DW_OUT_MASK_STATS ? 2.0 : 1.0) * B * C * (double)Tout;
    V100TimedLaunch timed(out_mode == DW_OUT_MASK_STATS ? V100_T_DW_BWD_DATA : V100_T_DW_FWD, 4.0 * (nin + nout) + 4.0 * C * K + 8.0 * C);
    bool done = false;
    const bool fits = (size_t)B * C * Tin * 4 < 0x7fffff00ull;
    if (!force_generic && upsample == 1 && fits) {
        if (in_mode == DW_IN_AFFINE_RELU6 && out_mode == DW_OUT_RAW_STATS) done = dw_launch_fwd_train(p, st, timed);
        else if (in_mode == DW_IN_NONE && out_mode == DW_OUT_AFFINE_RELU6) done = dw_launch_fwd_eval(p, st, timed);
        else if (in_mode == DW_IN_AFFINE2 && out_mode == DW_OUT_MASK_STATS) done = dw_launch_bwd_data(p, st, timed);
    }
    // backward-data of the stride-2 first layer: zero-stuffed by 2, flipped taps, symmetric padding
    if (!done && !force_generic && upsample == 2 && stride == 1 && flip && K == 11 && pad == (K - 1) / 2 && in_mode == DW_IN_AFFINE2 &&
        out_mode == DW_OUT_MASK_STATS) {
        V100_LAUNCH(timed, (dwconv_up2_bwd_kernel<11>), dim3(C, G), dim3(256), 0, st, p);
        done = true;
    }
    if (!done) V100_LAUNCH(timed, dwconv_generic_kernel, dim3(C, G), dim3(256), K * sizeof(float), st, p);
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
    V100TimedRegion timed(V100_T_DW_WGRAD, st, 4.0 * B * C * ((g_mode == DW_IN_AFFINE2 ? 2.0 : 1.0) * Tout + (double)Tin) + 4.0 * C * K);
    bool done = false;
    if (!force_generic && g_mode == DW_IN_AFFINE2 && x_mode == DW_IN_AFFINE_RELU6 && (size_t)B * C * Tin * 4 < 0x7fffff00ull) {
#define X(KK) if (!done && K == KK && stride == 1) { launch_dw_wgrad<KK, 1>(p, st); done = true; }
        V100_DW_SPECIALISED(X)
#undef X
        if (!done && K == 11 && stride == 2) { launch_dw_wgrad<11, 2>(p, st); done = true; }
    }
    if (!done) V100_GGL(dwconv_wgrad_generic_kernel, dim3(C, G), dim3(256), 4 * K * sizeof(float), st, p);
    const int n = C * K;
    V100_GGL(slab_reduce_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, partial, dw, G, n, 0);
    return v100_launch_status();
}

// 1 when the Toeplitz-MFMA kernels (and with them the 16-bit storage variants) exist for this (K, stride)
extern "C" int v100_dw_mfma_supported(int K, int stride) {
    if (stride != 1) return 0;
#define X(KK) if (K == KK) return 1;
    V100_DW_SPECIALISED(X)
#undef X
    return 0;
}

// "act16" entry points: stride-1 specialised K only, tensors selected by io16 (DW_IO_*) hold bf16 with pitched rows.
// No fallback: V100_ERR_SHAPE when there is no kernel for the shape (the executor then keeps fp32 storage).
extern "C" int v100_dwconv_fwd_train_io(const void* a1, const float* w, const float* in_a, const float* in_b, void* a2, float* stats,
                                        int G, int B, int C, int T, int K, int io16, void* stream) {
    return dw_fwd_train_io_fin(a1, w, in_a, in_b, a2, stats, G, B, C, T, K, io16, DwFin{}, DwPre{}, stream);
}

int dw_fwd_train_io_fin(const void* a1, const float* w, const float* in_a, const float* in_b, void* a2, float* stats, int G, int B, int C,
                        int T, int K, int io16, const DwFin& fin, const DwPre& pre, void* stream) {
    if (!a1 || !w || !in_a || !in_b || !a2 || !stats) return V100_ERR_NULL;
    if ((fin.mode != 0 || pre.f.mode != 0) && G != 1) return V100_ERR_SHAPE;
    const int cm = (io16 >> 4) & 1;              // DW_IO_CM: channel-major tensors [C][B][P] (streaming kernels: rows of up to 768 outputs)
    io16 &= 15;
    if (B <= 0 || C <= 0 || T <= 0 || K <= 0 || (K & 1) == 0 || G <= 0 || G > B || io16 != (DW_IO_X | DW_IO_Y)) return V100_ERR_SHAPE;
    if ((size_t)B * C * dw_pitch16(T, B) * 4 >= 0x7fffff00ull) return V100_ERR_SHAPE;
    DwParams p{(const float*)a1, nullptr, w, in_a, in_b, nullptr, (float*)a2, nullptr, nullptr, nullptr, stats,
               B, C, T, T, K, 1, (K - 1) / 2, 0, 1, G, DW_IN_AFFINE_RELU6, DW_OUT_RAW_STATS, nullptr, io16, fin, pre};
    p.cm = cm;
    hipStream_t st = (hipStream_t)stream;
    // (algorithmic bytes: the two streams, taps, coefficients -- and the producer's slab of partial sums when BatchNorm 1 is finalised here)
    V100TimedLaunch timed(V100_T_DW_FWD, 2.0 * B * C * 2.0 * T + 4.0 * C * K + 8.0 * C + (pre.f.mode != 0 ? 8.0 * pre.parts * C : 0.0));
    if (!dw_launch_fwd_train16(p, st, timed)) return V100_ERR_SHAPE;
    return v100_launch_status();
}

// eval-mode depthwise stage on bf16-stored hidden tensors (block executor, inference at precision "bf16"): h2 = relu6(conv(h1) * out_a
// + out_b), rows pitched to a multiple of 8 samples
int dw_fwd_eval_io(const void* h1, const float* w, const float* out_a, const float* out_b, void* h2, int B, int C, int T, int K, void* stream,
                   int cm, int f16) {
    if (!h1 || !w || !out_a || !out_b || !h2) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || K <= 0 || (K & 1) == 0) return V100_ERR_SHAPE;
    if ((size_t)B * C * dw_pitch16(T, B) * 4 >= 0x7fffff00ull) return V100_ERR_SHAPE;
    const int G = v100_dw_num_groups(B, C);
    DwParams p{(const float*)h1, nullptr, w, nullptr, nullptr, nullptr, (float*)h2, nullptr, out_a, out_b, nullptr,
               B, C, T, T, K, 1, (K - 1) / 2, 0, 1, G, DW_IN_NONE, DW_OUT_AFFINE_RELU6, nullptr, DW_IO_X | DW_IO_Y, DwFin{}, DwPre{}};
    p.cm = cm;
    // rows much shorter than a wave item (1-second chunks: 51 outputs against 512 positions): several utterances side by side in one
    // item, each followed by >= pad zeros (its own right padding = the next one's left padding)
    const int P = dw_pitch16(T, B), pad = (K - 1) / 2;
    // a multiple of 16: an output then keeps its row (t mod 16) inside the 16 x 16 Toeplitz block whatever its segment, so the packed
    // form sums exactly what the one-row form sums
    const int ss = (P + pad + 15) & ~15;
    const int room = T <= 512 ? 512 : 768;
    int segn = T <= 512 ? (room - P) / ss + 1 : 1;
    if (segn > 1 && (segn - 1) * ss + P > room) --segn;
    if (segn > 1) { p.segn = segn; p.segs = ss; }
    hipStream_t st = (hipStream_t)stream;
    V100TimedLaunch timed(V100_T_DW_FWD, 2.0 * B * C * 2.0 * T + 4.0 * C * K + 8.0 * C);
    if (!dw_launch_fwd_eval16(p, st, timed, f16 != 0)) return V100_ERR_SHAPE;
    return v100_launch_status();
}

extern "C" int v100_dwconv_bwd_io(const void* g, const void* g2, const float* w, const float* ga, const float* gb, const float* gc,
                                  const void* xpre, const float* xa, const float* xb, void* dxin, float* stats, float* wpartial,
                                  float* dw, int G, int B, int C, int T, int K, int io16, void* stream) {
    return dw_bwd_io_fin(g, g2, w, ga, gb, gc, xpre, xa, xb, dxin, stats, wpartial, dw, G, B, C, T, K, io16, DwFin{}, DwPre{}, stream);
}

// The same pass finishing BatchNorm 1's backward itself (one group) and writing the FINISHED gradient da1 = p dz1 + q a1 + r (bf16)
// instead of dz1: what the block executor issues for the 16-bit training step (block.hip), exposed for the kernel tests / benchmarks.
extern "C" int v100_dwconv_bwd_da1_io(const void* g, const void* g2, const float* w, const float* ga, const float* gb, const float* gc,
                                      const void* xpre, const float* xa, const float* xb, void* da1_out, float* stats, float* dw,
                                      const float* bn1_gamma, const float* bn1_mean, const float* bn1_rstd, float* pqr, float* dgamma,
                                      float* dbeta, int B, int C, int T, int K, void* stream) {
    if (!bn1_gamma || !bn1_mean || !bn1_rstd || !pqr || !dgamma || !dbeta) return V100_ERR_NULL;
    DwFin fin{};
    fin.mode = 2;
    fin.count = (double)B * T;
    fin.gamma = bn1_gamma; fin.a = bn1_mean; fin.b = bn1_rstd;
    fin.o0 = pqr; fin.o1 = pqr + C; fin.o2 = pqr + 2 * (size_t)C; fin.o3 = dgamma; fin.o4 = dbeta;
    return dw_bwd_io_fin(g, g2, w, ga, gb, gc, xpre, xa, xb, da1_out, stats, dw, dw, 1, B, C, T, K,
                         DW_IO_X | DW_IO_X2 | DW_IO_AUX | DW_IO_Y, fin, DwPre{}, stream, 1);
}

int dw_bwd_io_fin(const void* g, const void* g2, const float* w, const float* ga, const float* gb, const float* gc, const void* xpre,
                  const float* xa, const float* xb, void* dxin, float* stats, float* wpartial, float* dw, int G, int B, int C, int T, int K,
                  int io16, const DwFin& fin, const DwPre& pre, void* stream, int da1) {
    if (!g || !g2 || !w || !ga || !gb || !gc || !xpre || !xa || !xb || !dxin || !stats || !wpartial || !dw) return V100_ERR_NULL;
    if ((fin.mode != 0 || pre.f.mode != 0) && G != 1) return V100_ERR_SHAPE;
    if (da1 && (fin.mode != 2 || !dw_bwd_da1_supported(B, C, T, K, G))) return V100_ERR_SHAPE;
    if (B <= 0 || C <= 0 || T <= 0 || K <= 0 || (K & 1) == 0 || G <= 0 || G > B) return V100_ERR_SHAPE;
    if ((size_t)B * C * dw_pitch16(T, B) * 4 >= 0x7fffff00ull) return V100_ERR_SHAPE;
    const int cm = (io16 >> 4) & 1;              // DW_IO_CM (see dw_fwd_train_io_fin)
    io16 &= 15;
    hipStream_t st = (hipStream_t)stream;
    const int pad = (K - 1) / 2;
    DwParams p{(const float*)g, (const float*)g2, w, ga, gb, gc, (float*)dxin, (const float*)xpre, xa, xb, stats,
               B, C, T, T, K, 1, K - 1 - pad, 1, 1, G, DW_IN_AFFINE2, DW_OUT_MASK_STATS, wpartial, io16, fin, pre};
    p.cm = cm;
    p.da1 = da1;
    if (cm && io16 != (DW_IO_X | DW_IO_X2 | DW_IO_AUX | DW_IO_Y)) return V100_ERR_SHAPE;
    const bool all16 = io16 == (DW_IO_X | DW_IO_X2 | DW_IO_AUX | DW_IO_Y);
    if (!all16 && io16 != (DW_IO_X2 | DW_IO_AUX)) return V100_ERR_SHAPE;
    if (da1 && (!all16 || cm)) return V100_ERR_SHAPE;
    V100TimedLaunch timed(V100_T_DW_BWD_DATA, (all16 ? 2.0 : 3.0) * B * C * 4.0 * T + 8.0 * C * K + 8.0 * C + (pre.f.mode != 0 ? 8.0 * pre.parts * C : 0.0));
    if (G == 1) p.wpartial = dw;               // one group: the kernel's "partial" IS the weight gradient
    const bool done = all16 ? dw_launch_bwd_fused16g(p, st, timed) : dw_launch_bwd_fused16(p, st, timed);
    if (!done) return V100_ERR_SHAPE;
    const int n = C * K;
    if (G > 1) V100_GGL(slab_reduce_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, wpartial, dw, G, n, 0);
    return v100_launch_status();
}

// Backward of the training-mode depthwise stage in one call: dxin (through the ReLU6 mask of a1, with the BN1-backward
// partial sums) and dW.  Stride 1 with a specialised K: ONE fused kernel (see dwconv_kernel, WG); anything else: the
// two stand-alone passes above.  Same results either way up to fp32 summation order.
extern "C" int v100_dwconv_bwd(const float* g, const float* g2, const float* w, const float* ga, const float* gb,
                               const float* gc, const float* xpre, const float* xa, const float* xb, float* dxin,
                               float* stats, float* wpartial, float* dw, int G, int B, int C, int Tin, int Tout, int K,
                               int stride, int pad, int force_split, void* stream) {
    if (!g || !g2 || !w || !ga || !gb || !gc || !xpre || !xa || !xb || !dxin || !stats || !wpartial || !dw) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || Tin <= 0 || Tout <= 0 || K <= 0 || stride <= 0 || G <= 0 || G > B) return V100_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const bool fits = (size_t)B * C * (Tin > Tout ? Tin : Tout) * 4 < 0x7fffff00ull;
    if (!force_split && stride == 1 && fits) {
        // backward-data geometry: the conv runs over g (length Tout) and produces Tin outputs, taps flipped
        DwParams p{g, g2, w, ga, gb, gc, dxin, xpre, xa, xb, stats,
                   B, C, Tout, Tin, K, 1, K - 1 - pad, 1, 1, G, DW_IN_AFFINE2, DW_OUT_MASK_STATS, wpartial};
        V100TimedLaunch timed(V100_T_DW_BWD_DATA, 4.0 * B * C * (2.0 * Tout + 2.0 * Tin) + 8.0 * C * K + 8.0 * C);
        if (G == 1) p.wpartial = dw;
        const bool done = dw_launch_bwd_fused(p, st, timed);
        if (done) {
            const int n = C * K;
            if (G > 1) V100_GGL(slab_reduce_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, wpartial, dw, G, n, 0);
            return v100_launch_status();
        }
        p.wpartial = wpartial;
    }
    int rc = v100_dwconv_wgrad(g, g2, ga, gb, gc, DW_IN_AFFINE2, xpre, xa, xb, DW_IN_AFFINE_RELU6, wpartial, dw, G, B, C, Tin, Tout, K,
                               stride, pad, 0, stream);
    if (rc != V100_OK) return rc;
    return v100_dwconv(g, g2, w, ga, gb, gc, DW_IN_AFFINE2, dxin, xpre, xa, xb, DW_OUT_MASK_STATS, stats, G, B, C, Tout, Tin, K, 1,
                       K - 1 - pad, 1, stride, 0, stream);
}
