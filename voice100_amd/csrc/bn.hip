// K3 + glue: BatchNorm1d statistics finalisation (training), folded coefficients (eval), the
// BatchNorm-backward reductions, and the small per-channel elementwise / layout kernels the
// inverted-residual block needs between its GEMM and depthwise kernels.
//
// Reference semantics: nn.BatchNorm1d(eps=1e-5, momentum=0.1, affine, track_running_stats) as
// instantiated at voice100/models/asr.py:36,52; batch statistics over (B, T); running_var takes
// the unbiased batch variance; num_batches_tracked += 1 per training forward.
//
// The producers (pointwise.hip / depthwise.hip epilogues) leave per-channel partial sums in a
// [parts][C][2] slab; the finalisers below reduce the slab in double (fixed order, deterministic)
// and emit  scale = gamma * rstd,  shift = beta - mean * scale  for the consumer's prologue.
#include "common.h"
#include "depthwise_common.h"     // DwFin / dw_finalize: BatchNorm finalisation inside a producing kernel
#ifndef BN_XCD
#define BN_XCD 1      /* (step 3.218 -> 3.208 ms, A/B on one box) the one-workgroup-per-channel boundary passes: blockIdx -> channel in groups of 16 per XCD (v100_chan_of_block) */
#endif
#ifndef BN_FIN_WPB
#define BN_FIN_WPB 4              /* channels (= waves) per workgroup of the stand-alone BatchNorm finalisers */
#endif

__global__ void bn_finalize_train_kernel(const float* __restrict__ stats, int parts, double count,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                         long long* __restrict__ num_batches_tracked, float momentum, float eps,
                                         float* __restrict__ scale, float* __restrict__ shift,
                                         float* __restrict__ save_mean, float* __restrict__ save_rstd, int C) {
    // one wave per channel (BN_FIN_WPB channels per workgroup): lanes stride over the slab rows, fixed-order butterfly in double
    const int c = blockIdx.x * BN_FIN_WPB + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    if (c == 0 && lane == 0 && num_batches_tracked) *num_batches_tracked += 1;
    double s0 = 0.0, s1 = 0.0;
    for (int g = lane; g < parts; g += 64) {
        s0 += (double)stats[((size_t)g * C + c) * 2 + 0];
        s1 += (double)stats[((size_t)g * C + c) * 2 + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if (lane != 0) return;
    const double mean = s0 / count;
    double var = s1 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_rstd) save_rstd[c] = rstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(running_var[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - running_mean[c] * sc;
}

// FROZEN statistics under autograd (a block in eval() inside a model that trains: the reference's nn.BatchNorm1d then normalises with the
// running statistics and back-propagates through that fixed affine): the forward coefficients plus the (mean, rstd) pair the backward
// finaliser reads, all from the running statistics; nothing is updated.
__global__ void bn_frozen_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                        float eps, float* __restrict__ scale, float* __restrict__ shift,
                                        float* __restrict__ mean, float* __restrict__ rstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rs = 1.0f / sqrtf(running_var[c] + eps);
    const float sc = gamma[c] / sqrtf(running_var[c] + eps);          // exactly bn_eval_coeffs_kernel's value: eval and frozen-train forwards agree
    scale[c] = sc;
    shift[c] = beta[c] - running_mean[c] * sc;
    mean[c] = running_mean[c];
    rstd[c] = rs;
}

// Backward of y = gamma * (a - mean) * rstd + beta given the slab of (sum dz, sum dz*a):
//   dgamma = rstd * (sum dz*a - mean * sum dz),  dbeta = sum dz
//   da = p*dz + q*a + r   with  p = gamma*rstd,  q = -gamma*rstd^2 * dgamma/n,
//                               r = -gamma*rstd*dbeta/n - q*mean
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int parts, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, float* __restrict__ p, float* __restrict__ q,
                                       float* __restrict__ r, float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int frozen) {
    const int c = blockIdx.x * BN_FIN_WPB + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int g = lane; g < parts; g += 64) {
        s0 += (double)partial[((size_t)g * C + c) * 2 + 0];
        s1 += (double)partial[((size_t)g * C + c) * 2 + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if (lane != 0) return;
    const double mu = mean[c], rs = rstd[c], ga = gamma[c];
    const double dg = rs * (s1 - mu * s0);
    const double pp = ga * rs;
    // frozen statistics: mean and rstd are constants, only the direct term is left (da = gamma * rstd * dz); dgamma / dbeta as above
    const double qq = frozen ? 0.0 : -ga * rs * rs * dg / count;
    const double rr = frozen ? 0.0 : -pp * s0 / count - qq * mu;
    p[c] = (float)pp;
    q[c] = (float)qq;
    r[c] = (float)rr;
    if (dgamma) dgamma[c] = (float)dg;
    if (dbeta) dbeta[c] = (float)s0;
}

// partial[g][c] = (sum u, sum u*v) over this group's batches and all t; v may be null (then sum u*u)
__global__ __launch_bounds__(256) void chan_reduce2_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                           float* __restrict__ partial, int B, int C, int T, int G) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, g = blockIdx.y;
    const int bper = (B + G - 1) / G;
    const int b0 = g * bper, b1 = min(B, b0 + bper);
    float s0 = 0.f, s1 = 0.f;
    const bool vec = (T & 3) == 0;
    for (int b = b0; b < b1; ++b) {
        const size_t ro = ((size_t)b * C + c) * T;
        if (vec) {
            for (int t = threadIdx.x * 4; t < T; t += 1024) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(u + ro + t);
                const f32x4 w = v ? *reinterpret_cast<const f32x4*>(v + ro + t) : a;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0 += a[e]; s1 = fmaf(a[e], w[e], s1); }
            }
        } else {
            for (int t = threadIdx.x; t < T; t += 256) {
                const float a = u[ro + t];
                const float w = v ? v[ro + t] : a;
                s0 += a; s1 = fmaf(a, w, s1);
            }
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((size_t)g * C + c) * 2 + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        partial[((size_t)g * C + c) * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

// ---- the two block-boundary passes over tensors of which some are STORED as bf16 [B][C][P], P = v100_pitch16(T, B) ("act16", see
// include/voice100_hip.h): a3 (project output, saved for backward) and da3 (its BatchNorm-backward gradient, executor-internal).
// Row-wise addressing: fp32 operands have pitch T, bf16 operands pitch P; 4 samples per thread and step.
typedef unsigned int bn_u32x2 __attribute__((ext_vector_type(2)));
template <bool B16>
__device__ __forceinline__ void chan_load4(const void* base, size_t row, int T, int P, int t, float (&o)[4]) {
    if constexpr (B16) {
        const bn_u32x2 w = *reinterpret_cast<const bn_u32x2*>(reinterpret_cast<const u16*>(base) + row * P + t);     // t % 4 == 0, P % 8 == 0
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned d = w[e >> 1];
            o[e] = (t + e < T) ? __builtin_bit_cast(float, (e & 1) ? (d & 0xffff0000u) : (d << 16)) : 0.f;
        }
    } else {
        const float* p = reinterpret_cast<const float*>(base) + row * T + t;
        if (t + 3 < T) {
            const f32x4 v = *reinterpret_cast<const f32x4u*>(p);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (t + e < T) ? p[e] : 0.f;
        }
    }
}

// io: 1 u is bf16, 2 v is bf16
template <int IO>
__global__ __launch_bounds__(256) void chan_reduce2_io_kernel(const void* __restrict__ u, const void* __restrict__ v, float* __restrict__ partial,
                                                              int B, int C, int T, int G) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, g = blockIdx.y;
    const int bper = (B + G - 1) / G;
    const int b0 = g * bper, b1 = min(B, b0 + bper);
    const int P = v100_pitch16(T, B);
    float s0 = 0.f, s1 = 0.f;
    for (int b = b0; b < b1; ++b) {
        const size_t row = (size_t)b * C + c;
        for (int t = threadIdx.x * 4; t < T; t += 1024) {
            float a[4], w[4];
            chan_load4<(IO & 1) != 0>(u, row, T, P, t, a);
            chan_load4<(IO & 2) != 0>(v, row, T, P, t, w);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s0 += a[e]; s1 = fmaf(a[e], w[e], s1); }
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((size_t)g * C + c) * 2 + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        partial[((size_t)g * C + c) * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

// The same sums with ONE 1024-thread workgroup per channel, which therefore owns the channel's complete sums and runs the
// BatchNorm-backward finaliser itself (DwFin mode 2: p / q / r, dgamma, dbeta) -- no slab, no finaliser launch.
template <int IO>
__global__ __launch_bounds__(1024) void chan_reduce2_fin_kernel(const void* __restrict__ u, const void* __restrict__ v, float* __restrict__ partial,
                                                                int B, int C, int T, DwFin fin) {
    __shared__ float red[16][2];
    const int c = blockIdx.x;
    const int P = v100_pitch16(T, B);
    const int T4 = (T + 3) >> 2;
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < B * T4; i += 1024) {
        const int b = i / T4, t = (i - b * T4) * 4;
        const size_t row = (size_t)b * C + c;
        float a[4], w[4];
        chan_load4<(IO & 1) != 0>(u, row, T, P, t, a);
        chan_load4<(IO & 2) != 0>(v, row, T, P, t, w);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0 += a[e]; s1 = fmaf(a[e], w[e], s1); }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t0 += red[k][0]; t1 += red[k][1]; }       // fixed order
        partial[(size_t)c * 2 + 0] = t0;
        partial[(size_t)c * 2 + 1] = t1;
        dw_finalize(fin, c, t0, t1);
    }
}

// The whole BatchNorm-3 backward of a block in ONE pass over (dy, a3): the channel's B x T samples of both tensors stay in the
// registers of its 1024-thread workgroup between the reduction (sum dy, sum dy*a3 -> p, q, r, dgamma, dbeta) and the affine
// da3 = p*dy + q*a3 + r (bf16, pitched) that the two-kernel form re-read from memory.  NQ float4-quads per thread: B * ceil(T/4)
// <= 1024 * NQ (the caller checks).  u = dy fp32 -- or bf16 (pitched) when U16: the gradient stream between the blocks of a stack in its
// 16-bit form (round 6, block.hip) --, v = a3 bf16 (pitched), out = da3 bf16 (pitched).
template <int NQ, bool U16 = false>
__global__ __launch_bounds__(1024) void chan_bn3_bwd_kernel(const void* __restrict__ u, const void* __restrict__ v, float* __restrict__ partial,
                                                            void* __restrict__ out, int B, int C, int T, DwFin fin) {
    __shared__ float red[16][2];
    __shared__ float coef[3];
    const int c = BN_XCD ? v100_chan_of_block<16>(blockIdx.x, C) : (int)blockIdx.x;
    const int P = v100_pitch16(T, B);
    const int T4 = (T + 3) >> 2;
    const int n = B * T4;
    float a[NQ][4], w[NQ][4];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int i = threadIdx.x + 1024 * k;
        if (i < n) {
            const int b = i / T4, t = (i - b * T4) * 4;
            const size_t row = (size_t)b * C + c;
            chan_load4<U16>(u, row, T, P, t, a[k]);
            chan_load4<true>(v, row, T, P, t, w[k]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[k][e] = 0.f; w[k][e] = 0.f; }
        }
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0 += a[k][e]; s1 = fmaf(a[k][e], w[k][e], s1); }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t0 += red[k][0]; t1 += red[k][1]; }       // fixed order
        partial[(size_t)c * 2 + 0] = t0;
        partial[(size_t)c * 2 + 1] = t1;
        dw_finalize_d(fin, c, (double)t0, (double)t1, coef);
    }
    __syncthreads();
    const float pa = coef[0], qb = coef[1], rc = coef[2];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int i = threadIdx.x + 1024 * k;
        if (i < n) {
            const int b = i / T4, t = (i - b * T4) * 4;
            const size_t row = (size_t)b * C + c;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(a[k][e], pa, fmaf(w[k][e], qb, rc));       // as chan_affine2_io rounds it
            const bn_u32x2 w16 = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
            *reinterpret_cast<bn_u32x2*>(reinterpret_cast<u16*>(out) + row * P + t) = w16;      // samples past T land in the row's padding
        }
    }
}

// out = A[c]*u + Bc[c]*v + Cc[c]; io: 1 u is bf16, 2 v is bf16, 4 out is bf16; v may be null
template <int IO>
__global__ __launch_bounds__(256) void chan_affine2_io_kernel(const void* __restrict__ u, const void* __restrict__ v, const float* __restrict__ A,
                                                              const float* __restrict__ Bc, const float* __restrict__ Cc, void* __restrict__ out,
                                                              int C, int T, long rows) {
    const int P = v100_pitch16(T, (int)(rows / C));       // rows = B * C
    const int T4 = (T + 3) >> 2;
    const long total = rows * T4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const size_t row = (size_t)(i / T4);
        const int t = (int)(i % T4) * 4;
        const int c = (int)(row % C);
        const float a = A ? A[c] : 1.f, b = Bc ? Bc[c] : 1.f, cc = Cc ? Cc[c] : 0.f;
        float x[4], y[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        chan_load4<(IO & 1) != 0>(u, row, T, P, t, x);
        if (v) chan_load4<(IO & 2) != 0>(v, row, T, P, t, y);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v ? fmaf(x[e], a, fmaf(y[e], b, cc)) : fmaf(x[e], a, cc);
        if constexpr ((IO & 4) != 0) {
            const bn_u32x2 w = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
            *reinterpret_cast<bn_u32x2*>(reinterpret_cast<u16*>(out) + row * P + t) = w;       // samples past T land in the row's padding
        } else {
            float* q = reinterpret_cast<float*>(out) + row * T + t;
            if (t + 3 < T) { const f32x4 w = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4u*>(q) = w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (t + e < T) q[e] = o[e];
            }
        }
    }
}

// The forward block output with a bf16 SHADOW beside it: out = A[c]*u + Cc[c] (+ v) as fp32 [rows][T] and the same values rounded
// to bf16 in a pitched copy [rows][v100_pitch16(T, B)] -- what the next block's expand GEMM and expand weight gradient load as their X
// operand (they round X to bf16 anyway: identical results, half the bytes through the CU's 64 B/clk vector-memory path, which is
// what bounds the 256-row backward-weight kernel once its fp32 X tile is 64 of the 96 KB it stages per step).  UB: u is bf16 (pitched).
template <bool UB>
__global__ __launch_bounds__(256) void chan_affine2_shadow_kernel(const void* __restrict__ u, const float* __restrict__ v, const float* __restrict__ A,
                                                                  const float* __restrict__ Cc, float* __restrict__ out, u16* __restrict__ shadow,
                                                                  int C, int T, long rows) {
    const int P = v100_pitch16(T, (int)(rows / C));       // rows = B * C
    const int T4 = (T + 3) >> 2;
    const long total = rows * T4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const size_t row = (size_t)(i / T4);
        const int t = (int)(i % T4) * 4;
        const int c = (int)(row % C);
        const float a = A ? A[c] : 1.f, cc = Cc ? Cc[c] : 0.f;
        float x[4], y[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        chan_load4<UB>(u, row, T, P, t, x);
        if (v) chan_load4<false>(v, row, T, P, t, y);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v ? fmaf(x[e], a, fmaf(y[e], 1.f, cc)) : fmaf(x[e], a, cc);    // as chan_affine2(_io) rounds it
        float* q = out + row * T + t;
        if (t + 3 < T) { const f32x4 w = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4u*>(q) = w; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (t + e < T) q[e] = o[e];
        }
        const bn_u32x2 w16 = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
        *reinterpret_cast<bn_u32x2*>(shadow + row * P + t) = w16;          // samples past T land in the row's padding
    }
}

// The forward block output with BatchNorm 3 finalised by the kernel itself: one 1024-thread workgroup per channel, whose first wave
// turns the project GEMM's slab of partial sums into scale / shift (+ saved mean / rstd, running statistics: dw_finalize_parts), then
// out = scale*u + shift (+ v) for the channel's B x T samples, optionally with the bf16 shadow (SH).
// Activation storage level 5 (VB / NOF): the residual v is read from its bf16 shadow (the previous block's output as the next expand GEMM
// already reads it) and the fp32 copy of this block's output is not written at all (`out` == nullptr): interior blocks of a stack keep the
// residual stream in ONE 16-bit form, as the reference's autocast run does (half the bytes of this pass).
template <bool UB, bool SH, bool VB = false>
__global__ __launch_bounds__(1024) void chan_affine2_fin_kernel(const void* __restrict__ u, const void* __restrict__ v, float* __restrict__ out,
                                                                u16* __restrict__ shadow, int B, int C, int T, DwPre pre) {
    __shared__ float coef[3];
    const int c = BN_XCD ? v100_chan_of_block<16>(blockIdx.x, C) : (int)blockIdx.x;
    if (threadIdx.x < 64) dw_finalize_parts(pre, C, c, threadIdx.x, coef);
    __syncthreads();
    const float a = coef[0], cc = coef[1];
    const int P = v100_pitch16(T, B);
    const int T4 = (T + 3) >> 2;
    for (int i = threadIdx.x; i < B * T4; i += 1024) {
        const int b = i / T4, t = (i - b * T4) * 4;
        const size_t row = (size_t)b * C + c;
        float x[4], y[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        chan_load4<UB>(u, row, T, P, t, x);
        if (v) chan_load4<VB>(v, row, T, P, t, y);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v ? fmaf(x[e], a, fmaf(y[e], 1.f, cc)) : fmaf(x[e], a, cc);
        if (out) {
            float* q = out + row * T + t;
            if (t + 3 < T) { const f32x4 w = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4u*>(q) = w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (t + e < T) q[e] = o[e];
            }
        }
        if constexpr (SH) {
            const bn_u32x2 w16 = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
            *reinterpret_cast<bn_u32x2*>(shadow + row * P + t) = w16;
        }
    }
}

// out[c] = sum_g partial[g][c][0]   (bias gradients): one wave per channel, its lanes stride over the parts (a thread per channel
// walked `parts` dependent loads: 8 us for the vocabulary head's 29 channels)
__global__ __launch_bounds__(256) void slab_sum0_kernel(const float* __restrict__ partial, int parts, float* __restrict__ out, int C) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    for (int g = lane; g < parts; g += 64) s += (double)partial[((size_t)g * C + c) * 2];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[c] = (float)s;
}

// out = A[c]*u + Bc[c]*v + Cc[c]   (v / Bc / Cc optional: missing Bc = 1, missing v or Cc = 0)
//   forward block output:  y = scale3*a3 + shift3 (+ x)        (asr.py:55-59)
//   backward:              da3 = p*dy + q*a3 + r
__global__ __launch_bounds__(256) void chan_affine2_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                           const float* __restrict__ A, const float* __restrict__ Bc,
                                                           const float* __restrict__ Cc, float* __restrict__ out,
                                                           int C, int T, long total) {
    const bool vec = (T & 3) == 0;
    if (vec) {
        const long n4 = total >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            const long e0 = i << 2;
            const int c = (int)((e0 / T) % C);
            const float a = A ? A[c] : 1.f, b = Bc ? Bc[c] : 1.f, cc = Cc ? Cc[c] : 0.f;
            const f32x4 x = *reinterpret_cast<const f32x4*>(u + e0);
            f32x4 o;
            if (v) {
                const f32x4 y = *reinterpret_cast<const f32x4*>(v + e0);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(x[e], a, fmaf(y[e], b, cc));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(x[e], a, cc);
            }
            *reinterpret_cast<f32x4*>(out + e0) = o;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int c = (int)((i / T) % C);
            const float a = A ? A[c] : 1.f, b = Bc ? Bc[c] : 1.f, cc = Cc ? Cc[c] : 0.f;
            out[i] = fmaf(u[i], a, (v ? v[i] * b : 0.f) + cc);
        }
    }
}

// out = u * m * s   (inverted dropout with a pre-drawn keep mask; asr.py:90)
__global__ void mul_scale_kernel(const float* __restrict__ u, const float* __restrict__ m, float s, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = u[i] * m[i] * s;
}

// nn.Dropout(p) in ONE pass (asr.py:90): the keep decision of element i is bit-mixing of (seed, i) -- splitmix64, two 32-bit
// uniforms per call -- compared with p * 2^32; y = keep ? x / (1 - p) : 0 and a byte mask for the backward pass.  Replaces
// torch.rand + compare + cast + multiply (4 kernels, a float mask of the activation's size written and read twice).
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void dropout_fwd_kernel(const float* __restrict__ x, unsigned long long seed, unsigned thresh, float scale, float* __restrict__ y,
                                   unsigned char* __restrict__ mask, long n4, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long r0 = splitmix64(seed ^ (unsigned long long)(2 * i) * 0xD6E8FEB86659FD93ull);
        const unsigned long long r1 = splitmix64(seed ^ (unsigned long long)(2 * i + 1) * 0xD6E8FEB86659FD93ull);
        const unsigned u[4] = {(unsigned)r0, (unsigned)(r0 >> 32), (unsigned)r1, (unsigned)(r1 >> 32)};
        const long e0 = i * 4;
        if (e0 + 3 < n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + e0);
            f32x4 o;
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool keep = u[e] >= thresh;
                o[e] = keep ? v[e] * scale : 0.f;
                m |= (keep ? 1u : 0u) << (8 * e);
            }
            *reinterpret_cast<f32x4*>(y + e0) = o;
            *reinterpret_cast<unsigned*>(mask + e0) = m;
        } else {
            for (int e = 0; e < 4 && e0 + e < n; ++e) {
                const bool keep = u[e] >= thresh;
                y[e0 + e] = keep ? x[e0 + e] * scale : 0.f;
                mask[e0 + e] = keep ? 1 : 0;
            }
        }
    }
}
__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ mask, float scale, float* __restrict__ dx,
                                   long n4, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e0 = i * 4;
        if (e0 + 3 < n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dy + e0);
            const unsigned m = *reinterpret_cast<const unsigned*>(mask + e0);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((m >> (8 * e)) & 1u) ? v[e] * scale : 0.f;
            *reinterpret_cast<f32x4*>(dx + e0) = o;
        } else {
            for (int e = 0; e < 4 && e0 + e < n; ++e) dx[e0 + e] = mask[e0 + e] ? dy[e0 + e] * scale : 0.f;
        }
    }
}

// [B][R][Cc] -> [B][Cc][R]   (model-edge layouts: asr.py:111,114; tts.py:177,179)
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const float* src = in + (size_t)b * R * Cc;
    float* dst = out + (size_t)b * R * Cc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < R && c < Cc) tile[ty + 8 * i][tx] = src[(size_t)r * Cc + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < Cc) dst[(size_t)c * R + r] = tile[tx][ty + 8 * i];
    }
}

// Channel-major inference layout (block.hip, v100_ir_fwd_eval with shape[10] == 2): [B][C][T] -> [C][B][P], P = v100_pitch16(T, B), the padding
// columns zeroed; and back out of it at the model's edge: [C][B][P] -> [B][T][C]  (asr.py:114: transpose(1, 2) of the logits)
__global__ __launch_bounds__(256) void bct_to_cm_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int T, int P) {
    // flat over the output (rows of 51 samples would leave a workgroup per row 80 % idle: 30 us for 256 x 256 rows)
    const long long total = (long long)C * B * P;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / P;
        const int t = (int)(i - row * P);
        const int c = (int)(row / B), b = (int)(row - (long long)c * B);
        out[i] = t < T ? in[((size_t)b * C + c) * T + t] : 0.f;
    }
}
__global__ __launch_bounds__(256) void cm_to_btc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int T, int P) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, t = t0 + tx;
        if (c < C && t < T) tile[ty + 8 * i][tx] = in[((size_t)c * B + b) * P + t];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, c = c0 + tx;
        if (c < C && t < T) out[((size_t)b * T + t) * C + c] = tile[tx][ty + 8 * i];
    }
}

// K6: out[b][c][t] = table[idx[b][t]][c]   (nn.Embedding + transpose(1,2): tts.py:81-83, 176-177)
__global__ __launch_bounds__(256) void embedding_bct_kernel(const long long* __restrict__ idx, const float* __restrict__ table,
                                                            float* __restrict__ out, int V, int C, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, c = c0 + tx;
        if (t < T && c < C) {
            long long id = idx[(size_t)b * T + t];
            if (id < 0) id = 0;
            if (id >= V) id = V - 1;
            tile[ty + 8 * i][tx] = table[(size_t)id * C + c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, t = t0 + tx;
        if (t < T && c < C) out[((size_t)b * C + c) * T + t] = tile[tx][ty + 8 * i];
    }
}

// d_table[idx[b][t]][c] += g[b][c][t]  (embedding backward; vocab is tiny so fp32 atomics per row)
__global__ void embedding_bwd_kernel(const long long* __restrict__ idx, const float* __restrict__ g, float* __restrict__ dtable,
                                     int V, int C, int T, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const int c = (int)((i / T) % C);
        const int b = (int)(i / ((long)T * C));
        long long id = idx[(size_t)b * T + t];
        if (id >= 0 && id < V) atomicAdd(&dtable[(size_t)id * C + c], g[i]);
    }
}

// ---------------------------------------------------------------------------------------------
extern "C" int v100_bn_finalize_train(const float* stats, int parts, long long count, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                                      float eps, float* scale, float* shift, float* save_mean, float* save_rstd, int C, void* stream) {
    if (!stats || !gamma || !beta || !scale || !shift) return V100_ERR_NULL;
    if (C <= 0 || parts <= 0 || count <= 0) return V100_ERR_SHAPE;
    V100_GGL(bn_finalize_train_kernel, dim3(ceil_div(C, BN_FIN_WPB)), dim3(64 * BN_FIN_WPB), 0, (hipStream_t)stream, stats, parts, (double)count,
                       gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, save_mean, save_rstd, C);
    return v100_launch_status();
}

extern "C" int v100_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                   float eps, float* scale, float* shift, int C, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) return V100_ERR_NULL;
    if (C <= 0) return V100_ERR_SHAPE;
    V100_GGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, eps, scale, shift, C);
    return v100_launch_status();
}

extern "C" int v100_bn_bwd_finalize(const float* partial, int parts, long long count, const float* gamma, const float* mean,
                                    const float* rstd, float* p, float* q, float* r, float* dgamma, float* dbeta, int C, void* stream) {
    if (!partial || !gamma || !mean || !rstd || !p || !q || !r) return V100_ERR_NULL;
    if (C <= 0 || parts <= 0 || count <= 0) return V100_ERR_SHAPE;
    V100_GGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, BN_FIN_WPB)), dim3(64 * BN_FIN_WPB), 0, (hipStream_t)stream, partial, parts, (double)count,
                       gamma, mean, rstd, p, q, r, dgamma, dbeta, C, 0);
    return v100_launch_status();
}

extern "C" int v100_bn_bwd_finalize_frozen(const float* partial, int parts, long long count, const float* gamma, const float* mean,
                                           const float* rstd, float* p, float* q, float* r, float* dgamma, float* dbeta, int C, void* stream) {
    if (!partial || !gamma || !mean || !rstd || !p || !q || !r) return V100_ERR_NULL;
    if (C <= 0 || parts <= 0 || count <= 0) return V100_ERR_SHAPE;
    V100_GGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, BN_FIN_WPB)), dim3(64 * BN_FIN_WPB), 0, (hipStream_t)stream, partial, parts, (double)count,
                       gamma, mean, rstd, p, q, r, dgamma, dbeta, C, 1);
    return v100_launch_status();
}

extern "C" int v100_bn_frozen_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                     float eps, float* scale, float* shift, float* mean, float* rstd, int C, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || !mean || !rstd) return V100_ERR_NULL;
    if (C <= 0) return V100_ERR_SHAPE;
    V100_GGL(bn_frozen_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, eps, scale, shift, mean, rstd, C);
    return v100_launch_status();
}

extern "C" int v100_chan_reduce2(const float* u, const float* v, float* partial, int G, int B, int C, int T, void* stream) {
    if (!u || !partial) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || G > B) return V100_ERR_SHAPE;
    V100_GGL(chan_reduce2_kernel, dim3(C, G), dim3(256), 0, (hipStream_t)stream, u, v, partial, B, C, T, G);
    return v100_launch_status();
}

extern "C" int v100_chan_reduce2_io(const void* u, const void* v, float* partial, int G, int B, int C, int T, int io16, void* stream) {
    if (!u || !v || !partial) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || G > B) return V100_ERR_SHAPE;
    if (io16 == 2) V100_GGL(chan_reduce2_io_kernel<2>, dim3(C, G), dim3(256), 0, (hipStream_t)stream, u, v, partial, B, C, T, G);
    else return V100_ERR_SHAPE;
    return v100_launch_status();
}

// block executor: BatchNorm-3 forward finalisation + block output (+ shadow) in one launch; u = a3 (bf16 when u_bf16)
int chan_affine2_fin(const void* u, const void* v, float* out, void* shadow, int B, int C, int T, int u_bf16, const DwPre& pre, void* stream,
                     int v_bf16) {
    if (!u || (!out && !shadow) || !pre.stats) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || pre.f.mode != 1 || pre.parts <= 0) return V100_ERR_SHAPE;
    if (v_bf16 && !u_bf16) return V100_ERR_SHAPE;                       // level 5 runs on the all-bf16 form only
    hipStream_t st = (hipStream_t)stream;
#define CAF(UB_, SH_, VB_) V100_GGL((chan_affine2_fin_kernel<UB_, SH_, VB_>), dim3(C), dim3(1024), 0, st, u, v, out, (u16*)shadow, B, C, T, pre)
    if (v_bf16) { if (shadow) CAF(true, true, true); else CAF(true, false, true); }
    else if (u_bf16) { if (shadow) CAF(true, true, false); else CAF(true, false, false); }
    else { if (shadow) CAF(false, true, false); else CAF(false, false, false); }
#undef CAF
    return v100_launch_status();
}

// block executor: BatchNorm-3 backward in one pass (sums, coefficients, da3); 0 = shape not covered (more than 8 quads per thread)
int chan_bn3_bwd_fits(int B, int T) { return (long)B * ((T + 3) / 4) <= 1024 * 8; }
int chan_bn3_bwd(const void* u, const void* v, float* partial, void* out, int B, int C, int T, const DwFin& fin, void* stream, int u_bf16) {
    if (!u || !v || !partial || !out || fin.mode != 2) return 0;
    const long n = (long)B * ((T + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define CB3(NQ_) do { if (u_bf16) V100_GGL((chan_bn3_bwd_kernel<NQ_, true>), dim3(C), dim3(1024), 0, st, u, v, partial, out, B, C, T, fin); \
                      else V100_GGL((chan_bn3_bwd_kernel<NQ_, false>), dim3(C), dim3(1024), 0, st, u, v, partial, out, B, C, T, fin); } while (0)
    if (n <= 1024 * 2) CB3(2);
    else if (n <= 1024 * 4) CB3(4);
    else if (n <= 1024 * 6) CB3(6);
    else if (n <= 1024 * 8) CB3(8);
    else return 0;
#undef CB3
    return 1;
}

// block executor: sums of (dy, dy * a3) per channel + BatchNorm-3 backward coefficients in one launch (u fp32, v bf16)
int chan_reduce2_io_fin(const void* u, const void* v, float* partial, int B, int C, int T, const DwFin& fin, void* stream) {
    if (!u || !v || !partial) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || fin.mode != 2) return V100_ERR_SHAPE;
    V100_GGL(chan_reduce2_fin_kernel<2>, dim3(C), dim3(1024), 0, (hipStream_t)stream, u, v, partial, B, C, T, fin);
    return v100_launch_status();
}

extern "C" int v100_chan_affine2_io(const void* u, const void* v, const float* A, const float* Bc, const float* Cc, void* out,
                                    int B, int C, int T, int io16, void* stream) {
    if (!u || !out) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0) return V100_ERR_SHAPE;
    const long rows = (long)B * C;
    const long total = rows * ((T + 3) / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    // forward block output: u = a3 (bf16), v = x (fp32) or null -> y fp32;  backward: u = dy (fp32), v = a3 (bf16) -> da3 (bf16)
    if (io16 == 1) V100_GGL(chan_affine2_io_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, u, v, A, Bc, Cc, out, C, T, rows);
    else if (io16 == 6) V100_GGL(chan_affine2_io_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, st, u, v, A, Bc, Cc, out, C, T, rows);
    else return V100_ERR_SHAPE;
    return v100_launch_status();
}

extern "C" int v100_chan_affine2_shadow(const void* u, const float* v, const float* A, const float* Cc, float* out, void* shadow,
                                        int B, int C, int T, int u_bf16, void* stream) {
    if (!u || !out || !shadow) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0) return V100_ERR_SHAPE;
    const long rows = (long)B * C;
    const long total = rows * ((T + 3) / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    if (u_bf16) V100_GGL(chan_affine2_shadow_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, u, v, A, Cc, out, (u16*)shadow, C, T, rows);
    else V100_GGL(chan_affine2_shadow_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, u, v, A, Cc, out, (u16*)shadow, C, T, rows);
    return v100_launch_status();
}

extern "C" int v100_slab_sum0(const float* partial, int parts, float* out, int C, void* stream) {
    if (!partial || !out) return V100_ERR_NULL;
    if (parts <= 0 || C <= 0) return V100_ERR_SHAPE;
    V100_GGL(slab_sum0_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, (hipStream_t)stream, partial, parts, out, C);
    return v100_launch_status();
}

extern "C" int v100_chan_affine2(const float* u, const float* v, const float* A, const float* Bc, const float* Cc, float* out,
                                 int B, int C, int T, void* stream) {
    if (!u || !out) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0) return V100_ERR_SHAPE;
    const long total = (long)B * C * T;
    long blocks = (total / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    V100_GGL(chan_affine2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u, v, A, Bc, Cc, out, C, T, total);
    return v100_launch_status();
}

extern "C" int v100_mul_scale(const float* u, const float* m, float s, float* out, long long n, void* stream) {
    if (!u || !m || !out) return V100_ERR_NULL;
    if (n <= 0) return V100_ERR_SHAPE;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    V100_GGL(mul_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u, m, s, out, (long)n);
    return v100_launch_status();
}

extern "C" int v100_dropout_fwd(const float* x, long long seed, float p, float* y, void* mask, long long n, void* stream) {
    if (!x || !y || !mask) return V100_ERR_NULL;
    if (n <= 0 || !(p >= 0.f && p < 1.f)) return V100_ERR_SHAPE;
    const long n4 = (n + 3) / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const double t = (double)p * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    V100_GGL(dropout_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (unsigned long long)seed, thresh,
                       1.0f / (1.0f - p), y, (unsigned char*)mask, n4, (long)n);
    return v100_launch_status();
}

extern "C" int v100_dropout_bwd(const float* dy, const void* mask, float p, float* dx, long long n, void* stream) {
    if (!dy || !dx || !mask) return V100_ERR_NULL;
    if (n <= 0 || !(p >= 0.f && p < 1.f)) return V100_ERR_SHAPE;
    const long n4 = (n + 3) / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    V100_GGL(dropout_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, (const unsigned char*)mask,
                       1.0f / (1.0f - p), dx, n4, (long)n);
    return v100_launch_status();
}

extern "C" int v100_transpose_last2(const float* in, float* out, int B, int R, int Cc, void* stream) {
    if (!in || !out) return V100_ERR_NULL;
    if (B <= 0 || R <= 0 || Cc <= 0) return V100_ERR_SHAPE;
    V100_GGL(transpose_last2_kernel, dim3(ceil_div(Cc, 32), ceil_div(R, 32), B), dim3(256), 0, (hipStream_t)stream, in, out, R, Cc);
    return v100_launch_status();
}

extern "C" int v100_bct_to_cm(const float* in, float* out, int B, int C, int T, void* stream) {
    if (!in || !out) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0) return V100_ERR_SHAPE;
    const int P = v100_pitch16(T, B);
    const long long total = (long long)C * B * P;
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    V100_GGL(bct_to_cm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, B, C, T, P);
    return v100_launch_status();
}

extern "C" int v100_cm_to_btc(const float* in, float* out, int B, int C, int T, void* stream) {
    if (!in || !out) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || B > 65535) return V100_ERR_SHAPE;
    const int P = v100_pitch16(T, B);
    V100_GGL(cm_to_btc_kernel, dim3(ceil_div(C, 32), ceil_div(T, 32), B), dim3(256), 0, (hipStream_t)stream, in, out, B, C, T, P);
    return v100_launch_status();
}

extern "C" int v100_embedding_bct(const long long* idx, const float* table, float* out, int B, int V, int C, int T, void* stream) {
    if (!idx || !table || !out) return V100_ERR_NULL;
    if (B <= 0 || V <= 0 || C <= 0 || T <= 0) return V100_ERR_SHAPE;
    V100_GGL(embedding_bct_kernel, dim3(ceil_div(C, 32), ceil_div(T, 32), B), dim3(256), 0, (hipStream_t)stream, idx, table, out, V, C, T);
    return v100_launch_status();
}

extern "C" int v100_embedding_bwd(const long long* idx, const float* g, float* dtable, int B, int V, int C, int T, void* stream) {
    if (!idx || !g || !dtable) return V100_ERR_NULL;
    if (B <= 0 || V <= 0 || C <= 0 || T <= 0) return V100_ERR_SHAPE;
    const long total = (long)B * C * T;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipMemsetAsync(dtable, 0, (size_t)V * C * sizeof(float), (hipStream_t)stream);
    V100_GGL(embedding_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, idx, g, dtable, V, C, T, total);
    return v100_launch_status();
}
