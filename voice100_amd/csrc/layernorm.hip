// K11: channel LayerNorm + GELU over [B, C, T] activations, and the im2col / col2im moves of the dense k-tap
// convolutions in front of it (the reference's v2 conv blocks).
//
// Replaces, in ConvLayerBlock / ConvTransposeLayerBlock (voice100/models/_layers_v2.py:29-89):
//     x = conv(x); x = x.transpose(-2,-1); x = layer_norm(x); x = x.transpose(-2,-1); x = gelu(x)
// i.e. nn.LayerNorm(C) (eps 1e-5, affine) over the CHANNEL axis of every (b, t) column followed by the exact
// (erf) GELU.  The two transposes never happen here: one workgroup owns a [C x 32] tile of a [C, T] plane,
// thread (cr, tq) = (tid / 8, tid % 8) holds channels cr, cr+32, ... for 4 consecutive t in registers
// (16 float4 at C = 512), so every global access is a 128-byte row segment, the tile is read ONCE, and the
// column statistics are a 32-way cross-thread reduction through 4 KB of LDS.  Mean first, then the centred
// second moment (two passes over registers): LayerNorm parity at 1e-4 does not survive E[x^2] - E[x]^2.
// HBM-bound: 8 bytes per element forward (read + write), 12 backward.
#include "common.h"

#define LN_TT 32            // columns (t) per workgroup
#define LN_ROWS 32          // channel rows covered per pass by the 256 threads (8 threads x float4 per row)
#define LN_MAX_NI 32        // C <= 1024

__device__ __forceinline__ float gelu_exact(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float z) {
    const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
    return cdf + z * pdf;
}

// 4 consecutive floats of a row of length T starting at t (any alignment); zero past the end
__device__ __forceinline__ f32x4 ln_load4(const float* __restrict__ row, int t, int T) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t + 3 < T) v = *reinterpret_cast<const f32x4u*>(row + t);
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t + e < T) v[e] = row[t + e];
    }
    return v;
}
__device__ __forceinline__ void ln_store4(float* __restrict__ row, int t, int T, f32x4 v) {
    if (t + 3 < T) *reinterpret_cast<f32x4u*>(row + t) = v;
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t + e < T) row[t + e] = v[e];
    }
}

// sum over the 32 channel-row groups (cr) for each of the 32 columns of the tile; result for this thread's 4 columns
__device__ __forceinline__ f32x4 ln_col_reduce(f32x4 part, float (*red)[LN_TT + 1], int cr, int tq) {
    __syncthreads();                       // previous use of red[] is over
#pragma unroll
    for (int e = 0; e < 4; ++e) red[cr][4 * tq + e] = part[e];
    __syncthreads();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int r = 0; r < LN_ROWS; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += red[r][4 * tq + e];
    }
    return s;
}

template <int NI>
__global__ __launch_bounds__(256) void ln_gelu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float* __restrict__ out,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int C, int T) {
    __shared__ float red[LN_ROWS][LN_TT + 1];
    const int tid = threadIdx.x, tq = tid & 7, cr = tid >> 3;
    const int b = blockIdx.y, t = blockIdx.x * LN_TT + 4 * tq;
    const float* yb = y + (size_t)b * C * T;
    float* ob = out + (size_t)b * C * T;
    f32x4 v[NI];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = cr + LN_ROWS * i;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < C) v[i] = ln_load4(yb + (size_t)c * T, t, T);
        s += v[i];
    }
    s = ln_col_reduce(s, red, cr, tq);
    const float inv_c = 1.f / (float)C;
    const f32x4 mu = s * inv_c;
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (cr + LN_ROWS * i < C) {
            const f32x4 d = v[i] - mu;
            q += d * d;
        }
    }
    q = ln_col_reduce(q, red, cr, tq);
    f32x4 rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) rs[e] = 1.f / sqrtf(q[e] * inv_c + eps);     // biased variance, as nn.LayerNorm
    if (cr == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t + e < T) {
                mean[(size_t)b * T + t + e] = mu[e];
                rstd[(size_t)b * T + t + e] = rs[e];
            }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = cr + LN_ROWS * i;
        if (c < C) {
            const float g = gamma[c], be = beta[c];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = gelu_exact(fmaf((v[i][e] - mu[e]) * rs[e], g, be));
            ln_store4(ob + (size_t)c * T, t, T, o);
        }
    }
}

// Backward.  With xh = (y - mean) * rstd, z = xh*gamma + beta, g = dout * gelu'(z), a = g * gamma:
//   dy = rstd * (a - mean_c(a) - xh * mean_c(a * xh));   dgamma[c] = sum_{b,t} g * xh;   dbeta[c] = sum_{b,t} g
// The per-channel sums of this tile go to partial[part][c][0..1] (part = b * n_ttiles + tile): deterministic slab.
template <int NI>
__global__ __launch_bounds__(256) void ln_gelu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          float* __restrict__ dy, float* __restrict__ partial, int C, int T) {
    __shared__ float red[LN_ROWS][LN_TT + 1];
    const int tid = threadIdx.x, tq = tid & 7, cr = tid >> 3;
    const int b = blockIdx.y, t = blockIdx.x * LN_TT + 4 * tq;
    const size_t plane = (size_t)b * C * T;
    const size_t part = (size_t)b * gridDim.x + blockIdx.x;
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (t + e < T) { mu[e] = mean[(size_t)b * T + t + e]; rs[e] = rstd[(size_t)b * T + t + e]; }
    f32x4 a[NI], xh[NI];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = cr + LN_ROWS * i;
        a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        xh[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            const f32x4 yv = ln_load4(y + plane + (size_t)c * T, t, T);
            const f32x4 dv = ln_load4(dout + plane + (size_t)c * T, t, T);     // zero past T: those columns add nothing
            const float g = gamma[c], be = beta[c];
            float dg = 0.f, db = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = (yv[e] - mu[e]) * rs[e];
                const float gz = dv[e] * gelu_grad(fmaf(x, g, be));
                xh[i][e] = x;
                a[i][e] = gz * g;
                dg = fmaf(gz, x, dg);
                db += gz;
            }
            // the 8 threads of a row (consecutive lanes) -> one partial per (tile, channel)
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) { dg += __shfl_xor(dg, off, 64); db += __shfl_xor(db, off, 64); }
            if (tq == 0) { partial[(part * C + c) * 2 + 0] = dg; partial[(part * C + c) * 2 + 1] = db; }
            s1 += a[i];
            s2 += a[i] * xh[i];
        }
    }
    s1 = ln_col_reduce(s1, red, cr, tq);
    s2 = ln_col_reduce(s2, red, cr, tq);
    const float inv_c = 1.f / (float)C;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = cr + LN_ROWS * i;
        if (c < C) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs[e] * (a[i][e] - s1[e] * inv_c - xh[i][e] * (s2[e] * inv_c));
            ln_store4(dy + plane + (size_t)c * T, t, T, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// im2col for nn.Conv1d(Cin, Cout, k, stride, padding) as ONE GEMM with K = k*Cin (tap-major rows):
//   cols[b][j*Cin + c][u] = x[b][c][u*stride - pad + j]   (0 outside [0, Tin)),  u in [0, Tout)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, int Cin, int Tin,
                                                     int Tout, int k, int stride, int pad, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int u = (int)(i % Tout);
        const long r = i / Tout;
        const int row = (int)(r % ((long)k * Cin));
        const int b = (int)(r / ((long)k * Cin));
        const int j = row / Cin, c = row - j * Cin;
        const int ti = u * stride - pad + j;
        cols[i] = (ti >= 0 && ti < Tin) ? x[((size_t)b * Cin + c) * Tin + ti] : 0.f;
    }
}

// col2im (its adjoint, the backward-data of that conv): dx[b][c][ti] = sum_j dcols[b][j*Cin + c][(ti + pad - j) / stride]
// over the taps where the division is exact and the quotient lies in [0, Tout): a gather, so no atomics.
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int Cin, int Tin,
                                                     int Tout, int k, int stride, int pad, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ti = (int)(i % Tin);
        const long r = i / Tin;
        const int c = (int)(r % Cin);
        const int b = (int)(r / Cin);
        float s = 0.f;
        for (int j = 0; j < k; ++j) {
            const int num = ti + pad - j;
            if (num < 0 || num % stride != 0) continue;
            const int u = num / stride;
            if (u < Tout) s += dcols[((size_t)b * k * Cin + (size_t)j * Cin + c) * Tout + u];
        }
        dx[i] = s;
    }
}

// out0[c] = sum_p partial[p][c][0], out1[c] = sum_p partial[p][c][1]: one wave per channel, fixed order (deterministic)
__global__ __launch_bounds__(256) void slab_sum2_kernel(const float* __restrict__ partial, int parts, float* __restrict__ out0,
                                                        float* __restrict__ out1, int C) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    for (int p = lane; p < parts; p += 64) {
        s0 += partial[((size_t)p * C + c) * 2 + 0];
        s1 += partial[((size_t)p * C + c) * 2 + 1];
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) { out0[c] = s0; out1[c] = s1; }
}

static inline unsigned ln_grid_for(long total) {
    long g = (total + 255) / 256;
    if (g > 256L * 64) g = 256L * 64;
    return (unsigned)(g < 1 ? 1 : g);
}

extern "C" int v100_ln_num_parts(int B, int T) { return B * ceil_div(T, LN_TT); }

extern "C" int v100_ln_gelu_fwd(const float* y, const float* gamma, const float* beta, float eps, float* out, float* mean,
                                float* rstd, int B, int C, int T, void* stream) {
    if (!y || !gamma || !beta || !out || !mean || !rstd) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || C > LN_ROWS * LN_MAX_NI) return V100_ERR_SHAPE;
    dim3 grid(ceil_div(T, LN_TT), B);
    hipStream_t st = (hipStream_t)stream;
    const int ni = ceil_div(C, LN_ROWS);
    if (ni <= 8) V100_GGL((ln_gelu_fwd_kernel<8>), grid, dim3(256), 0, st, y, gamma, beta, eps, out, mean, rstd, C, T);
    else if (ni <= 16) V100_GGL((ln_gelu_fwd_kernel<16>), grid, dim3(256), 0, st, y, gamma, beta, eps, out, mean, rstd, C, T);
    else V100_GGL((ln_gelu_fwd_kernel<32>), grid, dim3(256), 0, st, y, gamma, beta, eps, out, mean, rstd, C, T);
    return v100_launch_status();
}

extern "C" int v100_ln_gelu_bwd(const float* dout, const float* y, const float* gamma, const float* beta, const float* mean,
                                const float* rstd, float* dy, float* partial, int B, int C, int T, void* stream) {
    if (!dout || !y || !gamma || !beta || !mean || !rstd || !dy || !partial) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || C > LN_ROWS * LN_MAX_NI) return V100_ERR_SHAPE;
    dim3 grid(ceil_div(T, LN_TT), B);
    hipStream_t st = (hipStream_t)stream;
    const int ni = ceil_div(C, LN_ROWS);
    if (ni <= 8) V100_GGL((ln_gelu_bwd_kernel<8>), grid, dim3(256), 0, st, dout, y, gamma, beta, mean, rstd, dy, partial, C, T);
    else if (ni <= 16) V100_GGL((ln_gelu_bwd_kernel<16>), grid, dim3(256), 0, st, dout, y, gamma, beta, mean, rstd, dy, partial, C, T);
    else V100_GGL((ln_gelu_bwd_kernel<32>), grid, dim3(256), 0, st, dout, y, gamma, beta, mean, rstd, dy, partial, C, T);
    return v100_launch_status();
}

extern "C" int v100_slab_sum2(const float* partial, int parts, float* out0, float* out1, int C, void* stream) {
    if (!partial || !out0 || !out1) return V100_ERR_NULL;
    if (parts <= 0 || C <= 0) return V100_ERR_SHAPE;
    V100_GGL(slab_sum2_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, (hipStream_t)stream, partial, parts, out0, out1, C);
    return v100_launch_status();
}

extern "C" int v100_im2col(const float* x, float* cols, int B, int Cin, int Tin, int Tout, int k, int stride, int pad, void* stream) {
    if (!x || !cols) return V100_ERR_NULL;
    if (B <= 0 || Cin <= 0 || Tin <= 0 || Tout <= 0 || k <= 0 || stride <= 0 || pad < 0) return V100_ERR_SHAPE;
    if (Tout != (Tin + 2 * pad - k) / stride + 1) return V100_ERR_SHAPE;
    const long total = (long)B * k * Cin * Tout;
    V100_GGL(im2col_kernel, dim3(ln_grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, cols, Cin, Tin, Tout, k, stride, pad, total);
    return v100_launch_status();
}

extern "C" int v100_col2im(const float* dcols, float* dx, int B, int Cin, int Tin, int Tout, int k, int stride, int pad, void* stream) {
    if (!dcols || !dx) return V100_ERR_NULL;
    if (B <= 0 || Cin <= 0 || Tin <= 0 || Tout <= 0 || k <= 0 || stride <= 0 || pad < 0) return V100_ERR_SHAPE;
    if (Tout != (Tin + 2 * pad - k) / stride + 1) return V100_ERR_SHAPE;
    const long total = (long)B * Cin * Tin;
    V100_GGL(col2im_kernel, dim3(ln_grid_for(total)), dim3(256), 0, (hipStream_t)stream, dcols, dx, Cin, Tin, Tout, k, stride, pad, total);
    return v100_launch_status();
}
