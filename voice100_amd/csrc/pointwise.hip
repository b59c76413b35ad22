// K1: pointwise (1x1) Conv1d as a per-utterance GEMM on the MI355X matrix cores.
//
//   forward / backward-data ("NN"):  Y[b][m][t] = sum_k A[m][k] * f(X[b][k][t])
//   backward-weight          ("NT"):  dW[m][k]   = sum_{b,t} g(G[b][m][t]) * f(X[b][k][t])
//
// Replaces nn.Conv1d(kernel_size=1) (+ BatchNorm1d / ReLU6 / residual add around it) of the
// reference's ConvBNActivate "pw" and "pw-linear" stages (voice100/models/asr.py:47,51-52) and the
// 1x1 heads (asr.py:91, tts.py:26,77).  Activations stay fp32 [B, C, T] in HBM; f() is applied
// while staging the tile into LDS, so the BatchNorm affine + ReLU6 of the producer (forward) or the
// BatchNorm-backward affine of two tensors (backward) never round-trips through HBM.  The epilogue
// emits the per-channel partial sums the next BatchNorm needs (training statistics or the two
// backward reductions) into a deterministic [parts][M][2] slab.
//
// Two precisions of the same tiling (DESIGN.md "K1"):
//   fp32 : v_mfma_f32_32x32x2_f32, exact fp32 (bitwise an fmaf chain)  -- parity path
//   bf16 : v_mfma_f32_32x32x16_bf16, operands rounded to bf16 while staging, fp32 accumulate
// 128x128 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles (64 acc VGPRs).
// 1-D grid with an XCD-aware remap: the M-tiles that share one X tile run on one XCD (one L2).
#include "pointwise_common.h"
#include <stdlib.h>
#include "timing.h"

void pw_launch_gemm_f32(const PwParams& p, dim3 grid, hipStream_t st);
void pw_launch_gemm_bf16(const PwParams& p, dim3 grid, hipStream_t st);
bool pw_launch_gemm_f16(const PwParams& p, dim3 grid, hipStream_t st);
void pw_launch_wgrad_f32(const WgParams& p, dim3 grid, hipStream_t st);
void pw_launch_wgrad_bf16(const WgParams& p, dim3 grid, hipStream_t st);
bool pw_launch_gemm_bf16_io(const PwParams& p, hipStream_t st);
bool pw_launch_wgrad_bf16_io(const WgParams& p, dim3 grid, hipStream_t st);
bool pw_taps_fit_bf16(int B, int M, int cx, int ntap, int T, int Tx);
bool pw_launch_gemm_taps_bf16(const PwParams& p, dim3 grid, hipStream_t st);
bool pw_launch_wgrad_taps_bf16(const WgParams& p, dim3 grid, hipStream_t st);
void pw_launch_gemm_taps_f32(const PwParams& p, dim3 grid, hipStream_t st);
void pw_launch_wgrad_taps_f32(const WgParams& p, dim3 grid, hipStream_t st);

// ---------------------------------------------------------------------------------------------
__global__ void pw_slab_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int S, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int g = 0; g < S; ++g) s += partial[(size_t)g * n + i];
    out[i] = s;
}
// the same sum (slab order 0 .. S-1: bit-identical) in 16-byte pieces, eight slabs' loads in flight per lane; n % 4 == 0
__global__ __launch_bounds__(256) void pw_slab_reduce4_kernel(const f32x4* __restrict__ partial, f32x4* __restrict__ out, int S, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int g = 0;
    for (; g + 8 <= S; g += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(partial + (size_t)(g + u) * n4 + i);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; g < S; ++g) s += __builtin_nontemporal_load(partial + (size_t)g * n4 + i);
    out[i] = s;
}
static void pw_slab_reduce(const float* partial, float* out, int S, long n, hipStream_t st) {
    if ((n & 3) == 0 && ((((size_t)partial) | ((size_t)out)) & 15) == 0) V100_GGL(pw_slab_reduce4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (const f32x4*)partial, (f32x4*)out, S, n / 4);
    else V100_GGL(pw_slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, out, S, n);
}

// fp32 [rows][cols] -> bf16 [rows][cols] and/or transposed copies (weights are tiny: <= 1M elements)
__global__ void weight_prep_kernel(const float* __restrict__ w, int rows, int cols, u16* __restrict__ w_bf,
                                   float* __restrict__ wt, u16* __restrict__ wt_bf, int f16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float v = w[i];
    if (w_bf) w_bf[i] = f16 ? f2h(v) : f2bf(v);
    if (wt) wt[(size_t)c * rows + r] = v;
    if (wt_bf) wt_bf[(size_t)c * rows + r] = f16 ? f2h(v) : f2bf(v);
}

extern "C" int v100_pw_num_parts(int B, int T) { return B * ceil_div(T, PW_BN); }

extern "C" int v100_pw_wgrad_splits(int B, int M, int K) {
    // enough (tile, split) workgroups to fill the chip, capped by the batch
    const int tiles = ceil_div(M, PW_BM) * ceil_div(K, PW_BN);
    static const int target = [] { const char* e = getenv("V100_WG_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    int S = ceil_div(target, tiles);
    if (S > B) {
        // fewer (tile, utterance) pairs than the chip has CUs (the 64 -> 256 opener's two tiles, the vocabulary head): split each
        // utterance's t range too -- S = B * TS, TS a power of two <= 8 with one workgroup per CU as the aim (WgSpan, pointwise_common.h)
        int TS = 1;
        while (TS < 8 && tiles * B * TS * 2 <= 256) TS *= 2;
        S = B * TS;
    }
    if (S < 1) S = 1;
    return S;
}

extern "C" int v100_weight_prep(const float* w, int rows, int cols, void* w_bf, float* wt, void* wt_bf, void* stream) {
    if (!w) return V100_ERR_NULL;
    if (rows <= 0 || cols <= 0) return V100_ERR_SHAPE;
    const long n = (long)rows * cols;
    V100_GGL(weight_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, rows, cols,
                       (u16*)w_bf, wt, (u16*)wt_bf, 0);
    return v100_launch_status();
}

// fp16 copy of a weight (inference precision "fp16"): w16[i] = half(w[i])
extern "C" int v100_weight_prep_f16(const float* w, int rows, int cols, void* w16, void* stream) {
    if (!w || !w16) return V100_ERR_NULL;
    if (rows <= 0 || cols <= 0) return V100_ERR_SHAPE;
    const long n = (long)rows * cols;
    V100_GGL(weight_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, rows, cols,
                       (u16*)w16, (float*)nullptr, (u16*)nullptr, 1);
    return v100_launch_status();
}

// Small-K GEMM on the VALU: Y[b][m][t] = sum_{k < K} A[m][k] * X[b][k][t] (+ bias[m]) for K <= 32 -- the data gradient of the
// vocabulary head (asr.py:91: dx[512 x T'] = W^T[512 x 29] dy[29 x T']), which the MFMA kernels ran through their generic
// (scalar-safe, K padded to 64) path at 34 us per step for 1 GFLOP.  29 FMAs per output are nothing; the kernel is the 33 MB store.
// A workgroup owns a [64 m x 128 t] tile: A tile in LDS, a thread holds 8 m x 4 t accumulators.  fmt 0: fp32 operands (exact fp32,
// fmaf chain in k order); fmt 1: operands rounded to bf16 like the MFMA path, fp32 accumulate.  T % 4 == 0.
// MASK: the vocabulary head's data gradient with Dropout's backward in the epilogue (keep: the byte mask of v100_dropout_fwd over
// Y's [B][M][T] elements; scale = 1 / (1 - p)) -- the separate pass read and wrote the 33 MB gradient once more.
template <bool MASK>
__global__ __launch_bounds__(256) void pw_smallk_kernel(PwParams p, const unsigned char* __restrict__ keep, float scale) {
    __shared__ float As[64][33];
    const int K = p.K, M = p.M, T = p.T;
    const int nmt = (M + 63) >> 6, ntt = (T + 127) >> 7;
    int v = blockIdx.x;
    const int mt = v % nmt; v /= nmt;
    const int tt = v % ntt;
    const int b = v / ntt;
    const int m0 = mt * 64, t0 = tt * 128;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {          // columns K .. 31 are zero: the k loop below runs in whole groups of 8
        const int r = i >> 5, k = i & 31;
        float a = 0.f;
        if (m0 + r < M && k < K) a = p.fmt ? __builtin_bit_cast(float, (unsigned)p.Abf[(size_t)(m0 + r) * K + k] << 16) : p.A[(size_t)(m0 + r) * K + k];
        As[r][k] = a;
    }
    __syncthreads();
    const int tq = threadIdx.x & 31, mi = threadIdx.x >> 5;
    const int t = t0 + 4 * tq;
    if (t >= T) return;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xb = p.X + (size_t)b * K * T + t;
    for (int k0 = 0; k0 < K; k0 += 8) {
        f32x4 xs[8];                                  // eight rows requested before the first is used (rows past K: row K-1 again, times zero)
#pragma unroll
        for (int u = 0; u < 8; ++u) xs[u] = *reinterpret_cast<const f32x4*>(xb + (size_t)min(k0 + u, K - 1) * T);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x4 x = xs[u];
            if (p.fmt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = __builtin_bit_cast(float, (unsigned)f2bf(x[e]) << 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = As[8 * mi + j][k0 + u];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(a, x[e], acc[j][e]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = m0 + 8 * mi + j;
        if (m < M) {
            f32x4 o = acc[j];
            if (p.bias) { const float bs = p.bias[m]; o += f32x4{bs, bs, bs, bs}; }
            if (MASK) {
                const unsigned mk = *reinterpret_cast<const unsigned*>(keep + ((size_t)b * M + m) * T + t);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ((mk >> (8 * e)) & 1u) ? o[e] * scale : 0.f;       // dropout_bwd_kernel's arithmetic
            }
            *reinterpret_cast<f32x4*>(p.Y + ((size_t)b * M + m) * T + t) = o;
        }
    }
}

extern "C" int v100_pw_gemm(const float* A, const void* A_bf16, const float* X, const float* X2, const float* xa,
                            const float* xb, const float* xc, int x_mode, float* Y, const float* bias, const float* ea,
                            const float* eb, const float* R, int epi_mode, float* stats, int B, int M, int K, int T,
                            int use_bf16, void* stream) {
    if (!X || !Y) return V100_ERR_NULL;
    if (use_bf16 ? !A_bf16 : !A) return V100_ERR_NULL;
    if (B <= 0 || M <= 0 || K <= 0 || T <= 0) return V100_ERR_SHAPE;
    if (x_mode < 0 || x_mode > 2 || epi_mode < 0 || epi_mode > 5) return V100_ERR_SHAPE;
    if (x_mode != PW_X_NONE && (!xa || !xb)) return V100_ERR_NULL;
    if (x_mode == PW_X_AFFINE2 && (!X2 || !xc)) return V100_ERR_NULL;
    if ((epi_mode == PW_EPI_AFFINE_RELU6 || epi_mode == PW_EPI_AFFINE_RES || epi_mode == PW_EPI_MASK_STATS) && (!ea || !eb)) return V100_ERR_NULL;
    if ((epi_mode == PW_EPI_MASK_STATS || epi_mode == PW_EPI_ADD) && !R) return V100_ERR_NULL;
    if ((epi_mode == PW_EPI_STATS || epi_mode == PW_EPI_MASK_STATS) && !stats) return V100_ERR_NULL;
    const int nmt = ceil_div(M, PW_BM), ntt = ceil_div(T, PW_BN);
    if (use_bf16 < 0 || use_bf16 > 2) return V100_ERR_SHAPE;
    PwParams p{A, (const u16*)A_bf16, X, X2, xa, xb, xc, Y, bias, ea, eb, R, stats, B, M, K, T, x_mode, epi_mode, nmt, ntt, use_bf16};
    const long nwg = (long)nmt * ntt * B;
    if (nwg > 0x7fffffffL) return V100_ERR_SHAPE;
    dim3 grid((unsigned)nwg);
    hipStream_t st = (hipStream_t)stream;
    V100TimedRegion timed(V100_T_PW_GEMM, st);
    if (K <= 32 && use_bf16 != 2 && x_mode == PW_X_NONE && epi_mode == PW_EPI_STORE && (T & 3) == 0 && M >= 64 &&
        (size_t)B * K * T * 4 < 0x7fffff00ull) {                                             // see pw_smallk_kernel
        const long nw = (long)((M + 63) >> 6) * ((T + 127) >> 7) * B;
        V100_GGL(pw_smallk_kernel<false>, dim3((unsigned)nw), dim3(256), 0, st, p, (const unsigned char*)nullptr, 1.f);
        return v100_launch_status();
    }
    if (use_bf16 == 2) { if (!pw_launch_gemm_f16(p, grid, st)) return V100_ERR_SHAPE; }     // fp16: inference combinations only
    else if (use_bf16) pw_launch_gemm_bf16(p, grid, st);
    else pw_launch_gemm_f32(p, grid, st);
    return v100_launch_status();
}

static bool pw_smallk_ok(int B, int M, int K, int T, int use_bf16) {
    return B > 0 && T > 0 && K > 0 && K <= 32 && (use_bf16 == 0 || use_bf16 == 1) && (T & 3) == 0 && M >= 64 && (size_t)B * K * T * 4 < 0x7fffff00ull &&
           (long)((M + 63) >> 6) * ((T + 127) >> 7) * B <= 0x7fffffffL;
}
extern "C" int v100_pw_gemm_dropmask_supported(int B, int M, int K, int T, int use_bf16) { return pw_smallk_ok(B, M, K, T, use_bf16) ? 1 : 0; }

extern "C" int v100_pw_gemm_dropmask(const float* A, const void* A_bf16, const float* X, float* Y, const void* keep, float pdrop, int B,
                                     int M, int K, int T, int use_bf16, void* stream) {
    if (!X || !Y || !keep) return V100_ERR_NULL;
    if (use_bf16 ? !A_bf16 : !A) return V100_ERR_NULL;
    if (!pw_smallk_ok(B, M, K, T, use_bf16) || !(pdrop >= 0.f && pdrop < 1.f)) return V100_ERR_SHAPE;
    PwParams p{A, (const u16*)A_bf16, X, nullptr, nullptr, nullptr, nullptr, Y, nullptr, nullptr, nullptr, nullptr, nullptr, B, M, K, T,
               PW_X_NONE, PW_EPI_STORE, ceil_div(M, PW_BM), ceil_div(T, PW_BN), use_bf16};
    hipStream_t st = (hipStream_t)stream;
    V100TimedRegion timed(V100_T_PW_GEMM, st);
    const long nw = (long)((M + 63) >> 6) * ((T + 127) >> 7) * B;
    V100_GGL(pw_smallk_kernel<true>, dim3((unsigned)nw), dim3(256), 0, st, p, (const unsigned char*)keep, 1.0f / (1.0f - pdrop));
    return v100_launch_status();
}

extern "C" int v100_pw_wgrad(const float* G, const float* G2, const float* ga, const float* gb, const float* gc, int g_mode,
                             const float* X, const float* xa, const float* xb, int x_mode, float* partial, float* dW,
                             int S, int B, int M, int K, int T, int use_bf16, void* stream) {
    if (!G || !X || !partial || !dW) return V100_ERR_NULL;
    if (B <= 0 || M <= 0 || K <= 0 || T <= 0 || S <= 0 || (S > B && (S % B != 0 || S / B > 64))) return V100_ERR_SHAPE;
    if (g_mode < 0 || g_mode > 2 || x_mode < 0 || x_mode > 1) return V100_ERR_SHAPE;
    if (g_mode != PW_X_NONE && (!ga || !gb)) return V100_ERR_NULL;
    if (g_mode == PW_X_AFFINE2 && (!G2 || !gc)) return V100_ERR_NULL;
    if (x_mode != PW_X_NONE && (!xa || !xb)) return V100_ERR_NULL;
    const int nmt = ceil_div(M, PW_BM), nkt = ceil_div(K, PW_BN);
    WgParams p{G, G2, ga, gb, gc, X, xa, xb, partial, B, M, K, T, S, g_mode, x_mode, nmt, nkt};
    dim3 grid((unsigned)(nmt * nkt * S));
    hipStream_t st = (hipStream_t)stream;
    V100TimedRegion timed(V100_T_PW_WGRAD, st);
    if (use_bf16) pw_launch_wgrad_bf16(p, grid, st);
    else pw_launch_wgrad_f32(p, grid, st);
    const long n = (long)M * K;
    pw_slab_reduce(partial, dW, S, n, st);
    return v100_launch_status();
}


// bf16-operand GEMMs over tensors of which some are STORED as bf16 (io16: PW_IO_* / WG_IO_* masks, include/voice100_hip.h):
// the block executor's "act16" mode.  Same semantics as v100_pw_gemm / v100_pw_wgrad; V100_ERR_SHAPE when the
// (modes, mask, shape) combination has no kernel -- there is no fallback, the caller picks fp32 storage instead.
extern "C" int v100_pw_gemm_io(const void* A_bf16, const void* X, const void* X2, const float* xa, const float* xb, const float* xc,
                               int x_mode, void* Y, const float* ea, const float* eb, const void* R, int epi_mode, float* stats,
                               int B, int M, int K, int T, int io16, void* stream) {
    if (!A_bf16 || !X || !Y) return V100_ERR_NULL;
    if (B <= 0 || M <= 0 || K <= 0 || T <= 0 || io16 <= 0 || io16 > 31 || io16 == PW_IO_F16) return V100_ERR_SHAPE;
    if (x_mode != PW_X_NONE && (!xa || !xb)) return V100_ERR_NULL;
    if (x_mode == PW_X_AFFINE2 && (!X2 || !xc)) return V100_ERR_NULL;
    if (epi_mode == PW_EPI_MASK_STATS && (!ea || !eb)) return V100_ERR_NULL;
    if ((epi_mode == PW_EPI_MASK_STATS || epi_mode == PW_EPI_ADD) && !R) return V100_ERR_NULL;
    if ((epi_mode == PW_EPI_STATS || epi_mode == PW_EPI_MASK_STATS) && !stats) return V100_ERR_NULL;
    PwParams p{nullptr, (const u16*)A_bf16, (const float*)X, (const float*)X2, xa, xb, xc, (float*)Y, nullptr, ea, eb, (const float*)R, stats,
               B, M, K, T, x_mode, epi_mode, ceil_div(M, PW_BM), ceil_div(T, PW_BN), (io16 & PW_IO_F16) ? 2 : 1, 0, 0, 0, 0u, io16};
    hipStream_t st = (hipStream_t)stream;
    V100TimedRegion timed(V100_T_PW_GEMM, st);
    if (!pw_launch_gemm_bf16_io(p, st)) return V100_ERR_SHAPE;
    return v100_launch_status();
}

extern "C" int v100_pw_wgrad_io(const void* G, const void* G2, const float* ga, const float* gb, const float* gc, int g_mode,
                                const void* X, const float* xa, const float* xb, int x_mode, float* partial, float* dW, int S, int B,
                                int M, int K, int T, int io16, void* stream) {
    if (!G || !X || !partial || !dW) return V100_ERR_NULL;
    if (B <= 0 || M <= 0 || K <= 0 || T <= 0 || S <= 0 || (S > B && (S % B != 0 || S / B > 64)) || io16 <= 0 || io16 > 7) return V100_ERR_SHAPE;
    if (g_mode != PW_X_NONE && (!ga || !gb)) return V100_ERR_NULL;
    if (g_mode == PW_X_AFFINE2 && (!G2 || !gc)) return V100_ERR_NULL;
    if (x_mode != PW_X_NONE && (!xa || !xb)) return V100_ERR_NULL;
    const int nmt = ceil_div(M, PW_BM), nkt = ceil_div(K, PW_BN);
    WgParams p{(const float*)G, (const float*)G2, ga, gb, gc, (const float*)X, xa, xb, partial, B, M, K, T, S, g_mode, x_mode, nmt, nkt,
               0, 0, 0, 0, 0, 0u, io16};
    dim3 grid((unsigned)(nmt * nkt * S));
    hipStream_t st = (hipStream_t)stream;
    V100TimedRegion timed(V100_T_PW_WGRAD, st);
    if (!pw_launch_wgrad_bf16_io(p, grid, st)) return V100_ERR_SHAPE;
    const long n = (long)M * K;
    pw_slab_reduce(partial, dW, S, n, st);
    return v100_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Dense k-tap Conv1d / ConvTranspose1d phases as ONE GEMM over a zero-padded copy of the input (no im2col):
//   Y[b][m][t] = sum_{tap, c} A[m][tap * cx + c] * Xp[b][c][t + shift[tap]]            (+ bias[m])  (+ R[b][m][t])
//   dA[m][tap * cx + c] = sum_{b, t < T} G[b][m][g_off + t] * Xp[b][c][t + shift[tap]]
// Xp [B][cx][Tx] is written by v100_pad_copy; every t + shift[tap] (t < T) must lie inside a row (Tx >= T + max shift).

// dst[b][c][lpad + i] = src[b][c][src_off + i * src_step] for i < n, zero elsewhere in the Tx-long row: zero padding,
// optional de-interleave (src_step 2) of a ConvTranspose gradient into its even / odd output phases
__global__ void pad_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int Tsrc, int src_step, int src_off,
                                int n, int Tx, int lpad, long rows) {
    const long row = (long)blockIdx.y * gridDim.z + blockIdx.z;
    if (row >= rows) return;
    const float* s = src + row * Tsrc;
    float* d = dst + row * Tx;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < Tx; j += gridDim.x * blockDim.x) {
        const int i = j - lpad;
        d[j] = (i >= 0 && i < n) ? s[src_off + (long)i * src_step] : 0.f;
    }
}

extern "C" int v100_pad_copy(const float* src, float* dst, int B, int C, int Tsrc, int src_step, int src_off, int n, int Tx,
                             int lpad, void* stream) {
    if (!src || !dst) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || Tsrc <= 0 || src_step <= 0 || src_off < 0 || n < 0 || lpad < 0 || Tx < lpad + n) return V100_ERR_SHAPE;
    if (n > 0 && src_off + (long)(n - 1) * src_step >= Tsrc) return V100_ERR_SHAPE;
    const long rows = (long)B * C;
    const unsigned gz = (unsigned)(rows < 32768 ? rows : 32768), gy = (unsigned)((rows + gz - 1) / gz);
    if (gy > 65535u) return V100_ERR_SHAPE;
    const unsigned gx = (unsigned)(ceil_div(Tx, 256) < 4 ? ceil_div(Tx, 256) : 4);
    V100_GGL(pad_copy_kernel, dim3(gx, gy, gz), dim3(256), 0, (hipStream_t)stream, src, dst, Tsrc, src_step, src_off, n, Tx,
                       lpad, rows);
    return v100_launch_status();
}

static unsigned pw_pack_shifts(const int* shifts, int ntap, int T, int Tx, bool& ok) {
    unsigned packed = 0;
    ok = shifts && ntap >= 1 && ntap <= 8;
    for (int i = 0; ok && i < ntap; ++i) {
        if (shifts[i] < 0 || shifts[i] > 15 || (long)T + shifts[i] > Tx) ok = false;
        else packed |= (unsigned)shifts[i] << (4 * i);
    }
    return packed;
}

// 1 when v100_pw_gemm_taps / v100_pw_wgrad_taps run this shape at the given precision (0 fp32, 1 bf16, 2 fp16)
extern "C" int v100_pw_taps_supported(int B, int M, int cx, int ntap, int T, int Tx, int use_bf16) {
    if (B <= 0 || M <= 0 || cx <= 0 || T <= 0 || Tx < T || ntap < 1 || ntap > 8) return 0;
    if ((long)ntap * cx > 0x7fffffffL / 4) return 0;
    return use_bf16 ? (pw_taps_fit_bf16(B, M, cx, ntap, T, Tx) ? 1 : 0) : 1;
}

extern "C" int v100_pw_gemm_taps(const float* A, const void* A_bf16, const float* Xp, float* Y, const float* bias, const float* R,
                                 int B, int M, int cx, int T, int Tx, int ntap, const int* shifts, int use_bf16, void* stream) {
    if (!Xp || !Y) return V100_ERR_NULL;
    if (use_bf16 ? !A_bf16 : !A) return V100_ERR_NULL;
    if (use_bf16 < 0 || use_bf16 > 2) return V100_ERR_SHAPE;
    if (!v100_pw_taps_supported(B, M, cx, ntap, T, Tx, use_bf16)) return V100_ERR_SHAPE;
    if (use_bf16 == 2 && R) return V100_ERR_SHAPE;                 // fp16 = inference: store(+bias) only
    bool ok;
    const unsigned packed = pw_pack_shifts(shifts, ntap, T, Tx, ok);
    if (!ok) return V100_ERR_SHAPE;
    const int nmt = ceil_div(M, PW_BM), ntt = ceil_div(T, PW_BN);
    PwParams p{A, (const u16*)A_bf16, Xp, nullptr, nullptr, nullptr, nullptr, Y, bias, nullptr, nullptr, R, nullptr,
               B, M, ntap * cx, T, PW_X_NONE, R ? PW_EPI_ADD : PW_EPI_STORE, nmt, ntt, use_bf16, ntap, cx, Tx, packed};
    const long nwg = (long)nmt * ntt * B;
    if (nwg > 0x7fffffffL) return V100_ERR_SHAPE;
    dim3 grid((unsigned)nwg);
    hipStream_t st = (hipStream_t)stream;
    V100TimedRegion timed(V100_T_PW_GEMM, st);
    if (use_bf16) { if (!pw_launch_gemm_taps_bf16(p, grid, st)) return V100_ERR_SHAPE; }
    else pw_launch_gemm_taps_f32(p, grid, st);
    return v100_launch_status();
}

extern "C" int v100_pw_wgrad_taps(const float* G, int Tg, int g_off, const float* Xp, float* partial, float* dW, int S, int B, int M,
                                  int cx, int T, int Tx, int ntap, const int* shifts, int use_bf16, void* stream) {
    if (!G || !Xp || !partial || !dW) return V100_ERR_NULL;
    if (S <= 0 || (S > B && (S % B != 0 || S / B > 64)) || g_off < 0 || Tg < g_off + T || use_bf16 < 0 || use_bf16 > 1) return V100_ERR_SHAPE;
    if (!v100_pw_taps_supported(B, M, cx, ntap, T, Tx, use_bf16)) return V100_ERR_SHAPE;
    bool ok;
    const unsigned packed = pw_pack_shifts(shifts, ntap, T, Tx, ok);
    if (!ok) return V100_ERR_SHAPE;
    const int K = ntap * cx;
    const int nmt = ceil_div(M, PW_BM), nkt = ceil_div(K, PW_BN);
    WgParams p{G, nullptr, nullptr, nullptr, nullptr, Xp, nullptr, nullptr, partial, B, M, K, T, S, PW_X_NONE, PW_X_NONE, nmt, nkt,
               ntap, cx, Tx, Tg, g_off, packed};
    dim3 grid((unsigned)(nmt * nkt * S));
    hipStream_t st = (hipStream_t)stream;
    V100TimedRegion timed(V100_T_PW_WGRAD, st);
    if (use_bf16) { if (!pw_launch_wgrad_taps_bf16(p, grid, st)) return V100_ERR_SHAPE; }
    else pw_launch_wgrad_taps_f32(p, grid, st);
    const long n = (long)M * K;
    pw_slab_reduce(partial, dW, S, n, st);
    return v100_launch_status();
}
