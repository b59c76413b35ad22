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
#include "common.h"

enum { PW_X_NONE = 0, PW_X_AFFINE_RELU6 = 1, PW_X_AFFINE2 = 2 };
enum { PW_EPI_STORE = 0, PW_EPI_STATS = 1, PW_EPI_AFFINE_RELU6 = 2, PW_EPI_AFFINE_RES = 3, PW_EPI_MASK_STATS = 4, PW_EPI_ADD = 5 };

#define PW_BM 128
#define PW_BN 128

struct PwParams {
    const float* A;       // [M][K] fp32 weights
    const u16* Abf;       // [M][K] bf16 weights (bf16 path)
    const float* X;       // [B][K][T]
    const float* X2;      // [B][K][T]  (PW_X_AFFINE2)
    const float* xa; const float* xb; const float* xc;   // [K]
    float* Y;             // [B][M][T]
    const float* bias;    // [M] or null
    const float* ea; const float* eb;                     // [M]
    const float* R;       // [B][M][T] residual / pre-activation tensor
    float* stats;         // [B * n_ttiles][M][2]
    int B, M, K, T, x_mode, epi_mode, n_mtiles, n_ttiles;
};

__device__ __forceinline__ float pw_x_transform(int mode, float v, float v2, float a, float b, float c) {
    if (mode == PW_X_AFFINE_RELU6) return relu6f(fmaf(v, a, b));
    if (mode == PW_X_AFFINE2) return fmaf(v, a, fmaf(v2, b, c));
    return v;
}

__device__ __forceinline__ float half_wave_sum(float v) {   // sum over the 32 lanes sharing lane>>5
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Branch-free tile loads.  A conditional load inside a branch makes hipcc wait for it (vmcnt(0)) at
// the join, which serialises a tile's loads one memory latency after another; here every load is
// unconditional (address clamped to the tensor base when out of range) and the value is selected
// afterwards, so a tile's loads are all in flight together.
//   elements i0..i0+3 of the row starting at base + row_off (row length n), zero where invalid
template <bool VEC>
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ base, size_t row_off, int i0, int n, bool row_ok) {
    f32x4 v;
    if constexpr (VEC) {
        const bool ok = row_ok && i0 < n;
        v = *reinterpret_cast<const f32x4*>(base + (ok ? row_off + i0 : 0));
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = row_ok && (i0 + e) < n;
            const float x = base[ok ? row_off + i0 + e : 0];
            v[e] = ok ? x : 0.f;
        }
    }
    return v;
}

__device__ __forceinline__ float ldc(const float* __restrict__ c, int i, bool ok, float dflt) {   // coefficient, predicated
    const float v = c[ok ? i : 0];
    return ok ? v : dflt;
}

// XCD-aware work-item index: consecutive block ids are dealt round-robin over the 8 XCDs, so give
// each XCD a contiguous chunk of the work list (blocks that share an operand tile then share an L2).
// Bijective for any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Shared epilogue for the NN kernels. acc[i][j] is the 32x32 tile (i: m sub-tile, j: t sub-tile);
// element r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ void pw_epilogue(const PwParams& p, f32x16 (&acc)[2][2], int b, int m0, int t0, int tt, int wm, int wn,
                                            int lane, float (*red)[2][64][2]) {
    const int epi = p.epi_mode;
    const int col = lane & 31, half = lane >> 5;
    const bool do_stats = (epi == PW_EPI_STATS || epi == PW_EPI_MASK_STATS);
    const bool use_e = (epi == PW_EPI_AFFINE_RELU6 || epi == PW_EPI_AFFINE_RES || epi == PW_EPI_MASK_STATS);
    const bool use_r = (epi == PW_EPI_MASK_STATS || epi == PW_EPI_ADD || (epi == PW_EPI_AFFINE_RES && p.R));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;    // row inside the wave's 64
            const int m = m0 + wm * 64 + rl;
            const bool mv = m < p.M;
            const float ea = use_e ? ldc(p.ea, m, mv, 1.f) : 1.f;
            const float eb = use_e ? ldc(p.eb, m, mv, 0.f) : 0.f;
            const float bs = p.bias ? ldc(p.bias, m, mv, 0.f) : 0.f;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t = t0 + wn * 64 + j * 32 + col;
                const bool ok = mv && t < p.T;
                const size_t o = ((size_t)b * p.M + m) * p.T + t;
                float rv = 0.f;
                if (use_r) { rv = p.R[ok ? o : 0]; rv = ok ? rv : 0.f; }
                float v = acc[i][j][r] + bs;
                if (epi == PW_EPI_STATS) {
                    if (ok) { s0 += v; s1 = fmaf(v, v, s1); }
                } else if (epi == PW_EPI_AFFINE_RELU6) {
                    v = relu6f(fmaf(v, ea, eb));
                } else if (epi == PW_EPI_AFFINE_RES) {
                    v = fmaf(v, ea, eb) + rv;
                } else if (epi == PW_EPI_MASK_STATS) {
                    const float pre = fmaf(rv, ea, eb);
                    v = (pre > 0.f && pre < 6.f) ? v : 0.f;
                    if (ok) { s0 += v; s1 = fmaf(v, rv, s1); }
                } else if (epi == PW_EPI_ADD) {
                    v += rv;
                }
                if (ok) p.Y[o] = v;
            }
            if (do_stats) {
                s0 = half_wave_sum(s0);
                s1 = half_wave_sum(s1);
                if (col == 0) { red[wm][wn][rl][0] = s0; red[wm][wn][rl][1] = s1; }
            }
        }
    }
    if (do_stats) {
        __syncthreads();
        const int tid = threadIdx.x;
        if (tid < 128) {
            const int m = m0 + tid;
            if (m < p.M) {
                const int w = tid >> 6, rl = tid & 63;
                const size_t part = (size_t)b * p.n_ttiles + tt;
                p.stats[(part * p.M + m) * 2 + 0] = red[w][0][rl][0] + red[w][1][rl][0];
                p.stats[(part * p.M + m) * 2 + 1] = red[w][0][rl][1] + red[w][1][rl][1];
            }
        }
    }
}

// work item -> (b, t-tile, m-tile), m-tile fastest
__device__ __forceinline__ void pw_work(const PwParams& p, int& b, int& tt, int& mt) {
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    mt = w % p.n_mtiles;
    const int rest = w / p.n_mtiles;
    tt = rest % p.n_ttiles;
    b = rest / p.n_ttiles;
}

// =============================================================================================
// fp32 path
// =============================================================================================
#define F32_BK 16
#define F32_LD (128 + 4)

template <bool TV, bool KV>
__global__ __launch_bounds__(256) void pw_gemm_f32_kernel(PwParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][F32_BK][F32_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][F32_BK][F32_LD];
    __shared__ float red[2][2][64][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int b, tt, mt;
    pw_work(p, b, tt, mt);
    const int m0 = mt * PW_BM, t0 = tt * PW_BN;
    const int M = p.M, K = p.K, T = p.T, x_mode = p.x_mode;
    const size_t xoff = (size_t)b * K * T;

    const int a_k = (tid & 3) * 4;          // + k0, 4 consecutive k
    const int a_m = tid >> 2;               // + 64*i
    const int b_t = (tid & 31) * 4;         // + t0, 4 consecutive t
    const int b_k = tid >> 5;               // + 8*i

    f32x4 ra[2], rb[2], rb2[2];
    float ca[2], cb[2], cc[2];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + a_m + 64 * i;
            ra[i] = ld4<KV>(p.A, (size_t)m * K, k0 + a_k, K, m < M);
            const int k = k0 + b_k + 8 * i;
            const bool kv = k < K;
            rb[i] = ld4<TV>(p.X, xoff + (size_t)k * T, t0 + b_t, T, kv);
            if (x_mode == PW_X_AFFINE2) rb2[i] = ld4<TV>(p.X2, xoff + (size_t)k * T, t0 + b_t, T, kv);
            if (x_mode != PW_X_NONE) { ca[i] = ldc(p.xa, k, kv, 1.f); cb[i] = ldc(p.xb, k, kv, 0.f); }
            if (x_mode == PW_X_AFFINE2) cc[i] = ldc(p.xc, k, kv, 0.f);
        }
    };
    auto store_tiles = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) As[buf][a_k + e][a_m + 64 * i] = ra[i][e];
            const int k = k0 + b_k + 8 * i;
            f32x4 v = rb[i];
            if (x_mode != PW_X_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = (k < K && t0 + b_t + e < T) ? pw_x_transform(x_mode, rb[i][e], rb2[i][e], ca[i], cb[i], cc[i]) : 0.f;
            }
            *reinterpret_cast<f32x4*>(&Bs[buf][b_k + 8 * i][b_t]) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + F32_BK - 1) / F32_BK;
    load_tiles(0);
    store_tiles(0, 0);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * F32_BK);
#pragma unroll
        for (int kk = 0; kk < F32_BK; kk += 2) {
            const float a0 = As[cur][kk + lk][wm * 64 + lr];
            const float a1 = As[cur][kk + lk][wm * 64 + 32 + lr];
            const float b0 = Bs[cur][kk + lk][wn * 64 + lr];
            const float b1 = Bs[cur][kk + lk][wn * 64 + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1, (kt + 1) * F32_BK);
        __syncthreads();
    }
    pw_epilogue(p, acc, b, m0, t0, tt, wm, wn, lane, red);
}

// ---------------------------------------------------------------------------------------------
// Backward-weight, fp32.  Both operands are contraction(t)-contiguous in HBM; the tile loader
// transposes them into LDS as [t][row] so a lane's MFMA operand is a conflict-free ds_read_b32.
struct WgParams {
    const float* G;  const float* G2;  const float* ga; const float* gb; const float* gc;   // A operand [B][M][T], coeffs [M]
    const float* X;  const float* xa; const float* xb;                                        // B operand [B][K][T], coeffs [K]
    float* partial;  // [S][M][K]
    int B, M, K, T, S, g_mode, x_mode, n_mtiles, n_ktiles;
};

// work item -> (split, m-tile, k-tile), k-tile fastest: one split's tiles sit on one XCD
__device__ __forceinline__ void wg_work(const WgParams& p, int& s, int& mt, int& kt) {
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    kt = w % p.n_ktiles;
    const int rest = w / p.n_ktiles;
    mt = rest % p.n_mtiles;
    s = rest / p.n_mtiles;
}

template <bool TV>
__global__ __launch_bounds__(256) void pw_wgrad_f32_kernel(WgParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][F32_BK][F32_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][F32_BK][F32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int s, mt, ktile;
    wg_work(p, s, mt, ktile);
    const int m0 = mt * PW_BM, n0 = ktile * PW_BN;
    const int M = p.M, K = p.K, T = p.T, g_mode = p.g_mode, x_mode = p.x_mode;
    const int bper = (p.B + p.S - 1) / p.S;
    const int b_lo = s * bper, b_hi = min(p.B, b_lo + bper);

    const int l_t = (tid & 3) * 4;          // 4 consecutive t inside the 16-wide step
    const int l_r = tid >> 2;               // row (m or k), + 64*i

    // per-row prologue coefficients are fixed for the whole kernel
    float ga[2], gb[2], gc[2], xa[2], xb[2];
    bool mv[2], kv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + l_r + 64 * i, k = n0 + l_r + 64 * i;
        mv[i] = m < M; kv[i] = k < K;
        ga[i] = (g_mode != PW_X_NONE) ? ldc(p.ga, m, mv[i], 1.f) : 1.f;
        gb[i] = (g_mode != PW_X_NONE) ? ldc(p.gb, m, mv[i], 0.f) : 0.f;
        gc[i] = (g_mode == PW_X_AFFINE2) ? ldc(p.gc, m, mv[i], 0.f) : 0.f;
        xa[i] = (x_mode != PW_X_NONE) ? ldc(p.xa, k, kv[i], 1.f) : 1.f;
        xb[i] = (x_mode != PW_X_NONE) ? ldc(p.xb, k, kv[i], 0.f) : 0.f;
    }

    f32x4 ra[2], ra2[2], rb[2];
    auto load_tiles = [&](int b, int t0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + l_r + 64 * i, k = n0 + l_r + 64 * i;
            ra[i] = ld4<TV>(p.G, ((size_t)b * M + m) * T, t0 + l_t, T, mv[i]);
            if (g_mode == PW_X_AFFINE2) ra2[i] = ld4<TV>(p.G2, ((size_t)b * M + m) * T, t0 + l_t, T, mv[i]);
            rb[i] = ld4<TV>(p.X, ((size_t)b * K + k) * T, t0 + l_t, T, kv[i]);
        }
    };
    auto store_tiles = [&](int buf, int t0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool tv = t0 + l_t + e < T;
                As[buf][l_t + e][l_r + 64 * i] = (mv[i] && tv) ? pw_x_transform(g_mode, ra[i][e], ra2[i][e], ga[i], gb[i], gc[i]) : 0.f;
                Bs[buf][l_t + e][l_r + 64 * i] = (kv[i] && tv) ? pw_x_transform(x_mode, rb[i][e], 0.f, xa[i], xb[i], 0.f) : 0.f;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (T + F32_BK - 1) / F32_BK;
    const int nsteps = (b_hi > b_lo) ? (b_hi - b_lo) * nt : 0;
    const int lr = lane & 31, lk = lane >> 5;
    if (nsteps > 0) {
        load_tiles(b_lo, 0);
        store_tiles(0, 0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const int nxt = st + 1;
        const int nb = b_lo + nxt / nt, ntt = (nxt % nt) * F32_BK;
        if (nxt < nsteps) load_tiles(nb, ntt);
#pragma unroll
        for (int kk = 0; kk < F32_BK; kk += 2) {
            const float a0 = As[cur][kk + lk][wm * 64 + lr];
            const float a1 = As[cur][kk + lk][wm * 64 + 32 + lr];
            const float b0 = Bs[cur][kk + lk][wn * 64 + lr];
            const float b1 = Bs[cur][kk + lk][wn * 64 + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (nxt < nsteps) store_tiles(cur ^ 1, ntt);
        __syncthreads();
    }
    const int col = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int k = n0 + wn * 64 + j * 32 + col;
                if (m < M && k < K) p.partial[((size_t)s * M + m) * K + k] = acc[i][j][r];
            }
}

// =============================================================================================
// bf16 path: operands rounded to bf16 while staging, fp32 accumulate (v_mfma_f32_32x32x16_bf16).
// LDS images are [row][k] with k contiguous (64 bf16 = 128 B per row) and a 16-byte-chunk XOR
// swizzle chunk ^= (row >> 1) & 7 so the fragment ds_read_b128 of 32 consecutive rows is
// conflict-free (bank rule (a/4) % 64, 16-lane groups).
// =============================================================================================
#define BF_BK 64

__device__ __forceinline__ int bf_off(int row, int chunk) {          // byte offset inside a [128][64] bf16 tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// 8 consecutive bf16 of A[m][k..k+7] (zero where invalid), branch-free
template <bool KV>
__device__ __forceinline__ uint4 ld8bf(const u16* __restrict__ base, size_t row_off, int k, int K, bool row_ok) {
    uint4 v;
    if constexpr (KV) {
        const bool ok = row_ok && k < K;
        v = *reinterpret_cast<const uint4*>(base + (ok ? row_off + k : 0));
        if (!ok) v = uint4{0u, 0u, 0u, 0u};
    } else {
        unsigned t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = row_ok && (k + e) < K;
            const unsigned x = base[ok ? row_off + k + e : 0];
            t[e] = ok ? x : 0u;
        }
        v.x = t[0] | (t[1] << 16); v.y = t[2] | (t[3] << 16); v.z = t[4] | (t[5] << 16); v.w = t[6] | (t[7] << 16);
    }
    return v;
}

template <bool TV, bool KV>
__global__ __launch_bounds__(256) void pw_gemm_bf16_kernel(PwParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2][128 * 128];   // [m][k] bf16, 16 KB per buffer
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][128 * 128];   // [t][k] bf16
    __shared__ float red[2][2][64][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int b, tt, mt;
    pw_work(p, b, tt, mt);
    const int m0 = mt * PW_BM, t0 = tt * PW_BN;
    const int M = p.M, K = p.K, T = p.T, x_mode = p.x_mode;
    const size_t xoff = (size_t)b * K * T;

    // A tile: 128 rows x 8 chunks(8 bf16) = 1024 16-byte pieces, 4 per thread
    // B tile: 64 k x 128 t fp32; thread owns 8 consecutive k (one chunk) x 4 consecutive t
    const int b_tq = (tid & 31) * 4;       // t offset in tile
    const int b_kc = tid >> 5;             // chunk 0..7  -> k = 8*b_kc .. +7

    uint4 ra[4];
    f32x4 rb[8], rb2[8];
    float ca[8], cb[8], cc[8];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = tid + 256 * i;
            const int row = piece >> 3, ch = piece & 7;
            ra[i] = ld8bf<KV>(p.Abf, (size_t)(m0 + row) * K, k0 + ch * 8, K, (m0 + row) < M);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + b_kc * 8 + e;
            const bool kv = k < K;
            rb[e] = ld4<TV>(p.X, xoff + (size_t)k * T, t0 + b_tq, T, kv);
            if (x_mode == PW_X_AFFINE2) rb2[e] = ld4<TV>(p.X2, xoff + (size_t)k * T, t0 + b_tq, T, kv);
            if (x_mode != PW_X_NONE) { ca[e] = ldc(p.xa, k, kv, 1.f); cb[e] = ldc(p.xb, k, kv, 0.f); }
            if (x_mode == PW_X_AFFINE2) cc[e] = ldc(p.xc, k, kv, 0.f);
        }
    };
    auto store_tiles = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = tid + 256 * i;
            const int row = piece >> 3, ch = piece & 7;
            *reinterpret_cast<uint4*>(&As[buf][bf_off(row, ch)]) = ra[i];
        }
        float v[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + b_kc * 8 + e;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x = rb[e][q];
                if (x_mode != PW_X_NONE) x = (k < K && t0 + b_tq + q < T) ? pw_x_transform(x_mode, x, rb2[e][q], ca[e], cb[e], cc[e]) : 0.f;
                v[e][q] = x;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 o;
            o.x = pack_bf16(v[0][q], v[1][q]); o.y = pack_bf16(v[2][q], v[3][q]);
            o.z = pack_bf16(v[4][q], v[5][q]); o.w = pack_bf16(v[6][q], v[7][q]);
            *reinterpret_cast<uint4*>(&Bs[buf][bf_off(b_tq + q, b_kc)]) = o;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BF_BK - 1) / BF_BK;
    load_tiles(0);
    store_tiles(0, 0);
    __syncthreads();
    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BF_BK);
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {          // 16 k per MFMA: lane half lh holds k = 16*ks + 8*lh .. +7
            const int ch = ks * 2 + lh;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[cur][bf_off(wm * 64 + lr, ch)]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[cur][bf_off(wm * 64 + 32 + lr, ch)]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[cur][bf_off(wn * 64 + lr, ch)]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[cur][bf_off(wn * 64 + 32 + lr, ch)]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1, (kt + 1) * BF_BK);
        __syncthreads();
    }
    pw_epilogue(p, acc, b, m0, t0, tt, wm, wn, lane, red);
}

// Backward-weight, bf16: contraction index is t; both operands are read as 8 consecutive t
// (two float4), transformed, rounded and written as one 16-byte chunk of a [row][t] image.
template <bool TV>
__global__ __launch_bounds__(256) void pw_wgrad_bf16_kernel(WgParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2][128 * 128];   // [m][t] bf16
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][128 * 128];   // [k][t] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int s, mt, ktile;
    wg_work(p, s, mt, ktile);
    const int m0 = mt * PW_BM, n0 = ktile * PW_BN;
    const int M = p.M, K = p.K, T = p.T, g_mode = p.g_mode, x_mode = p.x_mode;
    const int bper = (p.B + p.S - 1) / p.S;
    const int b_lo = s * bper, b_hi = min(p.B, b_lo + bper);

    // 128 rows x 8 chunks per operand = 1024 pieces, 4 per thread: piece = tid + 256*i (row = piece>>3, chunk = piece&7)
    float ga[4], gb[4], gc[4], xa[4], xb[4];
    bool mv[4], kv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid + 256 * i) >> 3;
        const int m = m0 + row, k = n0 + row;
        mv[i] = m < M; kv[i] = k < K;
        ga[i] = (g_mode != PW_X_NONE) ? ldc(p.ga, m, mv[i], 1.f) : 1.f;
        gb[i] = (g_mode != PW_X_NONE) ? ldc(p.gb, m, mv[i], 0.f) : 0.f;
        gc[i] = (g_mode == PW_X_AFFINE2) ? ldc(p.gc, m, mv[i], 0.f) : 0.f;
        xa[i] = (x_mode != PW_X_NONE) ? ldc(p.xa, k, kv[i], 1.f) : 1.f;
        xb[i] = (x_mode != PW_X_NONE) ? ldc(p.xb, k, kv[i], 0.f) : 0.f;
    }

    f32x4 ra[4][2], ra2[4][2], rb[4][2];
    auto load_tiles = [&](int b, int t0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = tid + 256 * i;
            const int row = piece >> 3, ch = piece & 7;
            const int m = m0 + row, k = n0 + row, t = t0 + ch * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ra[i][h] = ld4<TV>(p.G, ((size_t)b * M + m) * T, t + 4 * h, T, mv[i]);
                if (g_mode == PW_X_AFFINE2) ra2[i][h] = ld4<TV>(p.G2, ((size_t)b * M + m) * T, t + 4 * h, T, mv[i]);
                rb[i][h] = ld4<TV>(p.X, ((size_t)b * K + k) * T, t + 4 * h, T, kv[i]);
            }
        }
    };
    auto store_tiles = [&](int buf, int t0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = tid + 256 * i;
            const int row = piece >> 3, ch = piece & 7;
            const int t = t0 + ch * 8;
            float va[8], vb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool tv = t + e < T;
                va[e] = (mv[i] && tv) ? pw_x_transform(g_mode, ra[i][e >> 2][e & 3], ra2[i][e >> 2][e & 3], ga[i], gb[i], gc[i]) : 0.f;
                vb[e] = (kv[i] && tv) ? pw_x_transform(x_mode, rb[i][e >> 2][e & 3], 0.f, xa[i], xb[i], 0.f) : 0.f;
            }
            uint4 oa, ob;
            oa.x = pack_bf16(va[0], va[1]); oa.y = pack_bf16(va[2], va[3]); oa.z = pack_bf16(va[4], va[5]); oa.w = pack_bf16(va[6], va[7]);
            ob.x = pack_bf16(vb[0], vb[1]); ob.y = pack_bf16(vb[2], vb[3]); ob.z = pack_bf16(vb[4], vb[5]); ob.w = pack_bf16(vb[6], vb[7]);
            *reinterpret_cast<uint4*>(&As[buf][bf_off(row, ch)]) = oa;
            *reinterpret_cast<uint4*>(&Bs[buf][bf_off(row, ch)]) = ob;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (T + BF_BK - 1) / BF_BK;
    const int nsteps = (b_hi > b_lo) ? (b_hi - b_lo) * nt : 0;
    const int lr = lane & 31, lh = lane >> 5;
    if (nsteps > 0) {
        load_tiles(b_lo, 0);
        store_tiles(0, 0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const int nxt = st + 1;
        const int nb = b_lo + nxt / nt, ntt = (nxt % nt) * BF_BK;
        if (nxt < nsteps) load_tiles(nb, ntt);
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {
            const int ch = ks * 2 + lh;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[cur][bf_off(wm * 64 + lr, ch)]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[cur][bf_off(wm * 64 + 32 + lr, ch)]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[cur][bf_off(wn * 64 + lr, ch)]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[cur][bf_off(wn * 64 + 32 + lr, ch)]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (nxt < nsteps) store_tiles(cur ^ 1, ntt);
        __syncthreads();
    }
    const int col = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int k = n0 + wn * 64 + j * 32 + col;
                if (m < M && k < K) p.partial[((size_t)s * M + m) * K + k] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------------------------
__global__ void pw_slab_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int S, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int g = 0; g < S; ++g) s += partial[(size_t)g * n + i];
    out[i] = s;
}

// fp32 [rows][cols] -> bf16 [rows][cols] and/or transposed copies (weights are tiny: <= 1M elements)
__global__ void weight_prep_kernel(const float* __restrict__ w, int rows, int cols, u16* __restrict__ w_bf,
                                   float* __restrict__ wt, u16* __restrict__ wt_bf) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float v = w[i];
    if (w_bf) w_bf[i] = f2bf(v);
    if (wt) wt[(size_t)c * rows + r] = v;
    if (wt_bf) wt_bf[(size_t)c * rows + r] = f2bf(v);
}

extern "C" int v100_pw_num_parts(int B, int T) { return B * ceil_div(T, PW_BN); }

extern "C" int v100_pw_wgrad_splits(int B, int M, int K) {
    // enough (tile, split) workgroups to fill the chip, capped by the batch
    const int tiles = ceil_div(M, PW_BM) * ceil_div(K, PW_BN);
    int S = ceil_div(512, tiles);
    if (S > B) S = B;
    if (S < 1) S = 1;
    return S;
}

extern "C" int v100_weight_prep(const float* w, int rows, int cols, void* w_bf, float* wt, void* wt_bf, void* stream) {
    if (!w) return V100_ERR_NULL;
    if (rows <= 0 || cols <= 0) return V100_ERR_SHAPE;
    const long n = (long)rows * cols;
    hipLaunchKernelGGL(weight_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, rows, cols,
                       (u16*)w_bf, wt, (u16*)wt_bf);
    return v100_launch_status();
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
    PwParams p{A, (const u16*)A_bf16, X, X2, xa, xb, xc, Y, bias, ea, eb, R, stats, B, M, K, T, x_mode, epi_mode, nmt, ntt};
    const long nwg = (long)nmt * ntt * B;
    if (nwg > 0x7fffffffL) return V100_ERR_SHAPE;
    dim3 grid((unsigned)nwg);
    hipStream_t st = (hipStream_t)stream;
    const bool tv = (T & 3) == 0;
    if (use_bf16) {
        const bool kv = (K & 7) == 0;
        if (tv && kv) hipLaunchKernelGGL((pw_gemm_bf16_kernel<true, true>), grid, dim3(256), 0, st, p);
        else if (tv) hipLaunchKernelGGL((pw_gemm_bf16_kernel<true, false>), grid, dim3(256), 0, st, p);
        else if (kv) hipLaunchKernelGGL((pw_gemm_bf16_kernel<false, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pw_gemm_bf16_kernel<false, false>), grid, dim3(256), 0, st, p);
    } else {
        const bool kv = (K & 3) == 0;
        if (tv && kv) hipLaunchKernelGGL((pw_gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, p);
        else if (tv) hipLaunchKernelGGL((pw_gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, p);
        else if (kv) hipLaunchKernelGGL((pw_gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pw_gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, p);
    }
    return v100_launch_status();
}

extern "C" int v100_pw_wgrad(const float* G, const float* G2, const float* ga, const float* gb, const float* gc, int g_mode,
                             const float* X, const float* xa, const float* xb, int x_mode, float* partial, float* dW,
                             int S, int B, int M, int K, int T, int use_bf16, void* stream) {
    if (!G || !X || !partial || !dW) return V100_ERR_NULL;
    if (B <= 0 || M <= 0 || K <= 0 || T <= 0 || S <= 0 || S > B) return V100_ERR_SHAPE;
    if (g_mode < 0 || g_mode > 2 || x_mode < 0 || x_mode > 1) return V100_ERR_SHAPE;
    if (g_mode != PW_X_NONE && (!ga || !gb)) return V100_ERR_NULL;
    if (g_mode == PW_X_AFFINE2 && (!G2 || !gc)) return V100_ERR_NULL;
    if (x_mode != PW_X_NONE && (!xa || !xb)) return V100_ERR_NULL;
    const int nmt = ceil_div(M, PW_BM), nkt = ceil_div(K, PW_BN);
    WgParams p{G, G2, ga, gb, gc, X, xa, xb, partial, B, M, K, T, S, g_mode, x_mode, nmt, nkt};
    dim3 grid((unsigned)(nmt * nkt * S));
    hipStream_t st = (hipStream_t)stream;
    const bool tv = (T & 3) == 0;
    if (use_bf16) {
        if (tv) hipLaunchKernelGGL((pw_wgrad_bf16_kernel<true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pw_wgrad_bf16_kernel<false>), grid, dim3(256), 0, st, p);
    } else {
        if (tv) hipLaunchKernelGGL((pw_wgrad_f32_kernel<true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pw_wgrad_f32_kernel<false>), grid, dim3(256), 0, st, p);
    }
    const long n = (long)M * K;
    hipLaunchKernelGGL(pw_slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, dW, S, n);
    return v100_launch_status();
}
