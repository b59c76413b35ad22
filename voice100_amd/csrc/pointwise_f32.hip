// fp32 instantiations of the pointwise GEMM kernels (see pointwise_common.h / pointwise.hip).
#include "pointwise_common.h"

// =============================================================================================
// fp32 path
// =============================================================================================
#define F32_BK 16
#define F32_LD (128 + 4)

template <int XM_, int EPI_, bool TV, bool KV, bool TAPS = false>
__global__ __launch_bounds__(256) void pw_gemm_f32_kernel(PwParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][F32_BK][F32_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][F32_BK][F32_LD];
    __shared__ float red[2][2][64][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int b, tt, mt;
    pw_work(p, b, tt, mt);
    const int m0 = mt * PW_BM, t0 = tt * PW_BN;
    const int M = p.M, K = p.K, T = p.T;
    const int x_mode = PW_MODE(XM_, p.x_mode);
    const size_t xoff = TAPS ? (size_t)b * p.cx * p.Tx : (size_t)b * K * T;

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
            if constexpr (TAPS) {              // row k = tap * cx + c: row c of the padded X, read from column t + shift(tap)
                const int tap = kv ? k / p.cx : 0;
                rb[i] = ld4<true>(p.X, xoff + (size_t)(k - tap * p.cx) * p.Tx + pw_tap_shift(p.shifts, tap), t0 + b_t, T, kv);
            } else
            rb[i] = ld4<TV>(p.X, xoff + (size_t)k * T, t0 + b_t, T, kv);
            if (x_mode == PW_X_AFFINE2) rb2[i] = ld4<TV>(p.X2, xoff + (size_t)k * T, t0 + b_t, T, kv);
            if (x_mode != PW_X_NONE) { ca[i] = ldc(p.xa, k, kv, 1.f); cb[i] = ldc(p.xb, k, kv, 0.f); }
            if (x_mode == PW_X_AFFINE2) cc[i] = ldc(p.xc, k, kv, 0.f);
        }
    };
    auto store_tiles = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 av = mask4(ra[i], k0 + a_k, K, (m0 + a_m + 64 * i) < M);
#pragma unroll
            for (int e = 0; e < 4; ++e) As[buf][a_k + e][a_m + 64 * i] = av[e];
            const int k = k0 + b_k + 8 * i;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = (k < K && t0 + b_t + e < T) ? pw_x_transform(x_mode, rb[i][e], rb2[i][e], ca[i], cb[i], cc[i]) : 0.f;
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
        __builtin_amdgcn_sched_barrier(0);      // loads are issued before the MFMA block ...
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
        // ... and first USED after it: without this fence hipcc hoists the staging arithmetic (and the
        // vmcnt wait it needs) above the MFMAs, which exposes the whole memory latency every k-step.
        asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));   // accumulators stay in AGPRs
        asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]));
        if (x_mode == PW_X_AFFINE2) asm volatile("" : "+v"(rb2[0]), "+v"(rb2[1]));
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) store_tiles(cur ^ 1, (kt + 1) * F32_BK);
        __syncthreads();
    }
    pw_epilogue<EPI_>(p, acc, b, m0, t0, tt, wm, wn, lane, red);
}

// ---------------------------------------------------------------------------------------------
// Backward-weight, fp32.  Both operands are contraction(t)-contiguous in HBM; the tile loader
// transposes them into LDS as [t][row] so a lane's MFMA operand is a conflict-free ds_read_b32.
template <int GM_, int XM_, bool TV, bool TAPS = false>
__global__ __launch_bounds__(256) void pw_wgrad_f32_kernel(WgParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][F32_BK][F32_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][F32_BK][F32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int s, mt, ktile;
    wg_work(p, s, mt, ktile);
    const int m0 = mt * PW_BM, n0 = ktile * PW_BN;
    const int M = p.M, K = p.K, T = p.T;
    const int g_mode = PW_MODE(GM_, p.g_mode), x_mode = PW_MODE(XM_, p.x_mode);

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

    size_t xrow[2] = {0, 0};                // tap-addressed X: in-utterance offset of this thread's two rows (see WgParams)
    if constexpr (TAPS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = n0 + l_r + 64 * i, tap = kv[i] ? k / p.cx : 0;
            xrow[i] = (size_t)(k - tap * p.cx) * p.Tx + pw_tap_shift(p.shifts, tap);
        }
    }
    f32x4 ra[2], ra2[2], rb[2];
    auto load_tiles = [&](int b, int t0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + l_r + 64 * i, k = n0 + l_r + 64 * i;
            if constexpr (TAPS) {
                ra[i] = ld4<true>(p.G, ((size_t)b * M + m) * p.Tg + p.g_off, t0 + l_t, T, mv[i]);
                rb[i] = ld4<true>(p.X, (size_t)b * p.cx * p.Tx + xrow[i], t0 + l_t, T, kv[i]);
                continue;
            }
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
    const WgSpan sp = wg_span(p, s, nt);
    const int nsteps = sp.nb * sp.ntl, b_lo = sp.b_lo;
    const int lr = lane & 31, lk = lane >> 5;
    if (nsteps > 0) {
        load_tiles(b_lo, sp.t_first * F32_BK);
        store_tiles(0, sp.t_first * F32_BK);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const int nxt = st + 1;
        const int nb = b_lo + nxt / sp.ntl, ntt = (sp.t_first + nxt % sp.ntl) * F32_BK;
        if (nxt < nsteps) load_tiles(nb, ntt);
        __builtin_amdgcn_sched_barrier(0);
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
        // ... and first USED after it: without this fence hipcc hoists the staging arithmetic (and the
        // vmcnt wait it needs) above the MFMAs, which exposes the whole memory latency every k-step.
        asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));   // accumulators stay in AGPRs
        asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]));
        if (g_mode == PW_X_AFFINE2) asm volatile("" : "+v"(ra2[0]), "+v"(ra2[1]));
        __builtin_amdgcn_sched_barrier(0);
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
// Dispatch: fast kernels (modes fixed at compile time, aligned K) for the combinations the networks
// use; everything else goes to the generic instantiation (run-time modes, scalar-safe loads).
#define PW_NN_COMBOS(X) X(0, 0) X(0, 1) X(1, 1) X(0, 2) X(0, 3) X(0, 4) X(2, 0) X(2, 5)
#define PW_WG_COMBOS(X) X(0, 0) X(0, 1) X(2, 0)

void pw_launch_gemm_f32(const PwParams& p, dim3 grid, hipStream_t st) {
    const bool tv = (p.T & 3) == 0, kv = (p.K & 3) == 0;
    if (kv) {
#define X(XM, EP)                                                                                                   \
        if (p.x_mode == XM && p.epi_mode == EP) {                                                                   \
            if (tv) V100_GGL((pw_gemm_f32_kernel<XM, EP, true, true>), grid, dim3(256), 0, st, p);          \
            else V100_GGL((pw_gemm_f32_kernel<XM, EP, false, true>), grid, dim3(256), 0, st, p);            \
            return;                                                                                                 \
        }
        PW_NN_COMBOS(X)
#undef X
    }
    if (tv && kv) V100_GGL((pw_gemm_f32_kernel<-1, -1, true, true>), grid, dim3(256), 0, st, p);
    else V100_GGL((pw_gemm_f32_kernel<-1, -1, false, false>), grid, dim3(256), 0, st, p);
}

void pw_launch_wgrad_f32(const WgParams& p, dim3 grid, hipStream_t st) {
    const bool tv = (p.T & 3) == 0;
#define X(GM, XM)                                                                                                   \
    if (p.g_mode == GM && p.x_mode == XM) {                                                                         \
        if (tv) V100_GGL((pw_wgrad_f32_kernel<GM, XM, true>), grid, dim3(256), 0, st, p);                   \
        else V100_GGL((pw_wgrad_f32_kernel<GM, XM, false>), grid, dim3(256), 0, st, p);                     \
        return;                                                                                                     \
    }
    PW_WG_COMBOS(X)
#undef X
    if (tv) V100_GGL((pw_wgrad_f32_kernel<-1, -1, true>), grid, dim3(256), 0, st, p);
    else V100_GGL((pw_wgrad_f32_kernel<-1, -1, false>), grid, dim3(256), 0, st, p);
}


// Tap-addressed X operand (PwParams / WgParams): plain store (+bias) or +R epilogue, no prologues.
void pw_launch_gemm_taps_f32(const PwParams& p, dim3 grid, hipStream_t st) {
    V100_GGL((pw_gemm_f32_kernel<0, -1, true, true, true>), grid, dim3(256), 0, st, p);
}

void pw_launch_wgrad_taps_f32(const WgParams& p, dim3 grid, hipStream_t st) {
    V100_GGL((pw_wgrad_f32_kernel<0, 0, true, true>), grid, dim3(256), 0, st, p);
}
