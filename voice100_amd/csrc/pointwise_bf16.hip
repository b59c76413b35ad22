// bf16-operand instantiations of the pointwise GEMM kernels (see pointwise_common.h / pointwise.hip).
#include "pointwise_common.h"
#include <type_traits>

// =============================================================================================
// bf16 path: operands rounded to bf16 while staging, fp32 accumulate (v_mfma_f32_32x32x16_bf16).
// LDS images are [row][k] with k contiguous (64 bf16 = 128 B per row) and a 16-byte-chunk XOR
// swizzle chunk ^= (row >> 1) & 7 so the fragment ds_read_b128 of 32 consecutive rows is
// conflict-free (bank rule (a/4) % 64; the hardware's four 16-lane groups each see every slot of both row
// parities once).  (Also spreading the X-patch STORES, whose lanes hold rows 4 apart, with an extra
// ^ ((row >> 4) & 1) and a lane remap was measured: no gain, so the simpler form stays.)
// =============================================================================================
#define BF_BK 64

__device__ __forceinline__ int bf_off(int row, int chunk) {          // byte offset inside a [128][64] bf16 tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// 8 consecutive bf16 of A[m][k..k+7], RAW (address clamped when out of range; mask8bf at the use)
template <bool KV>
__device__ __forceinline__ uint4 ld8bf(const u16* __restrict__ base, size_t row_off, int k, int K, bool row_ok) {
    uint4 v;
    if constexpr (KV) {
        const bool ok = row_ok && k < K;
        v = *reinterpret_cast<const uint4*>(base + (ok ? row_off + k : 0));
    } else {
        unsigned t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = row_ok && (k + e) < K;
            t[e] = base[ok ? row_off + k + e : 0];
        }
        v.x = t[0] | (t[1] << 16); v.y = t[2] | (t[3] << 16); v.z = t[4] | (t[5] << 16); v.w = t[6] | (t[7] << 16);
    }
    return v;
}

__device__ __forceinline__ uint4 mask8bf(uint4 v, int k, int K, bool row_ok) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned lo = (row_ok && (k + 2 * e) < K) ? 0xffffu : 0u;
        const unsigned hi = (row_ok && (k + 2 * e + 1) < K) ? 0xffff0000u : 0u;
        w[e] &= (lo | hi);
    }
    return uint4{w[0], w[1], w[2], w[3]};
}

template <int XM_, int EPI_, bool TV, bool KV, bool F16 = false>
__global__ __launch_bounds__(256) void pw_gemm_bf16_kernel(PwParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2][128 * 128];   // [m][k] bf16, 16 KB per buffer
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][128 * 128];   // [t][k] bf16
    __shared__ float red[2][2][64][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int b, tt, mt;
    pw_work(p, b, tt, mt);
    const int m0 = mt * PW_BM, t0 = tt * PW_BN;
    const int M = p.M, K = p.K, T = p.T;
    const int x_mode = PW_MODE(XM_, p.x_mode);
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
            *reinterpret_cast<uint4*>(&As[buf][bf_off(row, ch)]) = mask8bf(ra[i], k0 + ch * 8, K, (m0 + row) < M);
        }
        float v[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + b_kc * 8 + e;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[e][q] = (k < K && t0 + b_tq + q < T) ? pw_x_transform(x_mode, rb[e][q], rb2[e][q], ca[e], cb[e], cc[e]) : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 o;
            o.x = pack16<F16>(v[0][q], v[1][q]); o.y = pack16<F16>(v[2][q], v[3][q]);
            o.z = pack16<F16>(v[4][q], v[5][q]); o.w = pack16<F16>(v[6][q], v[7][q]);
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
        __builtin_amdgcn_sched_barrier(0);      // loads are issued before the MFMA block ...
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {          // 16 k per MFMA: lane half lh holds k = 16*ks + 8*lh .. +7
            const int ch = ks * 2 + lh;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[cur][bf_off(wm * 64 + lr, ch)]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[cur][bf_off(wm * 64 + 32 + lr, ch)]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[cur][bf_off(wn * 64 + lr, ch)]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[cur][bf_off(wn * 64 + 32 + lr, ch)]);
            acc[0][0] = mfma16<F16>(a0, b0, acc[0][0]);
            acc[0][1] = mfma16<F16>(a0, b1, acc[0][1]);
            acc[1][0] = mfma16<F16>(a1, b0, acc[1][0]);
            acc[1][1] = mfma16<F16>(a1, b1, acc[1][1]);
        }
        // ... and first USED after it: without this fence hipcc hoists the staging arithmetic (and the
        // vmcnt wait it needs) above the MFMAs, which exposes the whole memory latency every k-step.
        asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));   // accumulators stay in AGPRs
        asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7]));
        if (x_mode == PW_X_AFFINE2)
            asm volatile("" : "+v"(rb2[0]), "+v"(rb2[1]), "+v"(rb2[2]), "+v"(rb2[3]), "+v"(rb2[4]), "+v"(rb2[5]), "+v"(rb2[6]), "+v"(rb2[7]));
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) store_tiles(cur ^ 1, (kt + 1) * BF_BK);
        __syncthreads();
    }
    pw_epilogue<EPI_>(p, acc, b, m0, t0, tt, wm, wn, lane, red);
}

// ---------------------------------------------------------------------------------------------
// Fast path of the NN kernel for full tiles (K % 64 == 0, T % 128 == 0 -- every layer of the
// reference networks at the benchmark shapes).  Same tiling and LDS images as above, but every
// global access is a buffer load: the descriptors are wave-uniform, the per-lane byte offsets are
// computed once, the k-step advance is a scalar offset, and rows past M fall outside the
// descriptor and read as zero in hardware -- no per-load address arithmetic, no masks.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int XM, int EPI, int BM, bool F16 = false, bool TAPS = false, int IO = 0, bool PERSIST = false>
__global__ __launch_bounds__(BM * 2) void pw_gemm_bf16_fast_kernel(PwParams p) {
    static_assert(!TAPS || XM == PW_X_NONE, "tap-addressed X has no prologue");
    static_assert(!(IO != 0 && TAPS), "16-bit activation storage: no tap-addressed form");
    static_assert(((IO & PW_IO_F16) != 0) == (F16 && IO != 0), "fp16-stored tensors go with fp16 operands (PW_IO_F16), bf16-stored ones with bf16");
    static_assert(!(F16 && IO != 0 && XM != PW_X_NONE), "fp16 storage: plain X operand only (inference)");
    // PERSIST: the grid is a divisor of the tile count and a workgroup walks tiles v = blockIdx.x, + gridDim.x, ...; the first two
    // k-tiles of the NEXT tile are requested into the (idle) staging registers before the epilogue of the current one, so a
    // tile's start does not wait a memory latency (2.5 us of the ~15 us a 256 x 128 x 512 tile takes) and the epilogue's
    // stores overlap the next tile's loads.  For the short-K GEMMs (several tiles per CU); XM == NONE only (register budget).
    static_assert(!PERSIST || (XM == PW_X_NONE && !TAPS), "persistent form: plain X operand");
    constexpr bool XB = (IO & PW_IO_X) != 0, X2B = (IO & PW_IO_X2) != 0;    // operand tensors stored as bf16 (pitched rows)
    using XReg = std::conditional_t<XB, u32x2, u32x4>;
    using X2Reg = std::conditional_t<X2B, u32x2, u32x4>;
    // BM x 128 block tile, BM/64 x 2 waves of 64x64.  BM = 256 (8 waves, one block per CU) halves the L2 traffic of
    // the X operand, which is what bounds these GEMMs (each X tile is re-read by every M-tile); BM = 128 for M <= 128.
    constexpr int NT = BM * 2;                      // threads
    constexpr int KPT = 2048 / NT;                  // k rows per thread in the X patch: 8 (256 threads) or 4 (512)
    constexpr int A_BYTES = BM * 128;               // one A stage: [BM][64] bf16
    constexpr int SMEM = (BM * 128 * 4 > 2 * A_BYTES + 2 * 128 * 128) ? BM * 128 * 4 : 2 * A_BYTES + 2 * 128 * 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];      // stages, reused by the epilogue as [BM][128] fp32
    unsigned char* As = smem;                       // [2][BM][64] bf16
    unsigned char* Bs = smem + 2 * A_BYTES;         // [2][128][64] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int b, tt, mt;
    int vtile = blockIdx.x;
    const int ntiles_all = PERSIST ? p.n_mtiles * p.n_ttiles * p.B : 0;
    if constexpr (PERSIST) pw_work_v(p, vtile, ntiles_all, b, tt, mt);
    else pw_work(p, b, tt, mt);
    int m0 = mt * BM, t0 = tt * PW_BN;
    const int M = p.M, K = p.K, T = p.T;
    // tap-addressed X: physical rows are the cx channels of the padded tensor, row pitch Tx (see PwParams)
    const int Tx = TAPS ? p.Tx : T;
    const int Kx = TAPS ? p.cx : K;
    const int P16 = pw_pitch16(T, p.B);                  // row pitch of the bf16-stored tensors
    const int TxX = XB ? P16 : Tx, TxX2 = X2B ? P16 : Tx;
    constexpr int EX = XB ? 2 : 4, EX2 = X2B ? 2 : 4;      // bytes per element

    const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.Abf, (PW_ABLATE & 2) ? 0u : (unsigned)M * K * 2u);
    __amdgpu_buffer_rsrc_t rX = make_rsrc(reinterpret_cast<const char*>(p.X) + (size_t)b * Kx * TxX * EX,
                                          (PW_ABLATE & 1) ? 0u : (unsigned)Kx * TxX * EX);
    const __amdgpu_buffer_rsrc_t rX2 = make_rsrc(reinterpret_cast<const char*>(XM == PW_X_AFFINE2 ? p.X2 : p.X) +
                                                     (size_t)b * Kx * (XM == PW_X_AFFINE2 ? TxX2 * EX2 : TxX * EX),
                                                 (PW_ABLATE & 1) ? 0u : (unsigned)Kx * (XM == PW_X_AFFINE2 ? TxX2 * EX2 : TxX * EX));
    const __amdgpu_buffer_rsrc_t rCa = make_rsrc(XM != PW_X_NONE ? p.xa : p.X, (unsigned)K * 4u);
    const __amdgpu_buffer_rsrc_t rCb = make_rsrc(XM != PW_X_NONE ? p.xb : p.X, (unsigned)K * 4u);
    const __amdgpu_buffer_rsrc_t rCc = make_rsrc(XM == PW_X_AFFINE2 ? p.xc : p.X, (unsigned)K * 4u);

    const int b_tq = (tid & 31) * 4;               // t offset in tile
    const int b_kg = tid >> 5;                     // k group: rows KPT*b_kg .. +KPT-1
    int voA[4], ldsA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = tid + NT * i;
        const int row = piece >> 3, ch = piece & 7;
        voA[i] = ((m0 + row) * K + ch * 8) * 2;
        ldsA[i] = bf_off(row, ch);
    }
    int voX[KPT], voX2[XM == PW_X_AFFINE2 ? KPT : 1];
#pragma unroll
    for (int e = 0; e < KPT; ++e) {
        voX[e] = ((KPT * b_kg + e) * TxX + t0 + b_tq) * EX;
        if constexpr (XM == PW_X_AFFINE2) voX2[e] = ((KPT * b_kg + e) * TxX2 + t0 + b_tq) * EX2;
    }
    const int voC = KPT * b_kg * 4;
    int ldsB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ldsB[q] = bf_off(b_tq + q, (KPT * b_kg) >> 3) + ((KPT * b_kg) & 7) * 2;
    // PERSIST: re-aim the X descriptor and the per-lane offsets at tile v
    auto retarget = [&](int v) {
        pw_work_v(p, v, ntiles_all, b, tt, mt);
        m0 = mt * BM; t0 = tt * PW_BN;
        rX = make_rsrc(reinterpret_cast<const char*>(p.X) + (size_t)b * Kx * TxX * EX, (PW_ABLATE & 1) ? 0u : (unsigned)Kx * TxX * EX);
#pragma unroll
        for (int i = 0; i < 4; ++i) voA[i] = ((m0 + ((tid + NT * i) >> 3)) * K + ((tid + NT * i) & 7) * 8) * 2;
#pragma unroll
        for (int e = 0; e < KPT; ++e) voX[e] = ((KPT * b_kg + e) * TxX + t0 + b_tq) * EX;
    };

    // NST register stages of global loads in flight: with two, the loads of tile k+2 are issued
    // before the MFMA block of tile k and first used during the MFMA block of tile k+1.  The two-tensor prologue
    // (XM == AFFINE2) keeps one stage at BM = 128 (register budget); at BM = 256 its patch is half as large.
    constexpr int NST = (XM == PW_X_AFFINE2 && KPT == 8) ? 1 : 2;
    constexpr int NC = KPT / 4;                     // float4 coefficient loads per array
    u32x4 ra[NST][4], rca[NC], rcb[NC], rcc[XM == PW_X_AFFINE2 ? NC : 1];
    XReg rb[NST][KPT];
    X2Reg rb2[NST][XM == PW_X_AFFINE2 ? KPT : 1];
    auto load_tiles = [&](int k0, auto stg) {
        constexpr int SG = decltype(stg)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[SG][i] = __builtin_amdgcn_raw_buffer_load_b128(rA, voA[i], k0 * 2, 0);
        int so = k0 * TxX * EX;
        if constexpr (TAPS) {                  // a k-tile never straddles two taps (cx % 64 == 0, checked by the launcher)
            const int tap = k0 / p.cx;
            so = ((k0 - tap * p.cx) * Tx + pw_tap_shift(p.shifts, tap)) * 4;
        }
#pragma unroll
        for (int e = 0; e < KPT; ++e) {
            if constexpr (XB) rb[SG][e] = __builtin_amdgcn_raw_buffer_load_b64(rX, voX[e], so, 0);
            else rb[SG][e] = __builtin_amdgcn_raw_buffer_load_b128(rX, voX[e], so, 0);
            if constexpr (XM == PW_X_AFFINE2) {
                if constexpr (X2B) rb2[SG][e] = __builtin_amdgcn_raw_buffer_load_b64(rX2, voX2[e], k0 * TxX2 * EX2, 0);
                else rb2[SG][e] = __builtin_amdgcn_raw_buffer_load_b128(rX2, voX2[e], k0 * TxX2 * EX2, 0);
            }
        }
    };
    // BN coefficients of the tile that is about to be STORED: tiny, L2-resident, single register stage
    auto load_coefs = [&](int k0) {
        if constexpr (XM != PW_X_NONE) {
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                rca[h] = __builtin_amdgcn_raw_buffer_load_b128(rCa, voC + 16 * h, k0 * 4, 0);
                rcb[h] = __builtin_amdgcn_raw_buffer_load_b128(rCb, voC + 16 * h, k0 * 4, 0);
                if constexpr (XM == PW_X_AFFINE2) rcc[h] = __builtin_amdgcn_raw_buffer_load_b128(rCc, voC + 16 * h, k0 * 4, 0);
            }
        }
    };
    // one quarter of a tile store: A piece `q` and t-column `q` of this thread's X patch
    auto store_slice = [&](int buf, auto stg, auto slc) {
        constexpr int SG = decltype(stg)::value;
        constexpr int q = decltype(slc)::value;
        if constexpr (PW_ABLATE & 32) return;          // timing-only: no transform / LDS stores
        *reinterpret_cast<u32x4*>(As + buf * A_BYTES + ldsA[q]) = ra[SG][q];
        if constexpr (XM == PW_X_NONE && XB) {
            // plain 16-bit X (bf16, or fp16 under F16: the stored format IS the operand format): the [k][t] -> [t][k] transposition is a byte shuffle of the loaded words (column q of rows 2j, 2j + 1
            // -> word j), one v_perm_b32 per two elements instead of unpack + unpack + v_cvt_pk
            constexpr unsigned sel = (q & 1) ? 0x07060302u : 0x05040100u;
            unsigned char* dst = Bs + buf * (128 * 128) + ldsB[q];
            if constexpr (KPT == 8) {
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __builtin_amdgcn_perm(rb[SG][2 * j + 1][q >> 1], rb[SG][2 * j][q >> 1], sel);
                *reinterpret_cast<u32x4*>(dst) = o;
            } else {
                uint2 o;
                o.x = __builtin_amdgcn_perm(rb[SG][1][q >> 1], rb[SG][0][q >> 1], sel);
                o.y = __builtin_amdgcn_perm(rb[SG][3][q >> 1], rb[SG][2][q >> 1], sel);
                *reinterpret_cast<uint2*>(dst) = o;
            }
            return;
        }
        float v[KPT];
#pragma unroll
        for (int e = 0; e < KPT; ++e) {
            float x;
            if constexpr (XB) x = pw_bf16_at(rb[SG][e], q);
            else x = __builtin_bit_cast(f32x4, rb[SG][e])[q];
            if constexpr (XM == PW_X_NONE) v[e] = x;
            else {
                const float ca = __builtin_bit_cast(f32x4, rca[e >> 2])[e & 3];
                const float cb = __builtin_bit_cast(f32x4, rcb[e >> 2])[e & 3];
                if constexpr (XM == PW_X_AFFINE_RELU6) v[e] = relu6f(fmaf(x, ca, cb));
                else {
                    float x2;
                    if constexpr (X2B) x2 = pw_bf16_at(rb2[SG][e], q);
                    else x2 = __builtin_bit_cast(f32x4, rb2[SG][e])[q];
                    v[e] = fmaf(x, ca, fmaf(x2, cb, __builtin_bit_cast(f32x4, rcc[e >> 2])[e & 3]));
                }
            }
        }
        unsigned char* dst = Bs + buf * (128 * 128) + ldsB[q];
        if constexpr (KPT == 8) {
            u32x4 o;
            o[0] = pack16<F16>(v[0], v[1]); o[1] = pack16<F16>(v[2], v[3]); o[2] = pack16<F16>(v[4], v[5]); o[3] = pack16<F16>(v[6], v[7]);
            *reinterpret_cast<u32x4*>(dst) = o;
        } else {
            uint2 o;
            o.x = pack16<F16>(v[0], v[1]); o.y = pack16<F16>(v[2], v[3]);
            *reinterpret_cast<uint2*>(dst) = o;
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NST - 1>;
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
    auto store_tiles = [&](int buf, auto stg) {
        store_slice(buf, stg, Q0{}); store_slice(buf, stg, Q1{}); store_slice(buf, stg, Q2{}); store_slice(buf, stg, Q3{});
    };

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    const int nk = (PW_ABLATE & 16) ? 0 : (K + BF_BK - 1) / BF_BK;      // bit 4 (timing-only): no main loop
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;                       // fragment rows are lr (+32, +64..): same swizzle key
    const int rdA0 = (wm * 64 + lr) * 128, rdB0 = (wn * 64 + lr) * 128;
    auto mfma_step = [&](int cur, int ks) {
        const int co = ((ks * 2 + lh) ^ sw) << 4;
        const unsigned char* Ab = As + cur * A_BYTES;
        const unsigned char* Bb = Bs + cur * (128 * 128);
        bf16x8 a0, a1, b0, b1;
        if constexpr (PW_ABLATE & 64) {                  // timing-only: fragments without LDS reads
            a0 = a1 = b0 = b1 = (bf16x8){(short)(cur + 1), (short)ks, 3, 4, 5, 6, 7, 8};
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
        } else {
            a0 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + co);
            a1 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + 32 * 128 + co);
            b0 = *reinterpret_cast<const bf16x8*>(Bb + rdB0 + co);
            b1 = *reinterpret_cast<const bf16x8*>(Bb + rdB0 + 32 * 128 + co);
        }
        acc[0][0] = mfma16<F16>(a0, b0, acc[0][0]);
        acc[0][1] = mfma16<F16>(a0, b1, acc[0][1]);
        acc[1][0] = mfma16<F16>(a1, b0, acc[1][0]);
        acc[1][1] = mfma16<F16>(a1, b1, acc[1][1]);
    };
    auto mfma_block = [&](int cur) {
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) mfma_step(cur, ks);
    };
    // pin: the registers of stage SG are first USED after this point.
    // (a macro, not a lambda: clang rejects captured arrays as inline-asm operands inside a generic lambda)
#define PW_LOOP_SYNC() do { if constexpr (!(PW_ABLATE & 128)) __syncthreads(); } while (0)   /* bit 7: timing-only */
#define PW_PIN(SG)                                                                                            \
    do {                                                                                                      \
        _Pragma("unroll") for (int e_ = 0; e_ < KPT; ++e_) {                                                  \
            asm volatile("" : "+v"(rb[SG][e_]));                                                              \
            if constexpr (XM == PW_X_AFFINE2) asm volatile("" : "+v"(rb2[SG][e_]));                           \
        }                                                                                                     \
        asm volatile("" : "+v"(ra[SG][0]), "+v"(ra[SG][1]), "+v"(ra[SG][2]), "+v"(ra[SG][3]));                \
    } while (0)
    if constexpr (PERSIST) {               // first tile: k-tiles 0 and 1 (later tiles: requested before the previous epilogue)
        load_tiles(0, S0{});
        if (nk > 1) load_tiles(BF_BK, S1{});
    }
    for (;;) {
    if constexpr (NST == 1) {
        load_tiles(0, S0{});
        load_coefs(0);
        store_tiles(0, S0{});
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) { load_tiles((kt + 1) * BF_BK, S0{}); load_coefs((kt + 1) * BF_BK); }
            __builtin_amdgcn_sched_barrier(0);      // loads are issued before the MFMA block ...
            mfma_block(cur);
            // ... and first USED after it: without the fence hipcc hoists the staging arithmetic (and the vmcnt wait it
            // needs) above the MFMAs, which exposes the whole memory latency every k-step
            asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));
            PW_PIN(0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) store_tiles(cur ^ 1, S0{});
            __syncthreads();
        }
    } else {
        // tile t lives in register stage t&1 and LDS buffer t&1
        if constexpr (!PERSIST) {
            load_tiles(0, S0{});
            load_coefs(0);
            if (nk > 1) load_tiles(BF_BK, S1{});
        }
        store_tiles(0, S0{});
        __syncthreads();
        // Steady state: the tile to be stored was loaded a whole iteration ago, so its transform + LDS writes are
        // interleaved with the MFMAs of the current tile (matrix pipe and VALU/LDS overlap inside one wave).
        int kt = 0;
        // Main loop: both prefetches are unconditional.  (A conditional load makes hipcc's waitcnt insertion assume
        // the not-taken count at the join, so the wait for the OLDER stage degenerates into a wait for the prefetch
        // just issued -- the whole point of the second register stage.)
        for (; kt + 3 < nk; kt += 2) {
            // even tile kt: compute LDS 0; stage 1 holds tile kt+1; stage 0 is free -> tile kt+2.
            // vmcnt retires in order: the coefficient loads needed first are issued BEFORE the tile prefetch
            load_coefs((kt + 1) * BF_BK);
            __builtin_amdgcn_sched_barrier(0);
            load_tiles((kt + 2) * BF_BK, S0{});
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(0, 0); mfma_step(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            PW_PIN(NST - 1);
            store_slice(1, S1{}, Q0{}); mfma_step(0, 2);
            store_slice(1, S1{}, Q1{}); mfma_step(0, 3);
            store_slice(1, S1{}, Q2{}); store_slice(1, S1{}, Q3{});
            PW_LOOP_SYNC();
            // odd tile kt+1: compute LDS 1; stage 0 holds tile kt+2; stage 1 is free -> tile kt+3
            load_coefs((kt + 2) * BF_BK);
            __builtin_amdgcn_sched_barrier(0);
            load_tiles((kt + 3) * BF_BK, S1{});
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(1, 0); mfma_step(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            PW_PIN(0);
            store_slice(0, S0{}, Q0{}); mfma_step(1, 2);
            store_slice(0, S0{}, Q1{}); mfma_step(1, 3);
            store_slice(0, S0{}, Q2{}); store_slice(0, S0{}, Q3{});
            PW_LOOP_SYNC();
        }
        // Tail: the last two or three tiles (at most one pass), prefetches guarded
        for (; kt + 1 < nk; kt += 2) {
            const bool more = kt + 2 < nk;          // wave-uniform; MFMAs stay outside the branches (one accumulator chain)
            load_coefs((kt + 1) * BF_BK);
            __builtin_amdgcn_sched_barrier(0);
            if (more) load_tiles((kt + 2) * BF_BK, S0{});
            __builtin_amdgcn_sched_barrier(0);
            PW_PIN(NST - 1);
            mfma_step(0, 0); mfma_step(0, 1);
            store_slice(1, S1{}, Q0{}); mfma_step(0, 2);
            store_slice(1, S1{}, Q1{}); mfma_step(0, 3);
            store_slice(1, S1{}, Q2{}); store_slice(1, S1{}, Q3{});
            __syncthreads();
            if (more) load_coefs((kt + 2) * BF_BK);
            __builtin_amdgcn_sched_barrier(0);
            if (more) PW_PIN(0);
            mfma_step(1, 0); mfma_step(1, 1);
            if (more) store_slice(0, S0{}, Q0{});
            mfma_step(1, 2);
            if (more) store_slice(0, S0{}, Q1{});
            mfma_step(1, 3);
            if (more) { store_slice(0, S0{}, Q2{}); store_slice(0, S0{}, Q3{}); }
            __syncthreads();
        }
        if (kt < nk) {                         // odd tile count: the last tile already sits in LDS 0
            mfma_block(0);
            __syncthreads();                   // the epilogue reuses the stage buffers: every wave must be done reading
        }
    }
#undef PW_PIN
#undef PW_LOOP_SYNC
    if constexpr (PW_ABLATE & 8) {            // timing-only: no epilogue (keep the accumulators alive)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 12345.678f) p.Y[0] = s;
        if constexpr (!PERSIST) return;
    }
    if constexpr (PERSIST) {
        const int eb_ = b, em0 = m0, et0 = t0, ett = tt;
        const int vnext = vtile + (int)gridDim.x;
        const bool more_tiles = vnext < ntiles_all;               // block-uniform
        // this tile's R / coefficient loads first, the next tile's first k-tiles queued behind them, then the epilogue proper
        const bool lean = pw_tile_is_full(p, BM, em0, et0);
        PwEpilogueFull<EPI, BM, IO> ef;
        if (lean && !(PW_ABLATE & 8)) ef.issue(p, eb_, em0, et0, tid);
        if (more_tiles) {
            retarget(vnext);
            load_tiles(0, S0{});
            if (nk > 1) load_tiles(BF_BK, S1{});
        }
        if constexpr (!(PW_ABLATE & 8)) {
            if (lean) ef.finish(p, acc, reinterpret_cast<float*>(smem), eb_, ett, wm, wn, tid);
            else pw_epilogue_lds<EPI, BM, IO>(p, acc, reinterpret_cast<float*>(smem), eb_, em0, et0, ett, wm, wn, tid);
        }
        if (more_tiles) {
            vtile = vnext;
            zero_acc();
            __syncthreads();               // the epilogue's reads of the parked tile are done: the stage buffers are free again
            continue;
        }
    } else {
        pw_epilogue_lds<EPI, BM, IO>(p, acc, reinterpret_cast<float*>(smem), b, m0, t0, tt, wm, wn, tid);
    }
    break;
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised NN GEMM (round 3) for the act16 training combinations whose X operands are bf16-stored: the 256 x 128 x 64 tile,
// LDS images, fragment reads and epilogue of pw_gemm_bf16_fast_kernel<.., 256, ..>, but TWELVE waves with two roles.
//   waves 0-7  ("matrix" waves, two per SIMD): fetch the A tile (weights: no transform) two k-tiles ahead by LDS-DMA
//              (`buffer_load_dwordx4 ... lds`, swizzle applied on the SOURCE address, three 32 KB ring slots), read fragments, MFMA.
//   waves 8-11 ("staging" waves, one per SIMD): the X tile -- register loads NSX k-tiles ahead, the on-load transform (BatchNorm +
//              ReLU6 / BatchNorm-backward affine, coefficients resident in LDS), bf16 packing, `ds_write_b128` into two 16 KB slots.
// Why (measurements: profiles/r03_ws_gemm.txt).  In the 8-wave kernel every wave does load-issue, MFMA block, transform + LDS
// stores one after the other and all eight in step (one barrier per k-tile): a k-tile takes the SUM, ~2500 cycles for 1024 cycles
// of MFMA per SIMD.  Timing-only builds of the first wave-specialised form (staging waves doing A and X through registers) put
// the cost on the VGPR -> LDS store path (`ds_write_b128`: ~79 B / clock / CU, MI355X_MICROARCH LDS table; 48 KB per k-tile) and on
// load ISSUE (64 B / clock / CU address unit), not on memory latency (every load redirected to one L2-resident tile: same time)
// and not on the transform (28 VALU per 8 x 4 patch column).  So: A never touches a VGPR, its DMA issue (~60+ cycles a piece) sits
// in the matrix waves where the partner wave's MFMAs cover it, and the staging waves are left with a third of the bytes.
// One barrier per k-tile; the matrix waves count their DMAs by hand (inline asm: beside a DMA it knows about hipcc waits vmcnt(0)
// before every LDS read) and use a raw s_barrier, the staging waves use ordinary loads / __syncthreads().
constexpr int WS_MAXK = 2048;
template <int XM, int EPI, int IO>
__global__ __launch_bounds__(768) void pw_gemm_bf16_ws_kernel(PwParams p) {
    static_assert((IO & PW_IO_X) != 0 && (XM != PW_X_AFFINE2 || (IO & PW_IO_X2) != 0), "bf16-stored X operands only");
    // PW_IO_F16 (inference at precision "fp16"): the stored words ARE the operand format -- the staging waves' decode / re-encode of a
    // plain X (bf16 -> fp32 -> bf16) is the identity on every finite 16-bit pattern, so only the matrix instruction changes
    constexpr bool F16 = (IO & PW_IO_F16) != 0;
    static_assert(!F16 || XM == PW_X_NONE, "fp16 storage: plain X operand only (inference)");
    constexpr int BM = 256;
    constexpr int A_BYTES = BM * 128, X_BYTES = 128 * 128;
    constexpr int SMEM = BM * 128 * 4;                  // stages: A 2 x 32 KB + X 2 x 16 KB; the epilogue's [256][128] fp32 tile = 128 KB
    constexpr int NCF = XM == PW_X_NONE ? 0 : (XM == PW_X_AFFINE2 ? 3 : 2);      // coefficient arrays kept in LDS ([WS_MAXK] floats each)
    constexpr int NSX = XM == PW_X_AFFINE2 ? 3 : 4;     // register stages of X (tiles in flight per staging wave)
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM + NCF * WS_MAXK * 4 + (NCF ? 2048 : 0)];   // (+ slack: tiles past the last are staged too)
    unsigned char* As = smem;                           // [2][256][64] bf16
    unsigned char* Bs = smem + 2 * A_BYTES;             // [2][128][64] bf16
    float* cf = reinterpret_cast<float*>(smem + SMEM);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, tt, mt;
    pw_work(p, b, tt, mt);
    const int m0 = mt * BM, t0 = tt * PW_BN;
    const int M = p.M, K = p.K;
    const int P16 = pw_pitch16(p.T, p.B);
    const int nk = (K + BF_BK - 1) / BF_BK;

    if (wave >= 8) {
        // ------------------------------------------------ staging waves: 256 threads ------------------------------------------------
        const int pt = tid - 512;
        const __amdgpu_buffer_rsrc_t rX = make_rsrc(reinterpret_cast<const char*>(p.X) + (size_t)b * K * P16 * 2, (unsigned)K * P16 * 2u);
        const __amdgpu_buffer_rsrc_t rX2 = make_rsrc(reinterpret_cast<const char*>(XM == PW_X_AFFINE2 ? p.X2 : p.X) + (size_t)b * K * P16 * 2,
                                                     (unsigned)K * P16 * 2u);
        const int b_tq = (pt & 31) * 4, b_kg = pt >> 5;                 // X patch: t columns b_tq .. +3, k rows 8 b_kg .. +7
        const int voX0 = (8 * b_kg * P16 + t0 + b_tq) * 2;
        const int stepX = P16 * 2;
        int ldsB[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ldsB[q] = bf_off(b_tq + q, b_kg);
        if constexpr (XM != PW_X_NONE) {                                // coefficients -> LDS, once
            for (int k = pt; k < K; k += 256) {
                cf[k] = p.xa[k];
                cf[WS_MAXK + k] = p.xb[k];
                if constexpr (XM == PW_X_AFFINE2) cf[2 * WS_MAXK + k] = p.xc[k];
            }
        }
        u32x2 rb[NSX][8], rb2[NSX][XM == PW_X_AFFINE2 ? 8 : 1];
#define WS_SB() __builtin_amdgcn_sched_barrier(0)
        // PW_WS_ABL: timing-only builds (tools/ab_variants.sh) -- 1 no transform / LDS stores (loads kept), 2 raw X stores (no
        // transform), 4 no fragment reads / MFMAs, 8 no X loads inside the loop, 16 no A DMA inside the loop
        // every load is unconditional (a conditional one degrades hipcc's counted waits): tiles past the last read rows beyond the
        // descriptor (zeros, no traffic)
        auto load_x = [&](int kt, auto stg, int e) {
            constexpr int SG = decltype(stg)::value;
            const int so = (kt * BF_BK + e) * stepX;
            if ((PW_WS_ABL & 8) && kt >= NSX) return;
            rb[SG][e] = __builtin_amdgcn_raw_buffer_load_b64(rX, voX0, so, 0);
            if constexpr (XM == PW_X_AFFINE2) rb2[SG][e] = __builtin_amdgcn_raw_buffer_load_b64(rX2, voX0, so, 0);
        };
        // tile kt (registers of stage SG) -> LDS slot kt & 1; tile kt + NSX requested into the same registers
        auto stage = [&](int kt, auto stg) {
            constexpr int SG = decltype(stg)::value;
            unsigned char* Bd = Bs + (kt & 1) * X_BYTES;
            if constexpr (PW_WS_ABL & 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { asm volatile("" :: "v"(rb[SG][e])); load_x(kt + NSX, stg, e); }
                return;
            }
            f32x4 ca[2], cb[2], cc[2];
            if constexpr (XM != PW_X_NONE) {
                const float* c0 = cf + kt * BF_BK + 8 * b_kg;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ca[h] = *reinterpret_cast<const f32x4*>(c0 + 4 * h);
                    cb[h] = *reinterpret_cast<const f32x4*>(c0 + WS_MAXK + 4 * h);
                    if constexpr (XM == PW_X_AFFINE2) cc[h] = *reinterpret_cast<const f32x4*>(c0 + 2 * WS_MAXK + 4 * h);
                }
            }
            u32x4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (PW_WS_ABL & 2) {
                    o[q] = (u32x4){rb[SG][0][q >> 1], rb[SG][2][q >> 1], rb[SG][4][q >> 1], rb[SG][6][q >> 1]};
                    continue;
                }
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = pw_bf16_at(rb[SG][e], q);
                    if constexpr (XM == PW_X_NONE) v[e] = x;
                    else if constexpr (XM == PW_X_AFFINE_RELU6) v[e] = relu6f(fmaf(x, ca[e >> 2][e & 3], cb[e >> 2][e & 3]));
                    else v[e] = fmaf(x, ca[e >> 2][e & 3], fmaf(pw_bf16_at(rb2[SG][e], q), cb[e >> 2][e & 3], cc[e >> 2][e & 3]));
                }
                o[q][0] = pack_bf16(v[0], v[1]); o[q][1] = pack_bf16(v[2], v[3]); o[q][2] = pack_bf16(v[4], v[5]); o[q][3] = pack_bf16(v[6], v[7]);
            }
            WS_SB();
#pragma unroll
            for (int q = 0; q < 4; ++q) {                   // the stores and the next requests interleaved: neither queue backs up
                *reinterpret_cast<u32x4*>(Bd + ldsB[q]) = o[q];
                load_x(kt + NSX, stg, 2 * q);
                load_x(kt + NSX, stg, 2 * q + 1);
                WS_SB();
            }
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
#pragma unroll
        for (int e = 0; e < 8; ++e) load_x(0, S0{}, e);
        WS_SB();
#pragma unroll
        for (int e = 0; e < 8; ++e) load_x(1, S1{}, e);
        WS_SB();
#pragma unroll
        for (int e = 0; e < 8; ++e) load_x(2, S2{}, e);
        WS_SB();
        if constexpr (NSX == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) load_x(3, S3{}, e);
            WS_SB();
        }
        if constexpr (XM != PW_X_NONE) __syncthreads();    // (a) the coefficients are in LDS (only these waves use them, but every wave
                                                           //     counts at the barrier)
        stage(0, S0{});
        __syncthreads();                                   // (b) tile 0 is in LDS
        // During the matrix waves' tile kt: tile kt + 1 -> LDS.  NSX tiles per trip with EXITS rather than skipped bodies (a skipped
        // body is a path on which the registers of the next one are the youngest loads, and hipcc then waits vmcnt(0) everywhere);
        // the stores are unconditional (a tile past the last is zeros, lands in the slot nobody reads, before the barrier that
        // precedes the epilogue's use of the LDS).
        // (Round 5: WHOLE trips without exits, then the last nk % NSX tiles straight-line.  With an exit behind every tile each `break`
        //  is a predecessor of the loop head -- hipcc's structurizer routes them through the latch -- and the wait at the head then
        //  covers the path on which stage 1 was requested LAST: the first tile of EVERY trip waited vmcnt(0), i.e. drained the three
        //  younger stages it exists to keep in flight, one exposed memory latency per NSX tiles.)
        int kt = 0;
        for (; kt + NSX <= nk; kt += NSX) {
            stage(kt + 1, S1{});
            __syncthreads();
            stage(kt + 2, S2{});
            __syncthreads();
            if constexpr (NSX == 4) {
                stage(kt + 3, S3{});
                __syncthreads();
            }
            stage(kt + NSX, S0{});
            __syncthreads();
        }
        if (kt < nk) {
            stage(kt + 1, S1{});
            __syncthreads();
            if (kt + 1 < nk) {
                stage(kt + 2, S2{});
                __syncthreads();
                if constexpr (NSX == 4) {
                    if (kt + 2 < nk) {
                        stage(kt + 3, S3{});
                        __syncthreads();
                    }
                }
            }
        }
#undef WS_SB
        __syncthreads();                                   // the epilogue's one barrier (accumulators parked in LDS)
        return;
    }

    // ---------------------------------------------------- matrix waves: 0-7 ----------------------------------------------------
    // A: wave w stages rows 32 w ... + 31 of the tile (4 pieces of 8 rows x 128 B: lane l -> row + l / 8, chunk l % 8), two register
    // stages, one piece stored and the piece two tiles ahead requested behind each k-step's MFMAs
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.Abf, (unsigned)M * K * 2u);
    const int arow = 32 * wave + (lane >> 3);
    const int voA0 = ((m0 + arow) * K + (lane & 7) * 8) * 2;
    const int ldsA0 = bf_off(arow, lane & 7);            // piece i: + 8 rows = + 1024 bytes (same swizzle key only for i even ...)
    int ldsA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ldsA[i] = bf_off(arow + 8 * i, lane & 7);
    (void)ldsA0;
    const int stepA = 16 * K;                            // 8 rows of A, bytes
    u32x4 ra[2][4];
    auto load_a = [&](int kt, auto stg, int i) {
        constexpr int SG = decltype(stg)::value;
        if ((PW_WS_ABL & 16) && kt >= 2) return;
        ra[SG][i] = __builtin_amdgcn_raw_buffer_load_b128(rA, voA0, kt * (BF_BK * 2) + i * stepA, 0);
    };
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;
    const int rdA0 = (wm * 64 + lr) * 128, rdB0 = (wn * 64 + lr) * 128;
    // tile kt: fragments from slot kt & 1; A tile kt + 1 (registers of stage SG) -> slot (kt + 1) & 1, tile kt + 3 requested
    auto block = [&](int kt, auto stg) {
        constexpr int SG = decltype(stg)::value;
        const unsigned char* Ab = As + (kt & 1) * A_BYTES;
        const unsigned char* Bb = Bs + (kt & 1) * X_BYTES;
        unsigned char* Ad = As + ((kt + 1) & 1) * A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {
            if constexpr (!(PW_WS_ABL & 4)) {
                const int co = ((ks * 2 + lh) ^ sw) << 4;
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + co);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + 32 * 128 + co);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bb + rdB0 + co);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bb + rdB0 + 32 * 128 + co);
                acc[0][0] = mfma16<F16>(a0, b0, acc[0][0]);
                acc[0][1] = mfma16<F16>(a0, b1, acc[0][1]);
                acc[1][0] = mfma16<F16>(a1, b0, acc[1][0]);
                acc[1][1] = mfma16<F16>(a1, b1, acc[1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            *reinterpret_cast<u32x4*>(Ad + ldsA[ks]) = ra[SG][ks];
            load_a(kt + 3, stg, ks);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(0, S0{}, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(1, S1{}, i);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (XM != PW_X_NONE) __syncthreads();        // (a) see the staging waves
#pragma unroll
    for (int i = 0; i < 4; ++i) { *reinterpret_cast<u32x4*>(As + ldsA[i]) = ra[0][i]; load_a(2, S0{}, i); }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                       // (b) tile 0 is in LDS
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {                         // pairs, then the odd tile (no skipped bodies: see the staging waves)
        block(kt, S1{});
        __syncthreads();
        block(kt + 1, S0{});
        __syncthreads();
    }
    if (kt < nk) {
        block(kt, S1{});
        __syncthreads();
    }
    pw_epilogue_lds<EPI, BM, IO>(p, acc, reinterpret_cast<float*>(smem), b, m0, t0, tt, wm, wn, tid);
}

// Backward-weight, bf16: contraction index is t; both operands are read as 8 consecutive t
// (two float4), transformed, rounded and written as one 16-byte chunk of a [row][t] image.
template <int GM_, int XM_, bool TV>
__global__ __launch_bounds__(256) void pw_wgrad_bf16_kernel(WgParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2][128 * 128];   // [m][t] bf16
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][128 * 128];   // [k][t] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int s, mt, ktile;
    wg_work(p, s, mt, ktile);
    const int m0 = mt * PW_BM, n0 = ktile * PW_BN;
    const int M = p.M, K = p.K, T = p.T;
    const int g_mode = PW_MODE(GM_, p.g_mode), x_mode = PW_MODE(XM_, p.x_mode);

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
    const WgSpan sp = wg_span(p, s, nt);
    const int nsteps = sp.nb * sp.ntl, b_lo = sp.b_lo;
    const int lr = lane & 31, lh = lane >> 5;
    if (nsteps > 0) {
        load_tiles(b_lo, sp.t_first * BF_BK);
        store_tiles(0, sp.t_first * BF_BK);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const int nxt = st + 1;
        const int nb = b_lo + nxt / sp.ntl, ntt = (sp.t_first + nxt % sp.ntl) * BF_BK;
        if (nxt < nsteps) load_tiles(nb, ntt);
        __builtin_amdgcn_sched_barrier(0);
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
        // ... and first USED after it: without this fence hipcc hoists the staging arithmetic (and the
        // vmcnt wait it needs) above the MFMAs, which exposes the whole memory latency every k-step.
        asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));   // accumulators stay in AGPRs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("" : "+v"(ra[i][0]), "+v"(ra[i][1]), "+v"(rb[i][0]), "+v"(rb[i][1]));
            if (g_mode == PW_X_AFFINE2) asm volatile("" : "+v"(ra2[i][0]), "+v"(ra2[i][1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nxt < nsteps) store_tiles(cur ^ 1, ntt);
        __syncthreads();
    }
    const int col = lane & 31, half = lane >> 5;
    if constexpr (PW_ABLATE & 8) {            // timing-only: no epilogue (keep the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 12345.678f) p.partial[0] = sum;
        return;
    }
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


// Fast path of the backward-weight kernel for T % 64 == 0: buffer loads with per-batch descriptors
// (rows past M / K read as zero in hardware), per-lane offsets computed once.
template <int GM, int XM, bool TAIL, bool TAPS = false, int IO = 0>
__global__ __launch_bounds__(256) void pw_wgrad_bf16_fast_kernel(WgParams p) {
    static_assert(!TAPS || (GM == PW_X_NONE && XM == PW_X_NONE), "tap-addressed X has no prologues");
    static_assert(!(IO != 0 && TAPS), "16-bit activation storage: plain operands only");
    // operands stored as bf16 [B][rows][pw_pitch16(T, p.B)]: the 8 consecutive t of a piece are ONE 16-byte load
    constexpr bool GB = (IO & WG_IO_G) != 0, G2B = (IO & WG_IO_G2) != 0, XB = (IO & WG_IO_X) != 0;
    __shared__ __attribute__((aligned(16))) unsigned char As[2][128 * 128];   // [m][t] bf16
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][128 * 128];   // [k][t] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int s, mt, ktile;
    wg_work(p, s, mt, ktile);
    const int m0 = mt * PW_BM, n0 = ktile * PW_BN;
    const int M = p.M, K = p.K, T = p.T;

    float ga[4], gb[4], gc[4], xa[4], xb[4];
    int voG[4], voX[4], ldsO[4], voG16[4], voX16[4];
    const int P16 = pw_pitch16(T, p.B);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = tid + 256 * i;
        const int row = piece >> 3, ch = piece & 7;
        const int m = m0 + row, k = n0 + row;
        const bool mv = m < M, kv = k < K;
        voG16[i] = (m * P16 + ch * 8) * 2;
        voX16[i] = (k * P16 + ch * 8) * 2;
        ga[i] = (GM != PW_X_NONE) ? p.ga[mv ? m : 0] : 1.f;
        gb[i] = (GM != PW_X_NONE) ? p.gb[mv ? m : 0] : 0.f;
        gc[i] = (GM == PW_X_AFFINE2) ? p.gc[mv ? m : 0] : 0.f;
        xa[i] = (XM != PW_X_NONE) ? p.xa[kv ? k : 0] : 1.f;
        xb[i] = (XM != PW_X_NONE) ? p.xb[kv ? k : 0] : 0.f;
        voG[i] = (m * T + ch * 8) * 4;
        voX[i] = (k * T + ch * 8) * 4;
        if constexpr (TAPS) {                  // column k of dW = tap * cx + c: row c of the padded X, shifted (see WgParams)
            const int tap = kv ? k / p.cx : 0;
            voG[i] = (m * p.Tg + p.g_off + ch * 8) * 4;
            voX[i] = kv ? ((k - tap * p.cx) * p.Tx + pw_tap_shift(p.shifts, tap) + ch * 8) * 4 : 0x7fffff00;   // past the descriptor: zero
        }
        ldsO[i] = bf_off(row, ch);
    }
    const int Tg = TAPS ? p.Tg : T, Kx = TAPS ? p.cx : K, Tx = TAPS ? p.Tx : T;

    u32x4 ra[4][GB ? 1 : 2], ra2[4][G2B ? 1 : 2], rb[4][XB ? 1 : 2];
    auto load_tiles = [&](int b, int t0) {
        const __amdgpu_buffer_rsrc_t rG = GB ? make_rsrc(reinterpret_cast<const u16*>(p.G) + (size_t)b * M * P16, (unsigned)M * P16 * 2u)
                                             : make_rsrc(p.G + (size_t)b * M * Tg, (unsigned)M * Tg * 4u);
        const float* g2p = GM == PW_X_AFFINE2 ? p.G2 : p.G;
        const __amdgpu_buffer_rsrc_t rG2 = (GM == PW_X_AFFINE2 ? G2B : GB)
                                               ? make_rsrc(reinterpret_cast<const u16*>(g2p) + (size_t)b * M * P16, (unsigned)M * P16 * 2u)
                                               : make_rsrc(g2p + (size_t)b * M * Tg, (unsigned)M * Tg * 4u);
        const __amdgpu_buffer_rsrc_t rX = XB ? make_rsrc(reinterpret_cast<const u16*>(p.X) + (size_t)b * Kx * P16, (unsigned)Kx * P16 * 2u)
                                             : make_rsrc(p.X + (size_t)b * Kx * Tx, (unsigned)Kx * Tx * 4u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (GB) ra[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rG, voG16[i], t0 * 2, 0);
            if constexpr (GM == PW_X_AFFINE2 && G2B) ra2[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rG2, voG16[i], t0 * 2, 0);
            if constexpr (XB) rb[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rX, voX16[i], t0 * 2, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (!GB) ra[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rG, voG[i] + 16 * h, t0 * 4, 0);
                if constexpr (GM == PW_X_AFFINE2 && !G2B) ra2[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rG2, voG[i] + 16 * h, t0 * 4, 0);
                if constexpr (!XB) rb[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rX, voX[i] + 16 * h, t0 * 4, 0);
            }
        }
    };
    auto store_tiles = [&](int buf, int t0) {
        if constexpr (PW_ABLATE & 32) return;              // timing-only: no transform / LDS stores
        const bool tail = TAIL && (t0 + BF_BK > T);        // contraction index past T must contribute zero
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float va[8], vb[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float gv;
                    if constexpr (GB) gv = pw_bf16_at(ra[i][0], 4 * h + e);
                    else gv = __builtin_bit_cast(f32x4, ra[i][h])[e];
                    if constexpr (GM == PW_X_AFFINE2) {
                        float g2;
                        if constexpr (G2B) g2 = pw_bf16_at(ra2[i][0], 4 * h + e);
                        else g2 = __builtin_bit_cast(f32x4, ra2[i][h])[e];
                        gv = fmaf(gv, ga[i], fmaf(g2, gb[i], gc[i]));
                    } else if constexpr (GM == PW_X_AFFINE_RELU6) gv = relu6f(fmaf(gv, ga[i], gb[i]));
                    float xv;
                    if constexpr (XB) xv = pw_bf16_at(rb[i][0], 4 * h + e);
                    else xv = __builtin_bit_cast(f32x4, rb[i][h])[e];
                    if constexpr (XM == PW_X_AFFINE_RELU6) xv = relu6f(fmaf(xv, xa[i], xb[i]));
                    va[4 * h + e] = gv;
                    vb[4 * h + e] = xv;
                }
            }
            // rows past M / K were read as zero, but an affine transform of zero is not zero: kill them
            if constexpr (GM != PW_X_NONE) { if (voG[i] >= M * T * 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) va[e] = 0.f; } }
            if constexpr (XM != PW_X_NONE) { if (voX[i] >= K * T * 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) vb[e] = 0.f; } }
            if constexpr (TAIL) if (tail) {
                const int tb = t0 + (((tid + 256 * i) & 7) << 3);
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (tb + e >= T) { va[e] = 0.f; vb[e] = 0.f; } }
            }
            u32x4 oa, ob;
            oa[0] = pack_bf16(va[0], va[1]); oa[1] = pack_bf16(va[2], va[3]); oa[2] = pack_bf16(va[4], va[5]); oa[3] = pack_bf16(va[6], va[7]);
            ob[0] = pack_bf16(vb[0], vb[1]); ob[1] = pack_bf16(vb[2], vb[3]); ob[2] = pack_bf16(vb[4], vb[5]); ob[3] = pack_bf16(vb[6], vb[7]);
            *reinterpret_cast<u32x4*>(&As[buf][ldsO[i]]) = oa;
            *reinterpret_cast<u32x4*>(&Bs[buf][ldsO[i]]) = ob;
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
    const WgSpan sp = wg_span(p, s, nt);
    const int nsteps = sp.nb * sp.ntl, b_lo = sp.b_lo;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;
    const int rdA0 = (wm * 64 + lr) * 128, rdB0 = (wn * 64 + lr) * 128;
    if (nsteps > 0) {
        load_tiles(b_lo, sp.t_first * BF_BK);
        store_tiles(0, sp.t_first * BF_BK);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const int nxt = st + 1;
        if (nxt < nsteps) load_tiles(b_lo + nxt / sp.ntl, (sp.t_first + nxt % sp.ntl) * BF_BK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {
            const int co = ((ks * 2 + lh) ^ sw) << 4;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[cur][rdA0 + co]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[cur][rdA0 + 32 * 128 + co]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[cur][rdB0 + co]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[cur][rdB0 + 32 * 128 + co]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));   // accumulators stay in AGPRs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("" : "+v"(ra[i][0]), "+v"(rb[i][0]));
            if constexpr (!GB) asm volatile("" : "+v"(ra[i][1]));
            if constexpr (!XB) asm volatile("" : "+v"(rb[i][1]));
            if constexpr (GM == PW_X_AFFINE2) {
                asm volatile("" : "+v"(ra2[i][0]));
                if constexpr (!G2B) asm volatile("" : "+v"(ra2[i][1]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nxt < nsteps) store_tiles(cur ^ 1, (sp.t_first + nxt % sp.ntl) * BF_BK);
        __syncthreads();
    }
    const int col = lane & 31, half = lane >> 5;
    if constexpr (PW_ABLATE & 8) {            // timing-only: no epilogue (keep the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 12345.678f) p.partial[0] = sum;
        return;
    }
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
// Backward-weight with a 256-row tile on the PLAIN operand (act16 training combinations).  In the 128 x 128 kernel above every
// thread transforms 8 + 8 elements per piece of both operands -- BatchNorm-backward affine of two bf16 tensors on G, or
// BatchNorm + ReLU6 on X -- and each operand tile is transformed again by every workgroup along the other tile axis (4x at
// 512 channels): ablation (PW_ABLATE=32, no transform / LDS stores) takes the project gradient from 62 to 41 us, so the kernel is
// bound by VALU issue of the staging, not by the matrix pipe or memory.  Here the block tile is GR x XR = 128 x 256 or 256 x 128
// with the 128 rows on the TRANSFORMED operand: half the redundant transforms per MFMA, and a plain bf16 operand is copied to
// LDS as loaded (no unpack / repack).  8 waves of 64 x 64, one workgroup per CU, same LDS images and fragment reads.
template <int GM, int XM, bool TAIL, int IO, int GR, int XR, int NST>
__global__ __launch_bounds__(512) void pw_wgrad_bf16_wide_kernel(WgParams p) {
    static_assert(GR % 64 == 0 && XR % 64 == 0 && (GR / 64) * (XR / 64) == 8, "8 waves of 64 x 64");
    constexpr bool GB = (IO & WG_IO_G) != 0, G2B = (IO & WG_IO_G2) != 0, XB = (IO & WG_IO_X) != 0;
    constexpr int NG = GR / 64, NX = XR / 64;               // 16-byte LDS pieces per thread and operand
    constexpr bool GCOPY = GB && GM == PW_X_NONE, XCOPY = XB && XM == PW_X_NONE;    // stored as loaded
    __shared__ __attribute__((aligned(16))) unsigned char As[2][GR * 128];   // [m][t] bf16
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][XR * 128];   // [k][t] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NX, wn = wave % NX;
    int s, mt, ktile;
    wg_work(p, s, mt, ktile);
    const int m0 = mt * GR, n0 = ktile * XR;
    const int M = p.M, K = p.K, T = p.T;
    const int P16 = pw_pitch16(T, p.B);

    float ga[NG], gb[NG], gc[NG], xa[NX], xb[NX];
    int voG[NG], voX[NX], ldsG[NG], ldsX[NX];
    bool gv_[NG], xv_[NX];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int piece = tid + 512 * i;
        const int row = piece >> 3, ch = piece & 7;
        const int m = m0 + row;
        gv_[i] = m < M;
        voG[i] = GB ? (m * P16 + ch * 8) * 2 : (m * T + ch * 8) * 4;
        ga[i] = (GM != PW_X_NONE) ? p.ga[gv_[i] ? m : 0] : 1.f;
        gb[i] = (GM != PW_X_NONE) ? p.gb[gv_[i] ? m : 0] : 0.f;
        gc[i] = (GM == PW_X_AFFINE2) ? p.gc[gv_[i] ? m : 0] : 0.f;
        ldsG[i] = bf_off(row, ch);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int piece = tid + 512 * i;
        const int row = piece >> 3, ch = piece & 7;
        const int k = n0 + row;
        xv_[i] = k < K;
        voX[i] = XB ? (k * P16 + ch * 8) * 2 : (k * T + ch * 8) * 4;
        xa[i] = (XM != PW_X_NONE) ? p.xa[xv_[i] ? k : 0] : 1.f;
        xb[i] = (XM != PW_X_NONE) ? p.xb[xv_[i] ? k : 0] : 0.f;
        ldsX[i] = bf_off(row, ch);
    }
    // a second G tensor (affine2) may have a different storage type than the first
    int voG2[GM == PW_X_AFFINE2 ? NG : 1];
    if constexpr (GM == PW_X_AFFINE2) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int piece = tid + 512 * i;
            const int m = m0 + (piece >> 3), ch = piece & 7;
            voG2[i] = G2B ? (m * P16 + ch * 8) * 2 : (m * T + ch * 8) * 4;
        }
    }

    // NST = 2 register stages: the tile of step st + 2 is requested before the MFMAs of step st and first used after the MFMAs
    // of step st + 1 (with one stage a step lasts about one memory latency: 8 waves per CU, nothing else to run meanwhile).
    // Measured: the project gradient 57 -> 53.5 us; the expand gradient, whose fp32 X pieces make a stage 48 registers, spills
    // with two stages (69 -> 82 us) and keeps one.
    // NST = 3: two stages for G, ONE for X -- for the expand gradient: its X (the block input, fp32, 33 MB re-read by all 16 row
    // tiles: L2 hits) is requested one step ahead, its G streams (dz1 and a1 from HBM) two steps ahead; 64 staging registers.
    constexpr int NSG = NST >= 2 ? 2 : 1, NSX = NST == 2 ? 2 : 1;
    u32x4 ra[NSG][NG][GB ? 1 : 2], ra2[NSG][GM == PW_X_AFFINE2 ? NG : 1][G2B ? 1 : 2], rb[NSX][NX][XB ? 1 : 2];
    auto load_x = [&](auto stg, int b, int t0) {
        constexpr int SX = NSX == 2 ? decltype(stg)::value : 0;
        const __amdgpu_buffer_rsrc_t rX = XB ? make_rsrc(reinterpret_cast<const u16*>(p.X) + (size_t)b * K * P16, (unsigned)K * P16 * 2u)
                                             : make_rsrc(p.X + (size_t)b * K * T, (unsigned)K * T * 4u);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int h = 0; h < (XB ? 1 : 2); ++h) rb[SX][i][h] = __builtin_amdgcn_raw_buffer_load_b128(rX, voX[i] + 16 * h, t0 * (XB ? 2 : 4), 0);
        }
    };
    auto load_tiles = [&](auto stg, int b, int t0) {
        constexpr int SG = decltype(stg)::value;
        const __amdgpu_buffer_rsrc_t rG = GB ? make_rsrc(reinterpret_cast<const u16*>(p.G) + (size_t)b * M * P16, (unsigned)M * P16 * 2u)
                                             : make_rsrc(p.G + (size_t)b * M * T, (unsigned)M * T * 4u);
        const float* g2p = GM == PW_X_AFFINE2 ? p.G2 : p.G;
        const __amdgpu_buffer_rsrc_t rG2 = G2B ? make_rsrc(reinterpret_cast<const u16*>(g2p) + (size_t)b * M * P16, (unsigned)M * P16 * 2u)
                                               : make_rsrc(g2p + (size_t)b * M * T, (unsigned)M * T * 4u);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
#pragma unroll
            for (int h = 0; h < (GB ? 1 : 2); ++h) ra[SG][i][h] = __builtin_amdgcn_raw_buffer_load_b128(rG, voG[i] + 16 * h, t0 * (GB ? 2 : 4), 0);
            if constexpr (GM == PW_X_AFFINE2) {
#pragma unroll
                for (int h = 0; h < (G2B ? 1 : 2); ++h)
                    ra2[SG][i][h] = __builtin_amdgcn_raw_buffer_load_b128(rG2, voG2[i] + 16 * h, t0 * (G2B ? 2 : 4), 0);
            }
        }
        if constexpr (NST != 3) load_x(stg, b, t0);        // NST 3: X is requested separately, one step ahead
    };
    auto store_tiles = [&](auto stg, int buf, int t0) {
        constexpr int SG = decltype(stg)::value;
        constexpr int SX = NSX == 2 ? SG : 0;
        if constexpr (PW_ABLATE & 32) return;              // timing-only: no transform / LDS stores
        const bool tail = TAIL && (t0 + BF_BK > T);        // contraction index past T must contribute zero
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            u32x4 oa;
            if (GCOPY && !tail) {
                oa = ra[SG][i][0];
            } else {
                float va[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float gv;
                    if constexpr (GB) gv = pw_bf16_at(ra[SG][i][0], e);
                    else gv = __builtin_bit_cast(f32x4, ra[SG][i][e >> 2])[e & 3];
                    if constexpr (GM == PW_X_AFFINE2) {
                        float g2;
                        if constexpr (G2B) g2 = pw_bf16_at(ra2[SG][i][0], e);
                        else g2 = __builtin_bit_cast(f32x4, ra2[SG][i][e >> 2])[e & 3];
                        gv = fmaf(gv, ga[i], fmaf(g2, gb[i], gc[i]));
                    } else if constexpr (GM == PW_X_AFFINE_RELU6) gv = relu6f(fmaf(gv, ga[i], gb[i]));
                    va[e] = gv;
                }
                // rows past M were read as zero, but an affine transform of zero is not zero: kill them
                if constexpr (GM != PW_X_NONE) { if (!gv_[i]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) va[e] = 0.f; } }
                if constexpr (TAIL) if (tail) {
                    const int tb = t0 + (((tid + 512 * i) & 7) << 3);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { if (tb + e >= T) va[e] = 0.f; }
                }
                oa[0] = pack_bf16(va[0], va[1]); oa[1] = pack_bf16(va[2], va[3]); oa[2] = pack_bf16(va[4], va[5]); oa[3] = pack_bf16(va[6], va[7]);
            }
            *reinterpret_cast<u32x4*>(&As[buf][ldsG[i]]) = oa;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            u32x4 ob;
            if (XCOPY && !tail) {
                ob = rb[SX][i][0];
            } else {
                float vb[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xv;
                    if constexpr (XB) xv = pw_bf16_at(rb[SX][i][0], e);
                    else xv = __builtin_bit_cast(f32x4, rb[SX][i][e >> 2])[e & 3];
                    if constexpr (XM == PW_X_AFFINE_RELU6) xv = relu6f(fmaf(xv, xa[i], xb[i]));
                    vb[e] = xv;
                }
                if constexpr (XM != PW_X_NONE) { if (!xv_[i]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vb[e] = 0.f; } }
                if constexpr (TAIL) if (tail) {
                    const int tb = t0 + (((tid + 512 * i) & 7) << 3);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { if (tb + e >= T) vb[e] = 0.f; }
                }
                ob[0] = pack_bf16(vb[0], vb[1]); ob[1] = pack_bf16(vb[2], vb[3]); ob[2] = pack_bf16(vb[4], vb[5]); ob[3] = pack_bf16(vb[6], vb[7]);
            }
            *reinterpret_cast<u32x4*>(&Bs[buf][ldsX[i]]) = ob;
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
    const WgSpan sp = wg_span(p, s, nt);
    const int nsteps = sp.nb * sp.ntl, b_lo = sp.b_lo;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;
    const int rdA0 = (wm * 64 + lr) * 128, rdB0 = (wn * 64 + lr) * 128;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NSG - 1>;
    // step -> (batch element, t offset); indices past the end are clamped to the last step (an unconditional, redundant load:
    // a conditional one would make hipcc wait for the YOUNGER stage at the join)
    auto issue = [&](auto stg, int step) {
        const int q = min(step, nsteps - 1);
        load_tiles(stg, b_lo + q / sp.ntl, (sp.t_first + q % sp.ntl) * BF_BK);
    };
    auto issue_x = [&](int step) {
        const int q = min(step, nsteps - 1);
        load_x(S0{}, b_lo + q / sp.ntl, (sp.t_first + q % sp.ntl) * BF_BK);
    };
    auto mfma_block = [&](int cur) {
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {
            const int co = ((ks * 2 + lh) ^ sw) << 4;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[cur][rdA0 + co]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[cur][rdA0 + 32 * 128 + co]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[cur][rdB0 + co]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[cur][rdB0 + 32 * 128 + co]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        // (accumulators pinned as VGPRs, not AGPRs: with any AGPR use hipcc splits the 256 registers of a 512-thread block
        // 128 / 128 and the two staging stages spill)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
    };
    // the registers of stage SG are first USED after this point (a macro: clang rejects captured arrays as asm operands in a generic lambda)
#define WG_PIN(SG)                                                                                  \
    do {                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < NG; ++i_) {                                         \
            asm volatile("" : "+v"(ra[SG][i_][0]));                                                 \
            if constexpr (!GB) asm volatile("" : "+v"(ra[SG][i_][1]));                              \
            if constexpr (GM == PW_X_AFFINE2) {                                                     \
                asm volatile("" : "+v"(ra2[SG][i_][0]));                                            \
                if constexpr (!G2B) asm volatile("" : "+v"(ra2[SG][i_][1]));                        \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < NX; ++i_) {                                         \
            asm volatile("" : "+v"(rb[NSX == 2 ? SG : 0][i_][0]));                                  \
            if constexpr (!XB) asm volatile("" : "+v"(rb[NSX == 2 ? SG : 0][i_][1]));               \
        }                                                                                           \
    } while (0)
    if constexpr (NST == 1) {
        if (nsteps > 0) {
            issue(S0{}, 0);
            store_tiles(S0{}, 0, sp.t_first * BF_BK);
        }
        __syncthreads();
        for (int st = 0; st < nsteps; ++st) {
            issue(S0{}, st + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(st & 1);
            WG_PIN(0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < nsteps) store_tiles(S0{}, (st + 1) & 1, (sp.t_first + (st + 1) % sp.ntl) * BF_BK);
            __syncthreads();
        }
    } else {
    if (nsteps > 0) {
        issue(S0{}, 0);
        if constexpr (NST == 3) issue_x(0);
        issue(S1{}, 1);
        store_tiles(S0{}, 0, sp.t_first * BF_BK);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; st += 2) {
        // even step st: LDS 0; stage 1 holds step st + 1; stage 0 is free -> step st + 2
        if constexpr (NST == 3) issue_x(st + 1);           // X first: it is needed a step sooner than the G tiles requested below
        issue(S0{}, st + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(0);
        WG_PIN(1);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < nsteps) store_tiles(S1{}, 1, (sp.t_first + (st + 1) % sp.ntl) * BF_BK);
        __syncthreads();
        if (st + 1 >= nsteps) break;
        // odd step st + 1: LDS 1; stage 0 holds step st + 2; stage 1 is free -> step st + 3
        if constexpr (NST == 3) issue_x(st + 2);
        issue(S1{}, st + 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(1);
        WG_PIN(0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < nsteps) store_tiles(S0{}, 0, (sp.t_first + (st + 2) % sp.ntl) * BF_BK);
        __syncthreads();
    }
    }
#undef WG_PIN
    const int col = lane & 31, half = lane >> 5;
    if constexpr (PW_ABLATE & 8) {            // timing-only: no epilogue (keep the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 12345.678f) p.partial[0] = sum;
        return;
    }
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
// Short-K GEMM with the epilogue of tile i UNDER the main loop of tile i + 1 (plain bf16 X, bf16 Y, whole tiles; K = 256 or 512).
// The short-K GEMMs (expand forward, project backward-data: 4-8 k-tiles) spend more of a tile in their epilogue than in their main
// loop, and the two add up: the epilogue is bound by vector-instruction issue and LDS round trips with the matrix pipe idle, the
// main loop by the matrix pipe with little else to issue (profiles/r03_ws_gemm.txt: a persistent form that only prefetches across
// the epilogue, a staggered start, an early R request and a re-spaced k-tile all measured +-0).  Here a persistent workgroup keeps
// TWO accumulator sets: when tile i's k-loop ends its accumulators are set aside and the k-loop of tile i + 1 starts at once; the
// epilogue of tile i is cut into wave-LOCAL units -- park one 32 x 32 block in the wave's own 4 KB of LDS, four row passes over it
// (ds_read_b128, partial sums, bf16 pack, 8-byte stores of 64-byte row segments), a per-row reduction of the partial sums -- that
// need no workgroup barrier and are issued one or three at a time behind the MFMAs of each k-step.  Only the last step (the two
// column halves of a row meet: waves wn = 1 hand their row sums to waves wn = 0 through LDS) leans on the k-loop's own barriers.
// Staging as in the wave-specialised kernels' matrix waves (A pieces and X columns behind each k-step's MFMAs).
template <int EPI, int IO, int NK>
__global__ __launch_bounds__(512) void pw_gemm_bf16_ov_kernel(PwParams p) {
    static_assert((EPI == PW_EPI_STATS && IO == (PW_IO_X | PW_IO_Y)) || (EPI == PW_EPI_MASK_STATS && IO == (PW_IO_X | PW_IO_R | PW_IO_Y)),
                  "expand forward on the bf16 shadow / project backward-data");
    constexpr bool MASK = EPI == PW_EPI_MASK_STATS;
    static_assert(NK == 4 || NK == 8, "K = 256 or 512");
    constexpr int BM = 256;
    constexpr int A_BYTES = BM * 128, X_BYTES = 128 * 128;
    constexpr int STAGES = 2 * A_BYTES + 2 * X_BYTES;   // 96 KB
    constexpr int SCR = 8 * 4096;                       // a 32 x 32 fp32 block per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES + SCR + 4 * 32 * 16 + 512 * 8];
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* scr = reinterpret_cast<float*>(smem + STAGES + wave * 4096);
    float* cfe = reinterpret_cast<float*>(smem + STAGES + SCR + 4 * 32 * 16);       // MASK: (ea, eb) of the previous tile's 256 rows
    const int M = p.M, K = p.K;
    const int P16 = pw_pitch16(p.T, p.B);
    const int total = p.n_mtiles * p.n_ttiles * p.B;
    int v = blockIdx.x;
    int b, tt, mt;
    pw_work_v(p, v, total, b, tt, mt);
#define OV_SB() __builtin_amdgcn_sched_barrier(0)
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.Abf, (unsigned)M * K * 2u);
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(p.X, (unsigned)p.B * K * P16 * 2u);
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(p.Y, (unsigned)p.B * M * P16 * 2u);
    const int arow = 32 * wave + (lane >> 3);
    int ldsA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ldsA[i] = bf_off(arow + 8 * i, lane & 7);
    const int stepA = 16 * K, stepX = P16 * 2;
#if PW_OV_XTR
    // X image [64 k][128 t] as loaded (256-byte rows, 16-byte chunks XOR-ed with ((k & 3) << 2) | ((k >> 2) & 3)): a lane copies two
    // 16-byte chunks per k-tile (rows x_row and x_row + 32, chunk x_ch) with ds_write_b128 -- no byte permutes, no 4-way-conflicted
    // 8-byte column stores -- and the B fragments are TRANSPOSED reads (ds_read_b64_tr_b16, cdna_hip_programming.md T10 image (b)).
    const int x_row = tid >> 4, x_ch = tid & 15;
    const int x_sw = ((x_row & 3) << 2) | ((x_row >> 2) & 3);          // the same for row + 32
    const int ldsX = 256 * x_row + 16 * (x_ch ^ x_sw);
    auto vo_x = [&](int b_, int tt_) { return ((b_ * K + x_row) * P16 + tt_ * PW_BN + x_ch * 8) * 2; };
    int t_lim = p.T - tt * PW_BN - x_ch * 8;            // this lane's chunk: columns e < t_lim exist
#else
    const int b_tq = (tid & 31) * 4, b_kg = tid >> 5;
    int ldsB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ldsB[q] = bf_off(b_tq + q, b_kg >> 1) + (b_kg & 1) * 8;
    auto vo_x = [&](int b_, int tt_) { return ((b_ * K + 4 * b_kg) * P16 + tt_ * PW_BN + b_tq) * 2; };
    int t_lim = p.T - tt * PW_BN - b_tq;                // this lane's X columns q < t_lim exist
#endif
    auto vo_a = [&](int mt_) { return ((mt_ * BM + arow) * K + (lane & 7) * 8) * 2; };
    int voA = vo_a(mt), voX = vo_x(b, tt), voAn, voXn;
    auto aim_next = [&]() {
        const int vn = v + (int)gridDim.x < total ? v + (int)gridDim.x : v;
        int b_, tt_, mt_;
        pw_work_v(p, vn, total, b_, tt_, mt_);
        voAn = vo_a(mt_); voXn = vo_x(b_, tt_);
    };
    aim_next();
    // MASK: ONE register stage for A (weights: L2 hits, a k-tile of lead) and one set of R pieces -- the mask epilogue's registers
    constexpr bool A1 = true;
    constexpr int AL = A1 ? 2 : 3;                      // A k-tile requested behind the store of k-tile kt + 1: kt + AL
    u32x4 ra[A1 ? 1 : 2][4];
#if PW_OV_XTR
    u32x4 rb[2][2];
#else
    u32x2 rb[2][4];
#endif
    auto load_a = [&](int kt, auto stg, int i) {        // kt >= NK: k-tile kt - NK of the next tile
        constexpr int SG = A1 ? 0 : decltype(stg)::value;
        const bool nx = kt >= NK;
        ra[SG][i] = __builtin_amdgcn_raw_buffer_load_b128(rA, nx ? voAn : voA, (nx ? kt - NK : kt) * (BF_BK * 2) + i * stepA, 0);
    };
#if PW_OV_XTR
    auto load_x = [&](int kt, auto stg, int e) {        // e = 0, 1: the chunk in rows x_row, x_row + 32 (e = 2, 3: nothing)
        constexpr int SG = decltype(stg)::value;
        const bool nx = kt >= NK;
        if (e < 2) rb[SG][e] = __builtin_amdgcn_raw_buffer_load_b128(rX, nx ? voXn : voX, ((nx ? kt - NK : kt) * BF_BK + 32 * e) * stepX, 0);
    };
    auto store_x_col = [&](int buf, auto stg, auto qc) {
        constexpr int SG = decltype(stg)::value;
        constexpr int q = decltype(qc)::value;
        if constexpr ((q & 1) == 0) {                   // two 16-byte stores per k-tile: behind k-steps 0 and 2
            constexpr int e = q >> 1;
            // columns past T (partial last t-tile) are staged as ZEROS (see below)
            u32x4 o = rb[SG][e];
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] &= (2 * d < t_lim ? 0xffffu : 0u) | (2 * d + 1 < t_lim ? 0xffff0000u : 0u);
            *reinterpret_cast<u32x4*>(Bs + buf * X_BYTES + ldsX + e * (32 * 256)) = o;
        }
    };
#else
    auto load_x = [&](int kt, auto stg, int e) {
        constexpr int SG = decltype(stg)::value;
        const bool nx = kt >= NK;
        rb[SG][e] = __builtin_amdgcn_raw_buffer_load_b64(rX, nx ? voXn : voX, ((nx ? kt - NK : kt) * BF_BK + e) * stepX, 0);
    };
    auto store_x_col = [&](int buf, auto stg, auto qc) {
        constexpr int SG = decltype(stg)::value;
        constexpr int q = decltype(qc)::value;
        constexpr unsigned sel = (q & 1) ? 0x07060302u : 0x05040100u;
        uint2 o;
        // columns past T (partial last t-tile) are staged as ZEROS: their accumulators are exactly 0, so they add nothing to the
        // statistics and the epilogue needs no per-element column tests
        const unsigned keep = q < t_lim ? 0xffffffffu : 0u;
        o.x = __builtin_amdgcn_perm(rb[SG][1][q >> 1], rb[SG][0][q >> 1], sel) & keep;
        o.y = __builtin_amdgcn_perm(rb[SG][3][q >> 1], rb[SG][2][q >> 1], sel) & keep;
        *reinterpret_cast<uint2*>(Bs + buf * X_BYTES + ldsB[q]) = o;
    };
#endif
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2], accp[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;
    const int rdA0 = (wm * 64 + lr) * 128, rdB0 = (wn * 64 + lr) * 128;
#if PW_OV_XTR
    // B fragment of MFMA column block j, k-step ks: lane l of 16-lane group g = l >> 4 takes column 16 (g & 1) + (l & 15) of the
    // block's 32 and k = 16 ks + 8 (g >> 1) ... + 7 as two transposed reads (h = 0, 1) of 4 k-rows x 16 columns; lane 4 q + pp of
    // the group supplies the address of row r0 + q, chunk c0 + (pp >> 1), + 8 (pp & 1) bytes.  The XOR term does not depend on ks
    // (rows 16 ks apart): one address register per (j, h), k-steps by immediate offsets of 4096 bytes.
    typedef short ov_s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) ov_s16x4 ov_lds_s16x4;
    int trB[2][2];
    {
        const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 8 * (g >> 1) + 4 * h + q4;
                const int ch = ((wn * 64 + j * 32 + 16 * (g & 1)) >> 3) + (pp >> 1);
                trB[j][h] = 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (pp & 1);
            }
    }
#endif

    // ---- the previous tile's epilogue, in units ----
    int pb = 0, ptt = 0, pmt = 0;                         // its coordinates
    int ep_vo = 0, ep_tl = 0;                             // this lane's byte offset into Y / R at block (0, 0), pass 0; columns left before T
    float rs0[2] = {0.f, 0.f}, rs1[2] = {0.f, 0.f};       // lanes 0-31: row sums (y, y^2) of rows i * 32 + lane over this wave's 64 columns
    float* xarea = reinterpret_cast<float*>(smem + STAGES + SCR);      // [4 wm][32 lanes][4]
    // 25 units: block blk = u / 6 (i = blk / 2, j = blk % 2), stage w = u % 6, software-pipelined so that every LDS read is issued
    // one unit (= one k-step: four MFMAs) before its data are used -- an in-order wave that waits for an LDS round trip inside a
    // unit also holds back its next MFMA:
    //   w 0: park block blk (16 ds_write_b32)   [+ blk > 0: add up the previous block's partial sums, read in its w 5]
    //   w 1: read row pass 0      w 2-4: finish pass w - 2 (sums, pack, store, partials -> LDS), read pass w - 1
    //   w 5: finish pass 3, read this lane's four partial pairs of row (lane & 31)
    //   u 24: add up block 3's partial sums
    // MASK (project backward-data): x = (0 < ea * R + eb < 6) ? x : 0, sums (x, x * R); a block's four R pieces (a2, bf16) and row
    // coefficients are requested a whole block (six units) ahead of their row passes
    const __amdgpu_buffer_rsrc_t rR = make_rsrc(MASK ? p.R : p.X, MASK ? (unsigned)p.B * M * P16 * 2u : 0u);
    const __amdgpu_buffer_rsrc_t rEa = make_rsrc(MASK ? p.ea : p.X, MASK ? (unsigned)M * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rEb = make_rsrc(MASK ? p.eb : p.X, MASK ? (unsigned)M * 4u : 0u);
    u32x2 ep_rr[1][MASK ? 4 : 1];
    float2 ep_c[2];                                     // (ea, eb) of the row of the pass in flight, read with its data
    auto ep_r_req = [&](auto bc) {                      // block blk's R pieces (requested when the previous block's last pass is done)
        constexpr int blk = decltype(bc)::value, i = blk >> 1, j = blk & 1;
        if constexpr (MASK && !(PW_OV_ABL & 128)) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                ep_rr[0][ps] = (PW_OV_ABL & 512) ? __builtin_amdgcn_raw_buffer_load_b64(rR, (ep_vo & 0xfff8) + blockIdx.x * 65536, 0, 0)
                                                 : __builtin_amdgcn_raw_buffer_load_b64(rR, ep_vo, ((i * 32 + 8 * ps) * P16 + j * 32) * 2, 0);
        }
    };
    auto ep_coef_tile = [&]() {                         // the tile's row coefficients -> LDS (each wave: its own 64 rows, read back only by itself)
        if constexpr (MASK) {
            const int row = pmt * BM + wm * 64 + lane;
            const float a_ = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rEa, row * 4, 0, 0));
            const float b_ = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rEb, row * 4, 0, 0));
            *reinterpret_cast<float2*>(cfe + (wave * 64 + lane) * 2) = make_float2(a_, b_);      // (its own copy: no cross-wave order needed)
        }
    };
    f32x4 ep_d[2];
    float2 ep_r[4];
    auto ep_read = [&](auto psc) {
        constexpr int ps = decltype(psc)::value;
        ep_d[ps & 1] = *reinterpret_cast<const f32x4*>(scr + (8 * ps + (lane >> 3)) * 32 + (lane & 7) * 4);
    };
    auto ep_read_c = [&](auto psc, auto ic) {           // MASK: the pass's row coefficients
        constexpr int ps = decltype(psc)::value, i = decltype(ic)::value;
        if constexpr (MASK) ep_c[ps & 1] = *reinterpret_cast<const float2*>(cfe + (wave * 64 + i * 32 + 8 * ps + (lane >> 3)) * 2);
    };
    auto ep_finish = [&](auto psc, auto ic, auto jc) {
        constexpr int ps = decltype(psc)::value, i = decltype(ic)::value, j = decltype(jc)::value;
        f32x4 a = ep_d[ps & 1];
        float s0 = 0.f, s1 = 0.f;
        if constexpr (MASK) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (columns past T of a partial last t-tile: their accumulators are exactly 0 -- zero X --, but R's row padding is
                //  whatever the allocation held; the median maps a NaN / Inf there to a finite value and leaves every bf16 number as
                //  it is, so 0 * R stays 0)
                const float r = __builtin_amdgcn_fmed3f(pw_bf16_at(ep_rr[0][ps], e), -3.3895314e38f, 3.3895314e38f);
                const float pre = fmaf(r, ep_c[ps & 1].x, ep_c[ps & 1].y);
                const float x = (pre > 0.f && pre < 6.f) ? a[e] : 0.f;
                a[e] = x;
                s0 += x; s1 = fmaf(x, r, s1);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s0 += a[e]; s1 = fmaf(a[e], a[e], s1); }
        }
        const u32x2 o2 = {pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3])};
        // (a straddling lane's 8 bytes stay inside the pitched row; a lane wholly past T aims outside the descriptor: dropped)
        if constexpr (PW_OV_ABL & 256) __builtin_amdgcn_raw_buffer_store_b64(o2, rY, (ep_vo & 0xfff8) + blockIdx.x * 65536, 0, 0);     // timing-only: every store of a workgroup into one 64 KB window
        else if constexpr (!(PW_OV_ABL & 32)) __builtin_amdgcn_raw_buffer_store_b64(o2, rY, j * 32 < ep_tl ? ep_vo : 0x7ffffff0, ((i * 32 + 8 * ps) * P16 + j * 32) * 2, PW_OV_CP_Y);
        else if (o2[0] == 0x12345678u) p.stats[1] = 1.f;
        if constexpr (!(PW_OV_ABL & 64)) *reinterpret_cast<float2*>(scr + (8 * ps + (lane >> 3)) * 32 + (lane & 7) * 4) = make_float2(s0, s1);
        else if (s0 + s1 == 12345.678f) p.stats[2] = 1.f;
    };
    auto ep_red_read = [&]() {       // (start rotated by the row: 2-way bank conflicts, not 16)
        if constexpr (PW_OV_ABL & 64) return;
        const int row = lane & 31, h4 = (lane >> 5) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) ep_r[q] = *reinterpret_cast<const float2*>(scr + row * 32 + ((h4 + q + row) & 7) * 4);
    };
    auto ep_red_sum = [&](auto ic) { // this half-wave's four of the row's eight partial pairs; the halves meet in epi_xchg
        constexpr int i = decltype(ic)::value;
        rs0[i] += (ep_r[0].x + ep_r[1].x) + (ep_r[2].x + ep_r[3].x);
        rs1[i] += (ep_r[0].y + ep_r[1].y) + (ep_r[2].y + ep_r[3].y);
    };
    auto epi_unit = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if constexpr (u == 24) { ep_red_sum(std::integral_constant<int, 1>{}); return; }
        constexpr int blk = (u / 6) & 3, i = blk >> 1, j = blk & 1, w = u % 6;
        using IC = std::integral_constant<int, i>; using JC = std::integral_constant<int, j>;
        if constexpr (w == 0) {
            if constexpr (blk > 0) ep_red_sum(std::integral_constant<int, ((blk - 1) >> 1)>{});
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = accp[i][j][r];
        } else if constexpr (w == 1) {
            ep_read(std::integral_constant<int, 0>{});
            ep_read_c(std::integral_constant<int, 0>{}, IC{});
        } else if constexpr (w <= 4) {
            ep_finish(std::integral_constant<int, w - 2>{}, IC{}, JC{});
            ep_read(std::integral_constant<int, w - 1>{});
            ep_read_c(std::integral_constant<int, w - 1>{}, IC{});
        } else {
            ep_finish(std::integral_constant<int, 3>{}, IC{}, JC{});
            if constexpr (blk < 3) ep_r_req(std::integral_constant<int, (blk + 1) & 3>{});
            ep_red_read();
        }
    };
    auto epi_xchg = [&]() {                              // half-waves meet; column half 1 -> LDS
#pragma unroll
        for (int i = 0; i < 2; ++i) { rs0[i] += __shfl_xor(rs0[i], 32, 64); rs1[i] += __shfl_xor(rs1[i], 32, 64); }
        if (wn == 1 && lane < 32) *reinterpret_cast<f32x4*>(xarea + (wm * 32 + lane) * 4) = (f32x4){rs0[0], rs1[0], rs0[1], rs1[1]};
    };
    auto epi_final = [&]() {                             // column half 0 adds it and stores the tile's statistics (a barrier after xchg)
        if (wn == 0 && lane < 32) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(xarea + (wm * 32 + lane) * 4);
            const size_t part = (size_t)pb * p.n_ttiles + ptt;
            float* sp = p.stats + (part * M + pmt * BM + wm * 64 + lane) * 2;
            *reinterpret_cast<float2*>(sp) = make_float2(rs0[0] + o[0], rs1[0] + o[1]);
            *reinterpret_cast<float2*>(sp + 64) = make_float2(rs0[1] + o[2], rs1[1] + o[3]);
        }
        rs0[0] = rs0[1] = rs1[0] = rs1[1] = 0.f;
    };
    // units of k-step `slot` of a k-loop that carries an epilogue: 25 data units over slots 0 .. S - 6, xchg in the last slot of k-tile
    // NK - 2, final in the last slot of all (one barrier between them)
    constexpr int NU = 25, SLOTS = 4 * NK, DSL = SLOTS - 5, UPS = (NU + DSL - 1) / DSL;
    auto epi_slot = [&](auto sc) {
        constexpr int slot = decltype(sc)::value;
#define OV_U(k_) if constexpr (slot * UPS + (k_) < NU && (k_) < UPS) epi_unit(std::integral_constant<int, (slot * UPS + (k_) < NU ? slot * UPS + (k_) : 0)>{});
        OV_U(0) OV_U(1) OV_U(2)
#undef OV_U
        if constexpr (slot == SLOTS - 5) epi_xchg();
        if constexpr (slot == SLOTS - 1) epi_final();
    };

    auto mfma_step = [&](int kt, int ks) {
        if constexpr (PW_OV_ABL & 16) {                  // timing-only: no fragment reads (operands = whatever the lane id makes)
            const bf16x8 a0 = {(short)lane, 1, 2, 3, 4, 5, 6, 7}, b0 = {(short)ks, 1, 2, 3, 4, 5, 6, (short)kt};
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[1][1], 0, 0, 0);
            return;
        }
        const unsigned char* Ab = As + (kt & 1) * A_BYTES;
        const unsigned char* Bb = Bs + (kt & 1) * X_BYTES;
        const int co = ((ks * 2 + lh) ^ sw) << 4;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + co);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + 32 * 128 + co);
#if PW_OV_XTR
        const ov_s16x4 b0l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ov_lds_s16x4*)(Bb + trB[0][0] + ks * 4096));
        const ov_s16x4 b0h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ov_lds_s16x4*)(Bb + trB[0][1] + ks * 4096));
        const ov_s16x4 b1l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ov_lds_s16x4*)(Bb + trB[1][0] + ks * 4096));
        const ov_s16x4 b1h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ov_lds_s16x4*)(Bb + trB[1][1] + ks * 4096));
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0l, b0h, 0, 1, 2, 3, 4, 5, 6, 7));
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b1l, b1h, 0, 1, 2, 3, 4, 5, 6, 7));
#else
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bb + rdB0 + co);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bb + rdB0 + 32 * 128 + co);
#endif
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
    // k-tile kt (static): fragments from slot kt & 1; k-tile kt + 1 (registers of stage (kt + 1) & 1) -> LDS, k-tile kt + 3 requested
    // (none of it in the last k-tile); EP: the previous tile's epilogue units of these four k-steps
    auto ktile = [&](auto ktc, auto epc) {
        constexpr int kt = decltype(ktc)::value;
        constexpr bool EP = decltype(epc)::value;
        constexpr bool STAGE = kt < NK - 1;
        using SG = std::integral_constant<int, (kt + 1) & 1>;
        constexpr int nb = (kt + 1) & 1;
        unsigned char* Ad = As + nb * A_BYTES;
        auto a_piece = [&](int i) {
            if constexpr (STAGE && !(PW_OV_ABL & 4)) { *reinterpret_cast<u32x4*>(Ad + ldsA[i]) = ra[A1 ? 0 : SG::value][i]; load_a(kt + AL, SG{}, i); }
        };
        // no fence between a k-step's MFMAs and its other work; the scheduler is asked for the pipeline
        // reads, MFMA, 8 others, MFMA, 8 others, MFMA, 8 others, MFMA, rest -- an in-order wave issues nothing else while it waits for
        // the matrix pipe to take its next MFMA, and its SIMD partner is in the same phase (fenced instead: 51.2 -> 53.7 us)
#define OV_MID() do { } while (0)
#define OV_PIPE() do {                                                                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, PW_OV_XTR ? 6 : 4, 0);                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x296, PW_OV_GAP, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x296, PW_OV_GAP, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x296, PW_OV_GAP, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); } while (0)
        mfma_step(kt, 0); OV_MID();
        a_piece(0); if constexpr (STAGE && !(PW_OV_ABL & 2)) store_x_col(nb, SG{}, Q0{});
        if constexpr (EP && !(PW_OV_ABL & 1)) epi_slot(std::integral_constant<int, kt * 4 + 0>{});
        OV_PIPE(); OV_SB();
        mfma_step(kt, 1); OV_MID();
        a_piece(1); if constexpr (STAGE && !(PW_OV_ABL & 2)) store_x_col(nb, SG{}, Q1{});
        if constexpr (EP && !(PW_OV_ABL & 1)) epi_slot(std::integral_constant<int, kt * 4 + 1>{});
        OV_PIPE(); OV_SB();
        mfma_step(kt, 2); OV_MID();
        a_piece(2); if constexpr (STAGE && !(PW_OV_ABL & 2)) store_x_col(nb, SG{}, Q2{});
        if constexpr (EP && !(PW_OV_ABL & 1)) epi_slot(std::integral_constant<int, kt * 4 + 2>{});
        OV_PIPE(); OV_SB();
        mfma_step(kt, 3); OV_MID();
        a_piece(3);
        if constexpr (STAGE && !(PW_OV_ABL & 2)) {
            store_x_col(nb, SG{}, Q3{});
#pragma unroll
            for (int e = 0; e < 4; ++e) load_x(kt + 3, SG{}, e);
        }
        if constexpr (EP && !(PW_OV_ABL & 1)) epi_slot(std::integral_constant<int, kt * 4 + 3>{});
        OV_PIPE(); OV_SB();
#undef OV_MID
#undef OV_PIPE
        if constexpr (!(PW_OV_ABL & 8)) { if constexpr (PW_OV_RAWBAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads(); }
    };
    unsigned touch = 0, touch2 = 0;
    auto kloop = [&](auto epc) {
        // PW_OV_TOUCH: the epilogue of THIS tile (which rides on the next tile's k-loop, ~6 us from now) reads its R tile (MASK) and
        // writes its Y tile: one dword per 128-byte line of those tiles is requested now (one load per wave and tensor: 64 lanes x
        // one line, 8 waves = the tile's 512 lines), so that the lines are in the XCD's L2 when the epilogue's loads / stores arrive
        // -- an L2 hit instead of an HBM round trip that the in-order vmcnt counter would make every younger load of the wave wait for.
        if constexpr (PW_OV_TOUCH != 0) {
            const int tvo = ((b * M + mt * BM + 32 * wave + (lane >> 1)) * P16 + tt * PW_BN + (lane & 1) * 64) * 2;
            if constexpr (MASK && (PW_OV_TOUCH & 1)) touch = __builtin_amdgcn_raw_buffer_load_b32(rR, tvo, 0, 0);
            if constexpr ((PW_OV_TOUCH & 2) != 0) touch2 = __builtin_amdgcn_raw_buffer_load_b32(rY, tvo, 0, 0);
        }
        // k-tile 0 -> LDS slot 0 (registers of stage 0), k-tile 2 requested
#pragma unroll
        for (int i = 0; i < 4; ++i) { *reinterpret_cast<u32x4*>(As + ldsA[i]) = ra[0][i]; load_a(AL - 1, S0{}, i); }
        store_x_col(0, S0{}, Q0{}); store_x_col(0, S0{}, Q1{}); store_x_col(0, S0{}, Q2{}); store_x_col(0, S0{}, Q3{});
#pragma unroll
        for (int e = 0; e < 4; ++e) load_x(2, S0{}, e);
        OV_SB();
        if constexpr (PW_OV_RAWBAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
        ktile(std::integral_constant<int, 0>{}, epc); ktile(std::integral_constant<int, 1>{}, epc);
        ktile(std::integral_constant<int, 2>{}, epc); ktile(std::integral_constant<int, 3>{}, epc);
        if constexpr (NK == 8) {
            ktile(std::integral_constant<int, 4>{}, epc); ktile(std::integral_constant<int, 5>{}, epc);
            ktile(std::integral_constant<int, 6>{}, epc); ktile(std::integral_constant<int, 7>{}, epc);
        }
        // (the touch loads are long done: consumed here, where the compiler's wait for them is vacuous, so that nothing else in
        //  the loop waits for them)
        if constexpr (PW_OV_TOUCH != 0) { const unsigned t1_ = touch, t2_ = touch2; asm volatile("" :: "v"(t1_), "v"(t2_)); }
    };
    // (loads past the last k-tile of a k-loop aim at the next tile: after it, stage 0 holds that tile's k-tile 0, stage 1 its k-tile 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(0, S0{}, i);
#pragma unroll
    for (int e = 0; e < 4; ++e) load_x(0, S0{}, e);
    OV_SB();
    if constexpr (!A1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_a(1, S1{}, i);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) load_x(1, S1{}, e);
    OV_SB();
    kloop(std::false_type{});
    for (;;) {
        // the tile just finished becomes "previous": its accumulators are set aside, its epilogue rides on the next k-loop
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) accp[i][j] = acc[i][j];
        pb = b; ptt = tt; pmt = mt;
        ep_tl = p.T - (tt * PW_BN + wn * 64 + (lane & 7) * 4);
        ep_vo = ((b * M + mt * BM + wm * 64 + (lane >> 3)) * P16 + tt * PW_BN + wn * 64 + (lane & 7) * 4) * 2;
        ep_r_req(std::integral_constant<int, 0>{});
        ep_coef_tile();
        const int vn = v + (int)gridDim.x;
        if (vn >= total) break;
        v = vn;
        pw_work_v(p, v, total, b, tt, mt);
        voA = voAn; voX = voXn;
#if PW_OV_XTR
        t_lim = p.T - tt * PW_BN - x_ch * 8;
#else
        t_lim = p.T - tt * PW_BN - b_tq;
#endif
        aim_next();
        zero_acc();
        kloop(std::true_type{});
    }
    // the last tile's epilogue on its own
    if constexpr (PW_OV_ABL & 1) {                       // timing-only: keep the accumulators alive, store nothing
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += accp[i][j][r];
        if (sum == 12345.678f) p.stats[0] = sum;
        return;
    }
#define OV_D(u_) epi_unit(std::integral_constant<int, u_>{});
    OV_D(0) OV_D(1) OV_D(2) OV_D(3) OV_D(4) OV_D(5) OV_D(6) OV_D(7) OV_D(8) OV_D(9) OV_D(10) OV_D(11)
    OV_D(12) OV_D(13) OV_D(14) OV_D(15) OV_D(16) OV_D(17) OV_D(18) OV_D(19) OV_D(20) OV_D(21) OV_D(22) OV_D(23) OV_D(24)
#undef OV_D
    epi_xchg();
    __syncthreads();
    epi_final();
#undef OV_SB
}

// ---------------------------------------------------------------------------------------------
// Short-K GEMM with SPLIT ROLES (round 5; plain bf16 X, bf16 Y, whole 256-row tiles, K = 64 NK).
// What the overlapped-epilogue kernel above could not hide (profiles/r05_ov_ablation.txt): its k-loop alone runs the wide expand
// forward GEMM in 34 us, the epilogue's VALU / LDS work adds 6, but the epilogue's MEMORY instructions add 16 (Y stores) and, in the
// mask form, another 18 (R loads) -- although they ride on the next tile's k-loop.  The reason is the in-order vmcnt counter: a wave
// that issues an HBM-latency store or load cannot consume any YOUNGER staging load until it has completed, so every wave that both
// stages operands (latency-critical, one k-tile ahead) and runs epilogue memory traffic stalls on the latter.  Here no wave does
// both.  Twelve waves (three per SIMD, 168 registers each).  Waves 0-7 (two per SIMD: the MATRIX waves) stage A pieces and X chunks (as
// loaded, 16 bytes at a time), read fragments (A by ds_read_b128, B by ds_read_b64_tr_b16 from the [k][t] image) and run the MFMAs of
// a 64 x 64 block each (one matrix wave per SIMD with a 64 x 128 block needs 128 accumulator registers and spills); at the end of a
// tile they put the accumulators, rounded to bf16 (what Y stores), into a 64 KB LDS tile and go straight on to the next tile.  The MFMAs
// run with their operands SWAPPED (acc = X^T-fragment x A-fragment: the 32 x 32 x 16 operand layouts are symmetric, so this is only
// the argument order): a lane then holds four CONSECUTIVE t of one output row per accumulator quad, i.e. 8 packed bytes of Y, and the
// hand-over is 16 ds_write_b64 per lane instead of scalar stores through a transposing image.  Waves 8-11 (one per SIMD:
// the EPILOGUE waves) turn the PREVIOUS tile's LDS tile into the output -- two 256-byte rows per pass: ReLU6 mask from R (the whole R
// sub-tile requested one TILE ahead into 64 registers), BatchNorm partial sums (a wave owns whole rows: no cross-wave step),
// 256-byte row stores -- and never touch the operands.  The workgroup's barriers (one per k-tile + one at the hand-over) are the only
// coupling: an epilogue wave processes 32 / NK passes between two of them.  LDS: 96 KB of operand stages + the 64 KB tile = all 160 KB.
// The partial sums are taken from the bf16 values Y stores (the fp32 tile would be 128 KB): BatchNorm statistics of the rounded
// tensor, which is what the reference's autocast run computes them from (its conv output IS the bf16 tensor).
// The workgroup barrier WITHOUT the vmcnt(0) that __syncthreads() puts in front of it (a workgroup-scope fence drains the wave's
// vector-memory queue): here that would make a matrix wave wait for its two-k-tiles-ahead prefetch and an epilogue wave for its Y
// stores' HBM acknowledgements at every one of a tile's nine barriers (measured: the epilogue waves then take 11 us a tile).  What a
// barrier has to order in this kernel is LDS traffic only: the wave's own ds_write / ds_read are complete (lgkmcnt(0)) when it arrives.
#if PW_SL_RAWBAR
#define SL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define SL_BARRIER() __syncthreads()
#endif
template <int EPI, int NK>
__global__ __launch_bounds__(768) void pw_gemm_bf16_sl_kernel(PwParams p) {
    static_assert(EPI == PW_EPI_STATS || EPI == PW_EPI_MASK_STATS, "expand forward on the bf16 shadow / project backward-data");
    static_assert(NK >= 2 && NK <= 32 && (32 % NK) == 0, "32 row passes are spread over the NK k-tile intervals");
    constexpr bool MASK = EPI == PW_EPI_MASK_STATS;
    constexpr int BM = 256;
    constexpr int A_BYTES = BM * 128, X_BYTES = 64 * 256, OUT_BYTES = BM * PW_BN * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 2 * X_BYTES + OUT_BYTES];     // 96 KB + 64 KB = all of the CU's LDS
    // (a DS instruction's immediate offset is 16 bits and the image is 160 KB: the region bases B_OFF / O_OFF are folded into the
    //  per-lane address registers, so that what is left of every address is a constant below 64 KB -- with the bases left in the
    //  constants hipcc materialises ~60 address registers and spills)
    unsigned char* As = smem;
    constexpr int B_OFF = 2 * A_BYTES, O_OFF = 2 * A_BYTES + 2 * X_BYTES;     // X stages; the hand-over tile [256 rows][32 slots of 8 bytes], slot = (t / 4) ^ (row & 31)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, K = p.K;
    const int P16 = pw_pitch16(p.T, p.B);
    const int total = p.n_mtiles * p.n_ttiles * p.B;
    const int ntl = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;       // tiles of this workgroup (>= 1: grid <= total)
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    if (wave < 8 && !(PW_SL_DBG & 1)) {
        // ================================================= matrix waves ==================================================
        const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.Abf, (unsigned)M * K * 2u);
        const __amdgpu_buffer_rsrc_t rX = make_rsrc(p.X, (unsigned)p.B * K * P16 * 2u);
        int v = blockIdx.x;
        int b, tt, mt;
        pw_work_v(p, v, total, b, tt, mt);
        // A: 256 rows x 8 chunks of 16 bytes per k-tile = 4 pieces per lane (rows arow + 64 i, chunk tid & 7; the image's XOR term
        // (row >> 1) & 7 is the same for all four), X: 64 k-rows x 16 chunks = 2 per lane (rows x_row + 32 e, chunk tid & 15)
        const int arow = tid >> 3;
        const int ldsA0 = bf_off(arow, tid & 7);
        const int stepA = 128 * K, stepX = P16 * 2;
        const int x_row = tid >> 4, x_ch = tid & 15;
        const int ldsX = B_OFF + 256 * x_row + 16 * (x_ch ^ (((x_row & 3) << 2) | ((x_row >> 2) & 3)));
        auto vo_a = [&](int mt_) { return ((mt_ * BM + arow) * K + (tid & 7) * 8) * 2; };
        auto vo_x = [&](int b_, int tt_) { return ((b_ * K + x_row) * P16 + tt_ * PW_BN + x_ch * 8) * 2; };
        int voA = vo_a(mt), voX = vo_x(b, tt), voAn = voA, voXn = voX;
        int t_lim = p.T - tt * PW_BN - x_ch * 8;            // this lane's chunk: columns e < t_lim exist
        auto aim_next = [&]() {
            const int vn = v + (int)gridDim.x < total ? v + (int)gridDim.x : v;
            int b_, tt_, mt_;
            pw_work_v(p, vn, total, b_, tt_, mt_);
            voAn = vo_a(mt_); voXn = vo_x(b_, tt_);
        };
        aim_next();
        u32x4 ra[4];
        u32x4 rx[2][2];
        auto load_a = [&](int kt, int i) {                   // kt >= NK: k-tile kt - NK of the next tile
            const bool nx = kt >= NK;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, nx ? voAn : voA, (nx ? kt - NK : kt) * (BF_BK * 2) + i * stepA, 0);
        };
        auto load_x = [&](int kt, auto stg, int e) {
            constexpr int SG = decltype(stg)::value;
            const bool nx = kt >= NK;
            rx[SG][e] = __builtin_amdgcn_raw_buffer_load_b128(rX, nx ? voXn : voX, ((nx ? kt - NK : kt) * BF_BK + 32 * e) * stepX, 0);
        };
        auto store_a = [&](int buf, int i) { *reinterpret_cast<u32x4*>(As + buf * A_BYTES + ldsA0 + i * 8192) = ra[i]; };
        auto store_x = [&](int buf, auto stg, int e) {
            constexpr int SG = decltype(stg)::value;
            // columns past T (partial last t-tile) are staged as ZEROS: their accumulators are exactly 0 (nothing for the statistics)
            u32x4 o = rx[SG][e];
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] &= (2 * d < t_lim ? 0xffffu : 0u) | (2 * d + 1 < t_lim ? 0xffff0000u : 0u);
            *reinterpret_cast<u32x4*>(smem + ldsX + (buf * X_BYTES + e * 8192)) = o;
        };
        const int wm = wave >> 1, wn = wave & 1;             // rows wm * 64 ... + 63, columns wn * 64 ... + 63
        f32x16 acc[2][2];
        auto zero_acc = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        };
        zero_acc();
        const int lr = lane & 31, lh = lane >> 5;
        const int sw = (lr >> 1) & 7;
        const int rdA0 = (wm * 64 + lr) * 128;
        // B fragment of column block j, k-step ks (see the transposed-read form of the kernel above): address register per (j, h)
        typedef short sl_s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) sl_s16x4 sl_lds_s16x4;
        int trB[2][2];
        {
            const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 8 * (g >> 1) + 4 * h + q4;
                    const int ch = ((wn * 64 + j * 32 + 16 * (g & 1)) >> 3) + (pp >> 1);
                    trB[j][h] = B_OFF + 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (pp & 1);
                }
        }
        auto mfma_step = [&](int kt, int ks) {
            const unsigned char* Ab = As + (kt & 1) * A_BYTES;
            const unsigned char* Bb = smem + (kt & 1) * X_BYTES;
            const int co = ((ks * 2 + lh) ^ sw) << 4;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + co);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + 32 * 128 + co);
            bf16x8 bq[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const sl_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sl_lds_s16x4*)(Bb + trB[j][0] + ks * 4096));
                const sl_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sl_lds_s16x4*)(Bb + trB[j][1] + ks * 4096));
                bq[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // (operands swapped: element r of lane l = output row i * 32 + (l & 31), column j * 32 + (r & 3) + 8 (r >> 2) + 4 (l >> 5))
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[j], a0, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[j], a1, acc[1][j], 0, 0, 0);
            }
        };
        // k-tile kt (static): fragments from slot kt & 1; k-tile kt + 1 (ra, rx[(kt + 1) & 1]) -> LDS slot (kt + 1) & 1 in four slices
        // behind the k-steps' MFMAs, k-tiles kt + 2 (A) / kt + 3 (X) requested (none of it in the last k-tile)
        auto ktile = [&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            constexpr bool STAGE = kt < NK - 1;
            using SG = std::integral_constant<int, (kt + 1) & 1>;
            constexpr int nb = (kt + 1) & 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mfma_step(kt, ks);
                if constexpr (STAGE) {
                    store_a(nb, ks); load_a(kt + 2, ks);
                    if ((ks & 1) == 0) store_x(nb, SG{}, ks >> 1);
                    if (ks == 3) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) load_x(kt + 3, SG{}, e);
                    }
                }
                if constexpr (!(PW_SL_DBG & 4)) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2b6, 2, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            SL_BARRIER();
        };
        auto kloop = [&]() {
            ktile(std::integral_constant<int, 0>{}); ktile(std::integral_constant<int, 1>{});
            if constexpr (NK > 2) { ktile(std::integral_constant<int, 2>{}); ktile(std::integral_constant<int, 3>{}); }
            if constexpr (NK > 4) { ktile(std::integral_constant<int, 4>{}); ktile(std::integral_constant<int, 5>{});
                                    ktile(std::integral_constant<int, 6>{}); ktile(std::integral_constant<int, 7>{}); }
            static_assert(NK == 2 || NK == 4 || NK == 8, "k-loop instantiated for K = 128, 256, 512");
        };
        // the first tile's k-tiles 0 and 1, then k-tile 0 -> LDS slot 0
#pragma unroll
        for (int i = 0; i < 4; ++i) load_a(0, i);
#pragma unroll
        for (int e = 0; e < 2; ++e) load_x(0, S0{}, e);
#pragma unroll
        for (int e = 0; e < 2; ++e) load_x(1, S1{}, e);
        for (int it = 0;; ++it) {
            // k-tile 0 of this tile -> LDS slot 0 (the registers hold it), k-tiles 1 (A) and 2 (X) requested
#pragma unroll
            for (int i = 0; i < 4; ++i) { store_a(0, i); load_a(1, i); }
#pragma unroll
            for (int e = 0; e < 2; ++e) { store_x(0, S0{}, e); load_x(2, S0{}, e); }
            __builtin_amdgcn_sched_barrier(0);
            SL_BARRIER();                               // it == 0: the start barrier; later: the hand-over of the previous tile's accumulators
            kloop();
            // the finished tile -> the LDS tile (the epilogue waves are done with the previous one: they passed the last k-tile's barrier)
            int out_row = O_OFF + (wm * 64 + lr) * 256, out_slot = (wn * 16 + lh) ^ lr;
            asm volatile("" : "+v"(out_row), "+v"(out_slot));
            if constexpr (PW_SL_DBG & 16) {                  // timing-only: no hand-over (keep the accumulators alive)
                float sum_ = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum_ += acc[i][j][r];
                if (sum_ == 12345.678f) p.stats[0] = sum_;
            } else
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const u32x2 o2 = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                        // slot (wn 16 + j 8 + 2 g + lh) ^ lr = out_slot ^ (j 8 + 2 g): one XOR + one shift-add per store, from a base that
                        // is re-derived per tile (hoisted, the sixteen addresses would live in registers across the k-loop)
                        *reinterpret_cast<u32x2*>(smem + out_row + i * (32 * 256) + ((out_slot ^ (j * 8 + 2 * g)) << 3)) = o2;
                    }
            if (it + 1 >= ntl) break;
            zero_acc();
            v += (int)gridDim.x;
            pw_work_v(p, v, total, b, tt, mt);
            voA = voAn; voX = voXn;
            t_lim = p.T - tt * PW_BN - x_ch * 8;
            aim_next();
        }
        SL_BARRIER();                                   // the last hand-over
        return;
    }
    // =================================================== epilogue waves ====================================================
    if constexpr (!(PW_SL_DBG & 2)) {
        // The epilogue waves are the workgroup's youngest: with issue arbitrated by priority, then age, they only get the slots the
        // matrix waves leave (measured: 11 us per tile for ~1200 instructions).  Their stream is short; at a raised priority it
        // costs the matrix waves little and is no longer what the tile time waits for.
        if constexpr (PW_SL_PRIO > 0) __builtin_amdgcn_s_setprio(PW_SL_PRIO);
        const int sw4 = wave - 8;                            // rows sw4 * 64 ... + 63 of a tile, two per pass
        const int col4 = (lane & 31) * 4, half = lane >> 5;
        const int in_row0 = O_OFF + (sw4 * 64 + half) * 256, in_slot0 = (lane & 31) ^ half;
        const __amdgpu_buffer_rsrc_t rS = make_rsrc(p.stats, (unsigned)((size_t)p.B * p.n_ttiles * M * 8u));
        const __amdgpu_buffer_rsrc_t rY = make_rsrc(p.Y, (unsigned)p.B * M * P16 * 2u);
        const __amdgpu_buffer_rsrc_t rR = make_rsrc(MASK ? p.R : p.X, MASK ? (unsigned)p.B * M * P16 * 2u : 0u);
        const __amdgpu_buffer_rsrc_t rEa = make_rsrc(MASK ? p.ea : p.X, MASK ? (unsigned)M * 4u : 0u);
        const __amdgpu_buffer_rsrc_t rEb = make_rsrc(MASK ? p.eb : p.X, MASK ? (unsigned)M * 4u : 0u);
        constexpr int PPC = 32 / NK;                         // passes between two barriers
        const int stepP = 2 * P16 * 2;                       // bytes from one pass to the next (two rows)
        auto tile_vo = [&](int it_, int& b_, int& tt_, int& mt_) {
            const int v_ = (int)blockIdx.x + (it_ < ntl ? it_ : ntl - 1) * (int)gridDim.x;
            pw_work_v(p, v_, total, b_, tt_, mt_);
            return ((b_ * M + mt_ * BM + sw4 * 64 + half) * P16 + tt_ * PW_BN + col4) * 2;
        };
        u32x2 rr[MASK ? 16 : 1];                              // R pieces of the next 16 passes (half a tile, ~2.5 us, ahead)
        float cea = 0.f, ceb = 0.f, cean = 0.f, cebn = 0.f; // lane l: (ea, eb) of row sw4 * 64 + l of the tile in hand / the next one
        int nb_, ntt_, nmt_;
        int nvo = tile_vo(0, nb_, ntt_, nmt_);
        if constexpr (MASK) {
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) rr[ps] = __builtin_amdgcn_raw_buffer_load_b64(rR, nvo, ps * stepP, 0);
            cean = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rEa, (nmt_ * BM + sw4 * 64 + lane) * 4, 0, 0));
            cebn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rEb, (nmt_ * BM + sw4 * 64 + lane) * 4, 0, 0));
        }
        // one pass: rows sw4 * 64 + 2 ps + half of the LDS tile -> Y (and the row's partial sums); MASK: R piece ps is consumed and, when
        // MORE, re-requested for the tile after this one
        int pb = 0, ptt = 0, pmt = 0, pvo = 0, ptl = 0, psvo = 0;      // the tile in hand
        auto pass = [&](auto psc, auto morec) {
            constexpr int ps = decltype(psc)::value;
            constexpr bool MORE = decltype(morec)::value;
            if constexpr (PW_SL_DBG & 8) return;             // timing-only: the epilogue waves only keep the barriers
            // row sw4 64 + 2 ps + half, slot (lane & 31) ^ (row & 31) = in_slot ^ (2 ps & 31) with in_slot = (lane & 31) ^ half
            int in_row = in_row0, in_slot = in_slot0;
            asm volatile("" : "+v"(in_row), "+v"(in_slot));     // (not hoisted: 32 address registers otherwise)
            u32x2 o2 = *reinterpret_cast<const u32x2*>(smem + in_row + ps * 512 + ((in_slot ^ ((2 * ps) & 31)) << 3));
            float s0 = 0.f, s1 = 0.f;
            if constexpr (MASK) {
                const int e0 = __builtin_amdgcn_readlane(__builtin_bit_cast(int, cea), 2 * ps), e1 = __builtin_amdgcn_readlane(__builtin_bit_cast(int, cea), 2 * ps + 1);
                const int f0 = __builtin_amdgcn_readlane(__builtin_bit_cast(int, ceb), 2 * ps), f1 = __builtin_amdgcn_readlane(__builtin_bit_cast(int, ceb), 2 * ps + 1);
                const float ea = __builtin_bit_cast(float, half ? e1 : e0), eb = __builtin_bit_cast(float, half ? f1 : f0);
                const u32x2 r2 = rr[ps & 15];
                // piece ps + 16: of this tile (pvo) for the first half, of the tile after it (nvo; only when there is one) for the second
                if constexpr (ps < 16) rr[ps & 15] = __builtin_amdgcn_raw_buffer_load_b64(rR, pvo, (ps + 16) * stepP, 0);
                else if constexpr (MORE) rr[ps & 15] = __builtin_amdgcn_raw_buffer_load_b64(rR, nvo, (ps - 16) * stepP, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (columns past T of a partial last t-tile: accumulators exactly 0 -- zero X --, R's row padding is whatever the
                    //  allocation held: the median maps a NaN / Inf there to a finite value, so 0 * R stays 0)
                    const float r = __builtin_amdgcn_fmed3f(pw_bf16_at(r2, e), -3.3895314e38f, 3.3895314e38f);
                    const float pre = fmaf(r, ea, eb);
                    const bool keep = pre > 0.f && pre < 6.f;
                    const float x = keep ? pw_bf16_at(o2, e) : 0.f;
                    if (!keep) o2[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
                    s0 += x; s1 = fmaf(x, r, s1);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float x = pw_bf16_at(o2, e); s0 += x; s1 = fmaf(x, x, s1); }
            }
            // (a straddling lane's 8 bytes stay inside the pitched row; a lane wholly past T aims outside the descriptor: dropped)
            if constexpr (!(PW_SL_DBG & 32)) __builtin_amdgcn_raw_buffer_store_b64(o2, rY, ptl > 0 ? pvo : 0x7ffffff0, ps * stepP, 0);
            else if (o2[0] == 0x12345678u) p.stats[1] = 1.f;
            if constexpr (PW_SL_DBG & 128) { if (s0 + s1 == 12345.678f) p.stats[2] = 1.f; return; }
            s0 = half_wave_sum_dpp(s0);
            s1 = half_wave_sum_dpp(s1);
            // lanes 31 and 63 hold the two rows' sums: 8 bytes each at stats[part][row][0..1]
            const u32x2 st2 = {__builtin_bit_cast(unsigned, s0), __builtin_bit_cast(unsigned, s1)};
            if constexpr (!(PW_SL_DBG & 64)) __builtin_amdgcn_raw_buffer_store_b64(st2, rS, (lane & 31) == 31 ? psvo : 0x7ffffff0, ps * 16, 0);
            else if (st2[0] == 0x12345678u) p.stats[1] = 1.f;
        };
        auto chunk = [&](auto cc, auto morec) {
            constexpr int c = decltype(cc)::value;
#define SL_P(k_) if constexpr ((k_) < PPC) pass(std::integral_constant<int, c * PPC + ((k_) < PPC ? (k_) : 0)>{}, morec);
            SL_P(0) SL_P(1) SL_P(2) SL_P(3) SL_P(4) SL_P(5) SL_P(6) SL_P(7) SL_P(8) SL_P(9) SL_P(10) SL_P(11) SL_P(12) SL_P(13) SL_P(14) SL_P(15)
#undef SL_P
        };
        auto begin_tile = [&](int it_) {                   // tile it_ becomes the tile in hand; the one after it the prefetch target
            pvo = tile_vo(it_, pb, ptt, pmt);
            ptl = p.T - (ptt * PW_BN + col4);
            psvo = (((pb * p.n_ttiles + ptt) * M + pmt * BM + sw4 * 64 + half) * 2) * 4;
            cea = cean; ceb = cebn;
            nvo = tile_vo(it_ + 1, nb_, ntt_, nmt_);
            if constexpr (MASK) {
                cean = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rEa, (nmt_ * BM + sw4 * 64 + lane) * 4, 0, 0));
                cebn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rEb, (nmt_ * BM + sw4 * 64 + lane) * 4, 0, 0));
            }
        };
        auto process = [&](auto morec, auto barc) {         // NK chunks, a workgroup barrier behind each when BAR
            constexpr bool BAR = decltype(barc)::value;
#define SL_C(c_) if constexpr ((c_) < NK) { chunk(std::integral_constant<int, ((c_) < NK ? (c_) : 0)>{}, morec); if constexpr (BAR) SL_BARRIER(); }
            SL_C(0) SL_C(1) SL_C(2) SL_C(3) SL_C(4) SL_C(5) SL_C(6) SL_C(7)
#undef SL_C
        };
        SL_BARRIER();                                   // the start barrier
        // tile 0's k-loop: nothing to do yet
#pragma unroll
        for (int c = 0; c < NK; ++c) SL_BARRIER();
        SL_BARRIER();                                   // hand-over of tile 0
        for (int it = 1; it < ntl; ++it) {
            begin_tile(it - 1);
            process(std::true_type{}, std::true_type{});
            SL_BARRIER();                               // hand-over of tile it
        }
        begin_tile(ntl - 1);
        process(std::false_type{}, std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised backward-weight (round 3): pw_gemm_bf16_ws_kernel's division of labour on pw_wgrad_bf16_wide_kernel's tile.
// Twelve waves: waves 0-7 (64 x 64 each of the GR x XR tile) copy the PLAIN operand's 256 x 64 bf16 tile to LDS as loaded (4 pieces
// per lane, two register stages, one ds_write_b128 + the request two steps ahead behind each k-step's MFMAs) and run the MFMAs;
// waves 8-11 stage the TRANSFORMED operand's 128 x 64 tile (BatchNorm-backward affine of two bf16 tensors, or BatchNorm + ReLU6):
// NSQ register stages of loads, transform, bf16 pack, 4 ds_write_b128 per lane and step.  One barrier per step.  Full tiles only
// (M % GR == 0, K % XR == 0); TAIL: T % 64 != 0, contraction indices past T are zeroed in BOTH operands (masks on the plain one).
template <int GM, int XM, bool TAIL, int IO, int GR, int XR, int NSW>
__global__ __launch_bounds__(512 + 64 * NSW) void pw_wgrad_bf16_ws_kernel(WgParams p) {
    // (both plain -- G = the finished gradient da1, round 5 --: G takes the 128-row side and its "transform" is a copy)
    static_assert(!(GM != PW_X_NONE && XM != PW_X_NONE), "at most one transformed operand (128 rows); the other one plain (256 rows)");
    static_assert((IO & WG_IO_G) && (IO & WG_IO_X) && (GM != PW_X_AFFINE2 || (IO & WG_IO_G2)), "bf16-stored operands only");
    constexpr bool PG = GM == PW_X_NONE && XM != PW_X_NONE; // the plain 256-row operand is G
    static_assert((PG ? GR : XR) == 256 && (PG ? XR : GR) == 128, "tile shape");
    constexpr int QM = PG ? XM : GM;                        // the transform
    constexpr int NX = XR / 64;
    static_assert(NSW == 4 || NSW == 8, "staging waves: one or two per SIMD");
    constexpr int NSQ = (QM == PW_X_AFFINE2 && NSW == 4) ? 3 : 4;    // register stages of the transformed operand
    constexpr int NQP = 16 / NSW;                           // 16-byte pieces per staging lane and step
    constexpr int QRS = 8 * NSW;                            // rows between a lane's pieces
    __shared__ __attribute__((aligned(16))) unsigned char As[2][GR * 128];   // [m][t] bf16
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][XR * 128];   // [k][t] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int s, mt, ktile;
    wg_work(p, s, mt, ktile);
    const int m0 = mt * GR, n0 = ktile * XR;
    const int M = p.M, K = p.K, T = p.T;
    const int P16 = pw_pitch16(T, p.B);
    const int nt = (T + BF_BK - 1) / BF_BK;
    const WgSpan sp = wg_span(p, s, nt);
    const int nsteps = sp.nb * sp.ntl, b_lo = sp.b_lo;
    // step -> (batch element, t offset); steps past the end are clamped to the last (an unconditional, redundant load: a
    // conditional one would make hipcc wait for the YOUNGER stage at the join)
    auto step_bt = [&](int step, int& b, int& t0) {
        const int q = min(step, nsteps - 1);
        (void)q;
        const int bi = q / sp.ntl;
        b = b_lo + bi;
        t0 = (sp.t_first + q - bi * sp.ntl) * BF_BK;
    };
    // contraction indices t0 + 8 ch + e >= T of a 16-byte piece -> zero
    auto tail_mask = [&](u32x4 v, int tb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] &= (tb + 2 * j < T ? 0xffffu : 0u) | (tb + 2 * j + 1 < T ? 0xffff0000u : 0u);
        return v;
    };
    if (nsteps == 0) {                                      // (no work for this split: the partial tile is zeros)
        if (wave < 8) {
            const int wm = wave / NX, wn = wave % NX, col = lane & 31, half = lane >> 5;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, k = n0 + wn * 64 + j * 32 + col;
                        p.partial[((size_t)s * M + m) * K + k] = 0.f;
                    }
        }
        return;
    }

    if (wave >= 8) {
        // ------------------------------------------------ staging waves: the transformed operand ------------------------------------------------
        const int pt = tid - 512;
        const int qrow = pt >> 3, ch = pt & 7;              // piece i: row qrow + QRS i (same swizzle key), chunk ch
        const int row0 = PG ? n0 : m0, rows = PG ? K : M;
        const float* q1 = PG ? p.X : p.G;
        const float* q2 = PG ? p.X : p.G2;
        const float* pa = PG ? p.xa : p.ga;
        const float* pb = PG ? p.xb : p.gb;
        unsigned char (*Qs)[128 * 128] = PG ? reinterpret_cast<unsigned char (*)[128 * 128]>(Bs) : reinterpret_cast<unsigned char (*)[128 * 128]>(As);
        float ca[NQP], cb[NQP], cc[NQP];
#pragma unroll
        for (int i = 0; i < NQP; ++i) {
            const int r = row0 + qrow + QRS * i;
            ca[i] = QM != PW_X_NONE ? pa[r] : 1.f; cb[i] = QM != PW_X_NONE ? pb[r] : 0.f;
            cc[i] = QM == PW_X_AFFINE2 ? p.gc[r] : 0.f;
        }
        const int voQ = ((row0 + qrow) * P16 + ch * 8) * 2;
        const int stepQ = QRS * P16 * 2;
        const int ldsQ = bf_off(qrow, ch);
        u32x4 rq[NSQ][NQP], rq2[NSQ][QM == PW_X_AFFINE2 ? NQP : 1];
#define WS_SB() __builtin_amdgcn_sched_barrier(0)
        auto load_q = [&](int step, auto stg, int i) {
            constexpr int SG = decltype(stg)::value;
            int b, t0;
            step_bt((PW_WG_ABL & 2) ? 0 : step, b, t0);           // (timing-only bit 1: every step re-reads the first tile)
            const __amdgpu_buffer_rsrc_t r1 = make_rsrc(reinterpret_cast<const u16*>(q1) + (size_t)b * rows * P16, (unsigned)rows * P16 * 2u);
            rq[SG][i] = __builtin_amdgcn_raw_buffer_load_b128(r1, voQ, t0 * 2 + i * stepQ, 0);
            if constexpr (QM == PW_X_AFFINE2) {
                const __amdgpu_buffer_rsrc_t r2 = make_rsrc(reinterpret_cast<const u16*>(q2) + (size_t)b * rows * P16, (unsigned)rows * P16 * 2u);
                rq2[SG][i] = __builtin_amdgcn_raw_buffer_load_b128(r2, voQ, t0 * 2 + i * stepQ, 0);
            }
        };
        // step st (registers of stage SG) -> LDS slot st & 1; step st + NSQ requested into the same registers
        auto stage = [&](int st, auto stg) {
            constexpr int SG = decltype(stg)::value;
            unsigned char* Qd = Qs[st & 1] + ldsQ;
            int tb = 0;
            bool tail = false;
            if constexpr (TAIL) {
                int b, t0;
                step_bt(st, b, t0);
                tail = t0 + BF_BK > T;
                tb = t0 + ch * 8;
            }
            u32x4 o[NQP];
            if constexpr (PW_WG_ABL & 8) {                  // timing-only: loads kept, nothing else
#pragma unroll
                for (int i = 0; i < NQP; ++i) {
                    asm volatile("" :: "v"(rq[SG][i]));
                    if constexpr (QM == PW_X_AFFINE2) asm volatile("" :: "v"(rq2[SG][i]));
                    load_q(st + NSQ, stg, i);
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < NQP; ++i) {
                if constexpr (PW_WG_ABL & 4) {              // timing-only: raw copy, no transform
                    o[i] = rq[SG][i];
                    if constexpr (QM == PW_X_AFFINE2) o[i][0] ^= rq2[SG][i][0];
                    continue;
                }
                if constexpr (QM == PW_X_NONE) {            // plain operand on the staged side: copied as loaded
                    o[i] = rq[SG][i];
                    if constexpr (TAIL) { if (tail) o[i] = tail_mask(o[i], tb); }
                    continue;
                }
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = pw_bf16_at(rq[SG][i], e);
                    if constexpr (QM == PW_X_AFFINE2) v[e] = fmaf(x, ca[i], fmaf(pw_bf16_at(rq2[SG][i], e), cb[i], cc[i]));
                    else v[e] = relu6f(fmaf(x, ca[i], cb[i]));
                }
                o[i][0] = pack_bf16(v[0], v[1]); o[i][1] = pack_bf16(v[2], v[3]); o[i][2] = pack_bf16(v[4], v[5]); o[i][3] = pack_bf16(v[6], v[7]);
                if constexpr (TAIL) { if (tail) o[i] = tail_mask(o[i], tb); }
            }
            WS_SB();
#pragma unroll
            for (int i = 0; i < NQP; ++i) {                 // stores and the next requests interleaved
                *reinterpret_cast<u32x4*>(Qd + i * (QRS * 128)) = o[i];
                load_q(st + NSQ, stg, i);
                WS_SB();
            }
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
#pragma unroll
        for (int i = 0; i < NQP; ++i) load_q(0, S0{}, i);
        WS_SB();
#pragma unroll
        for (int i = 0; i < NQP; ++i) load_q(1, S1{}, i);
        WS_SB();
#pragma unroll
        for (int i = 0; i < NQP; ++i) load_q(2, S2{}, i);
        WS_SB();
        if constexpr (NSQ == 4) {
#pragma unroll
            for (int i = 0; i < NQP; ++i) load_q(3, S3{}, i);
            WS_SB();
        }
        stage(0, S0{});
        __syncthreads();                                   // step 0 is in LDS
        // NSQ steps per trip with EXITS, not skipped bodies (pw_gemm_bf16_ws_kernel); the stores are unconditional (a step past the
        // last re-stages the last tile into the slot nobody reads)
        // (Round 5: whole trips WITHOUT exits, then the last nsteps % NSQ steps straight-line -- with an exit behind every step the first
        //  step of every trip waited vmcnt(0): see the staging waves of pw_gemm_bf16_ws_kernel)
        int st = 0;
        for (; st + NSQ <= nsteps; st += NSQ) {
            stage(st + 1, S1{});
            __syncthreads();
            stage(st + 2, S2{});
            __syncthreads();
            if constexpr (NSQ == 4) {
                stage(st + 3, S3{});
                __syncthreads();
            }
            stage(st + NSQ, S0{});
            __syncthreads();
        }
        if (st < nsteps) {
            stage(st + 1, S1{});
            __syncthreads();
            if (st + 1 < nsteps) {
                stage(st + 2, S2{});
                __syncthreads();
                if constexpr (NSQ == 4) {
                    if (st + 2 < nsteps) {
                        stage(st + 3, S3{});
                        __syncthreads();
                    }
                }
            }
        }
#undef WS_SB
        return;
    }

    // ---------------------------------------------------- matrix waves: the plain operand + MFMA ----------------------------------------------------
    const int wm = wave / NX, wn = wave % NX;
    const int prow = tid >> 3, pch = tid & 7;               // piece i: row prow + 64 i (same swizzle key), chunk pch
    const int prow0 = PG ? m0 : n0, prows = PG ? M : K;
    const float* pp = PG ? p.G : p.X;
    unsigned char (*Ps)[256 * 128] = PG ? reinterpret_cast<unsigned char (*)[256 * 128]>(As) : reinterpret_cast<unsigned char (*)[256 * 128]>(Bs);
    const int voP = ((prow0 + prow) * P16 + pch * 8) * 2;
    const int stepP = 64 * P16 * 2;
    const int ldsP = bf_off(prow, pch);
    u32x4 rp[2][4];
    auto load_p = [&](int step, auto stg, int i) {
        constexpr int SG = decltype(stg)::value;
        int b, t0;
        step_bt((PW_WG_ABL & 1) ? 0 : step, b, t0);               // (timing-only bit 0)
        const __amdgpu_buffer_rsrc_t r = make_rsrc(reinterpret_cast<const u16*>(pp) + (size_t)b * prows * P16, (unsigned)prows * P16 * 2u);
        rp[SG][i] = __builtin_amdgcn_raw_buffer_load_b128(r, voP, t0 * 2 + i * stepP, 0);
    };
    auto piece_out = [&](int step, u32x4 v) {
        if constexpr (TAIL) {
            int b, t0;
            step_bt(step, b, t0);
            if (t0 + BF_BK > T) v = tail_mask(v, t0 + pch * 8);
        }
        return v;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;
    const int rdA0 = (wm * 64 + lr) * 128, rdB0 = (wn * 64 + lr) * 128;
    // step st: fragments from slot st & 1; the plain tile of step st + 1 (registers of stage SG) -> slot (st + 1) & 1, step st + 3 requested
    auto block = [&](int st, auto stg) {
        constexpr int SG = decltype(stg)::value;
        const unsigned char* Ab = As[st & 1];
        const unsigned char* Bb = Bs[st & 1];
        unsigned char* Pd = Ps[(st + 1) & 1] + ldsP;
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {
            const int co = ((ks * 2 + lh) ^ sw) << 4;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + co);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Ab + rdA0 + 32 * 128 + co);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bb + rdB0 + co);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bb + rdB0 + 32 * 128 + co);
            if constexpr (!(PW_WG_ABL & 16)) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            *reinterpret_cast<u32x4*>(Pd + ks * 8192) = piece_out(st + 1, rp[SG][ks]);
            load_p(st + 3, stg, ks);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int i = 0; i < 4; ++i) load_p(0, S0{}, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_p(1, S1{}, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { *reinterpret_cast<u32x4*>(Ps[0] + ldsP + i * 8192) = piece_out(0, rp[0][i]); load_p(2, S0{}, i); }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                       // step 0 is in LDS
    int st = 0;
    for (; st + 1 < nsteps; st += 2) {                     // pairs, then the odd step
        block(st, S1{});
        __syncthreads();
        block(st + 1, S0{});
        __syncthreads();
    }
    if (st < nsteps) {
        block(st, S1{});
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
                p.partial[((size_t)s * M + m) * K + k] = acc[i][j][r];      // (full tiles only: no bounds tests, no exec-mask branches)
            }
}

// ---------------------------------------------------------------------------------------------
// Dispatch: fast kernels (modes fixed at compile time, aligned K) for the combinations the networks
// use; everything else goes to the generic instantiation (run-time modes, scalar-safe loads).
#define PW_NN_COMBOS(X) X(0, 0) X(0, 1) X(1, 1) X(0, 2) X(0, 3) X(0, 4) X(2, 0) X(2, 5)
#define PW_WG_COMBOS(X) X(0, 0) X(0, 1) X(2, 0)

// fp16 operands (inference): the forward combinations only -- plain / bias (heads, ConvTranspose, dense conv), folded
// BN+ReLU6, folded BN(+residual).  false = this combination has no fp16 instantiation.
bool pw_launch_gemm_f16(const PwParams& p, dim3 grid, hipStream_t st) {
    if (p.x_mode != PW_X_NONE) return false;
    const bool tv = (p.T & 3) == 0, kv = (p.K & 7) == 0;
    const bool full = (p.K & 1) == 0 && (long)(p.K + 64) * p.T * 4 < 0x7fffffffL && (long)(p.M + 128) * p.K * 2 < 0x7fffffffL &&
                      (long)p.B * p.M * p.T * 4 < 0x7fffff00L;
    const bool big = full && p.M >= 256;
    PwParams pb = p;
    pb.n_mtiles = (p.M + 255) / 256;
    const dim3 gridb((unsigned)((long)pb.n_mtiles * p.n_ttiles * p.B));
#define X(EP)                                                                                                           \
    if (p.epi_mode == EP) {                                                                                             \
        if (big) V100_GGL((pw_gemm_bf16_fast_kernel<0, EP, 256, true>), gridb, dim3(512), 0, st, pb);        \
        else if (full) V100_GGL((pw_gemm_bf16_fast_kernel<0, EP, 128, true>), grid, dim3(256), 0, st, p);     \
        else if (tv && kv) V100_GGL((pw_gemm_bf16_kernel<0, EP, true, true, true>), grid, dim3(256), 0, st, p); \
        else V100_GGL((pw_gemm_bf16_kernel<0, EP, false, false, true>), grid, dim3(256), 0, st, p);           \
        return true;                                                                                                    \
    }
    X(0) X(2) X(3)
#undef X
    return false;
}

void pw_launch_gemm_bf16(const PwParams& p_in, dim3 grid_in, hipStream_t st) {
    const PwParams& p = p_in;
    const dim3 grid = grid_in;
    const bool tv = (p.T & 3) == 0, kv = (p.K & 7) == 0;
    // Any M, K, T: rows / columns past the tensor fall outside the buffer descriptors and read as zero; columns
    // t >= T inside a row read the next row's (finite) values, which only reach output columns that are never
    // stored; k >= K rows of X are zero, so whatever A holds there is multiplied by zero.
    const bool full = (p.K & 1) == 0 && (long)(p.K + 64) * p.T * 4 < 0x7fffffffL && (long)(p.M + 128) * p.K * 2 < 0x7fffffffL &&
                      (long)p.B * p.M * p.T * 4 < 0x7fffff00L;
    // 256-row block tile: half the X staging per flop.  (A/B in one process, tools/step_time.py: 128-row tiles everywhere cost
    // +0.2 ms of GEMM time per step; 128-row tiles only for the K <= 512 GEMMs -- two workgroups per CU, epilogue of one
    // over the main loop of the other -- measured equal, so one rule.)
    const bool big = full && p.M >= 256;
    if (full) {
        PwParams pb = p;                       // the 256-row tiling has its own m-tile count / grid
        pb.n_mtiles = (p.M + 255) / 256;
        const dim3 gridb((unsigned)((long)pb.n_mtiles * p.n_ttiles * p.B));
#define X(XM, EP)                                                                                                   \
        if (p.x_mode == XM && p.epi_mode == EP) {                                                                   \
            if (big) V100_GGL((pw_gemm_bf16_fast_kernel<XM, EP, 256>), gridb, dim3(512), 0, st, pb);     \
            else V100_GGL((pw_gemm_bf16_fast_kernel<XM, EP, 128>), grid, dim3(256), 0, st, p);            \
            return;                                                                                                 \
        }
        PW_NN_COMBOS(X)
#undef X
    }
    if (kv) {
#define X(XM, EP)                                                                                                   \
        if (p.x_mode == XM && p.epi_mode == EP) {                                                                   \
            if (tv) V100_GGL((pw_gemm_bf16_kernel<XM, EP, true, true>), grid, dim3(256), 0, st, p);          \
            else V100_GGL((pw_gemm_bf16_kernel<XM, EP, false, true>), grid, dim3(256), 0, st, p);            \
            return;                                                                                                 \
        }
        PW_NN_COMBOS(X)
#undef X
    }
    if (tv && kv) V100_GGL((pw_gemm_bf16_kernel<-1, -1, true, true>), grid, dim3(256), 0, st, p);
    else V100_GGL((pw_gemm_bf16_kernel<-1, -1, false, false>), grid, dim3(256), 0, st, p);
}

void pw_launch_wgrad_bf16(const WgParams& p, dim3 grid, hipStream_t st) {
    const bool tv = (p.T & 3) == 0;
    const bool full = (long)(p.M + 128) * p.T * 4 < 0x7fffffffL && (long)(p.K + 128) * p.T * 4 < 0x7fffffffL;
    if (full) {
#define X(GM, XM)                                                                                                   \
        if (p.g_mode == GM && p.x_mode == XM) {                                                                     \
            if (p.T % BF_BK == 0) V100_GGL((pw_wgrad_bf16_fast_kernel<GM, XM, false>), grid, dim3(256), 0, st, p); \
            else V100_GGL((pw_wgrad_bf16_fast_kernel<GM, XM, true>), grid, dim3(256), 0, st, p);          \
            return;                                                                                                 \
        }
        PW_WG_COMBOS(X)
#undef X
    }
#define X(GM, XM)                                                                                                   \
    if (p.g_mode == GM && p.x_mode == XM) {                                                                         \
        if (tv) V100_GGL((pw_wgrad_bf16_kernel<GM, XM, true>), grid, dim3(256), 0, st, p);                   \
        else V100_GGL((pw_wgrad_bf16_kernel<GM, XM, false>), grid, dim3(256), 0, st, p);                     \
        return;                                                                                                     \
    }
    PW_WG_COMBOS(X)
#undef X
    if (tv) V100_GGL((pw_wgrad_bf16_kernel<-1, -1, true>), grid, dim3(256), 0, st, p);
    else V100_GGL((pw_wgrad_bf16_kernel<-1, -1, false>), grid, dim3(256), 0, st, p);
}


// ---------------------------------------------------------------------------------------------
// Round 5: the LATENCY form of the two inference GEMMs (channel-major eval blocks, block.hip v100_ir_fwd_eval: ONE matrix
// Y [M][N] = epi(W [M][K] X [K][N]), N = B * pitch(T)).  At the configs' own sizes (configs[4] at B = 32: N = 1792; configs[0]: N = 256) the
// throughput kernels above launch 4-112 workgroups of 256 x 128 and each walks its k-tiles with two register stages in flight: the
// k-loop is a chain of memory latencies (30 us for K = 2048 whatever N is; profiles/r05_infer_small_chains.txt), on a chip that is
// 90 % idle.  Here: 64 x 64 tiles (4 waves, one 32 x 32 accumulator each) so that even N = 256 gives 32-128 workgroups, and
// PW_LAT_NSTG (6) k-tiles of loads in flight per workgroup -- 16-24 registers a stage -- so a workgroup waits for memory about once,
// not once per k-tile.  Same LDS images, fragment reads, MFMA order (k ascending) and epilogue arithmetic as the kernels above.
// EPI 2 (expand): X fp32 -> Y 16-bit = relu6(ea acc + eb);  EPI 3 (project): X 16-bit -> Y fp32 = ea acc + eb (+ R).
template <int EPI, bool F16>
__global__ __launch_bounds__(256) void pw_gemm_lat_kernel(PwParams p) {
    static_assert(EPI == PW_EPI_AFFINE_RELU6 || EPI == PW_EPI_AFFINE_RES, "inference epilogues only");
    constexpr bool XB = EPI == PW_EPI_AFFINE_RES;                   // project: 16-bit X;  expand: fp32 X
    using XReg = std::conditional_t<XB, u32x2, u32x4>;
    constexpr int EX = XB ? 2 : 4;
    constexpr int NSTG = PW_LAT_NSTG;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 8192];     // As[2][64][64] | Bs[2][64][64] bf16; epilogue: [64][68] fp32
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * 8192;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_mt = p.M >> 6;
    const int mt = blockIdx.x % n_mt, tt = blockIdx.x / n_mt;
    const int m0 = mt * 64, t0 = tt * 64;
    const int K = p.K, N = p.T;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.Abf, (unsigned)p.M * K * 2u);
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(p.X, (unsigned)K * N * EX);
    int voA[2], ldsA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = tid + 256 * i, row = piece >> 3, ch = piece & 7;
        voA[i] = ((m0 + row) * K + ch * 8) * 2;
        ldsA[i] = bf_off(row, ch);
    }
    const int b_tq = (tid & 15) * 4, b_kg = tid >> 4;               // X patch: columns b_tq .. +3 of rows 4 b_kg .. +3
    int voX[4], ldsB[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) voX[e] = ((4 * b_kg + e) * N + t0 + b_tq) * EX;
#pragma unroll
    for (int q = 0; q < 4; ++q) ldsB[q] = bf_off(b_tq + q, b_kg >> 1) + (b_kg & 1) * 8;
    const int nk = K >> 6;
    u32x4 ra[NSTG][2];
    XReg rb[NSTG][4];
    auto load = [&](int kt, auto stg) {                             // (past the end: the last k-tile again -- unconditional, see the kernels above)
        constexpr int SG = decltype(stg)::value;
        const int k0 = min(kt, nk - 1) * BF_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[SG][i] = __builtin_amdgcn_raw_buffer_load_b128(rA, voA[i], k0 * 2, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (XB) rb[SG][e] = __builtin_amdgcn_raw_buffer_load_b64(rX, voX[e], k0 * N * EX, 0);
            else rb[SG][e] = __builtin_amdgcn_raw_buffer_load_b128(rX, voX[e], k0 * N * EX, 0);
        }
    };
    auto store = [&](int buf, auto stg) {
        constexpr int SG = decltype(stg)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(As + buf * 8192 + ldsA[i]) = ra[SG][i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint2 o;
            if constexpr (XB) {                                     // [k][t] -> [t][k]: a byte shuffle of the loaded words
                const unsigned sel = (q & 1) ? 0x07060302u : 0x05040100u;
                o.x = __builtin_amdgcn_perm(rb[SG][1][q >> 1], rb[SG][0][q >> 1], sel);
                o.y = __builtin_amdgcn_perm(rb[SG][3][q >> 1], rb[SG][2][q >> 1], sel);
            } else {
                o.x = pack16<F16>(__builtin_bit_cast(f32x4, rb[SG][0])[q], __builtin_bit_cast(f32x4, rb[SG][1])[q]);
                o.y = pack16<F16>(__builtin_bit_cast(f32x4, rb[SG][2])[q], __builtin_bit_cast(f32x4, rb[SG][3])[q]);
            }
            *reinterpret_cast<uint2*>(Bs + buf * 8192 + ldsB[q]) = o;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int lr = lane & 31, lh = lane >> 5, sw = (lr >> 1) & 7;
    const int rdA0 = (wm * 32 + lr) * 128, rdB0 = (wn * 32 + lr) * 128;
    auto mfma_block = [&](int buf) {                                // all eight fragment reads first: one LDS latency per k-tile, not four
        bf16x8 a[BF_BK / 16], b[BF_BK / 16];
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {
            const int co = ((ks * 2 + lh) ^ sw) << 4;
            a[ks] = *reinterpret_cast<const bf16x8*>(As + buf * 8192 + rdA0 + co);
            b[ks] = *reinterpret_cast<const bf16x8*>(Bs + buf * 8192 + rdB0 + co);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) acc = mfma16<F16>(a[ks], b[ks], acc);
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1 % NSTG>;
    using S2 = std::integral_constant<int, 2 % NSTG>; using S3 = std::integral_constant<int, 3 % NSTG>;
    using S4 = std::integral_constant<int, 4 % NSTG>; using S5 = std::integral_constant<int, 5 % NSTG>;
    static_assert(NSTG == 2 || NSTG == 4 || NSTG == 6, "register stages");
    // (the requests stay in stage order -- sched_barrier -- so that the wait at a stage's use counts the YOUNGER stages, in the loop's
    //  first trip as in every later one; reordered, hipcc's merge at the loop head degenerates into vmcnt(0) once per trip)
#define LAT_SB() __builtin_amdgcn_sched_barrier(0)
    load(0, S0{}); LAT_SB(); load(1, S1{}); LAT_SB();
    if constexpr (NSTG >= 4) { load(2, S2{}); LAT_SB(); load(3, S3{}); LAT_SB(); }
    if constexpr (NSTG >= 6) { load(4, S4{}); LAT_SB(); load(5, S5{}); LAT_SB(); }
    int kt = 0;
    // k-tile kt: registers of stage kt % NSTG -> LDS buffer kt & 1, the request NSTG tiles ahead into the same registers, ONE barrier
    // (buffer kt & 1 was last read by the MFMAs of tile kt - 2, which every wave finished before the barrier of tile kt - 1)
#define LAT_STEP(SG)                                                                                                              \
    {                                                                                                                             \
        store(kt & 1, SG{});                                                                                                      \
        LAT_SB();                                                                                                                 \
        load(kt + NSTG, SG{});                                                                                                    \
        LAT_SB();                                                                                                                 \
        __syncthreads();                                                                                                          \
        mfma_block(kt & 1);                                                                                                       \
        ++kt;                                                                                                                     \
    }
    // whole trips without exits (an exit inside the trip makes every `break` a predecessor of the loop head, and hipcc's wait at the
    // head then covers the path on which stage 0 was requested LAST: vmcnt(0) once per trip), then the last nk % NSTG tiles straight-line
    for (; kt + NSTG <= nk;) {
        LAT_STEP(S0) LAT_STEP(S1)
        if constexpr (NSTG >= 4) { LAT_STEP(S2) LAT_STEP(S3) }
        if constexpr (NSTG >= 6) { LAT_STEP(S4) LAT_STEP(S5) }
    }
    if (kt < nk) LAT_STEP(S0)
    if (kt < nk) LAT_STEP(S1)
    if constexpr (NSTG >= 4) {
        if (kt < nk) LAT_STEP(S2)
        if (kt < nk) LAT_STEP(S3)
    }
    if constexpr (NSTG >= 6) {
        if (kt < nk) LAT_STEP(S4)
    }
#undef LAT_STEP
#undef LAT_SB
    // epilogue through LDS: the 64 x 64 fp32 tile (row pitch 68), then 16 consecutive columns of one row per thread
    __syncthreads();
    float* tile = reinterpret_cast<float*>(smem);
    {
        const int col = wn * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 68 + col] = acc[r];
    }
    __syncthreads();
    const int row = tid >> 2, cg = (tid & 3) * 16;
    const int m = m0 + row;
    const float ea = p.ea[m], eb = p.eb[m];
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(tile + row * 68 + cg + 4 * j);
    if constexpr (EPI == PW_EPI_AFFINE_RES) {
        float* y = p.Y + (size_t)m * N;
        const float* rr = p.R ? p.R + (size_t)m * N : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + cg + 4 * j;
            if (t >= N) continue;
            f32x4 o;
            f32x4 rv = {0.f, 0.f, 0.f, 0.f};
            if (rr) rv = *reinterpret_cast<const f32x4*>(rr + t);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(v[j][e], ea, eb) + rv[e];
            *reinterpret_cast<f32x4*>(y + t) = o;
        }
    } else {
        u16* y = reinterpret_cast<u16*>(p.Y) + (size_t)m * N;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + cg + 8 * h;
            if (t >= N) continue;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 2 * h + (e >> 1), c = (e & 1) * 2;
                o[e] = pack16<F16>(relu6f(fmaf(v[j][c], ea, eb)), relu6f(fmaf(v[j][c + 1], ea, eb)));
            }
            *reinterpret_cast<u32x4*>(y + t) = o;
        }
    }
}

static bool pw_launch_gemm_lat(const PwParams& p, hipStream_t st) {
    if (!PW_LAT || p.B != 1 || p.x_mode != 0 || p.bias || (p.M & 63) || (p.K & 63) || (p.T & 7) || p.K < 64) return false;
    if ((long)p.K * p.T * 4 >= 0x7fffffffL || (long)p.M * p.T * 4 >= 0x7fffffffL || (long)p.M * p.K * 2 >= 0x7fffffffL) return false;
    const long big_tiles = (long)((p.M + 255) / 256) * ((p.T + PW_BN - 1) / PW_BN);
    static const long max_tiles = [] { const char* e = getenv("V100_PW_LAT_MAXTILES"); return e ? atol(e) : (long)PW_LAT_MAXTILES; }();
    if (big_tiles > max_tiles) return false;
    const dim3 grid((unsigned)((p.M >> 6) * ((p.T + 63) >> 6)));
    const int f16 = p.io16 & PW_IO_F16, io = p.io16 & ~PW_IO_F16;
    if ((p.fmt == 2) != (f16 != 0)) return false;
    if (p.epi_mode == PW_EPI_AFFINE_RELU6 && io == PW_IO_Y) {
        if (f16) V100_GGL((pw_gemm_lat_kernel<PW_EPI_AFFINE_RELU6, true>), grid, dim3(256), 0, st, p);
        else V100_GGL((pw_gemm_lat_kernel<PW_EPI_AFFINE_RELU6, false>), grid, dim3(256), 0, st, p);
        return true;
    }
    if (p.epi_mode == PW_EPI_AFFINE_RES && io == PW_IO_X) {
        if (f16) V100_GGL((pw_gemm_lat_kernel<PW_EPI_AFFINE_RES, true>), grid, dim3(256), 0, st, p);
        else V100_GGL((pw_gemm_lat_kernel<PW_EPI_AFFINE_RES, false>), grid, dim3(256), 0, st, p);
        return true;
    }
    return false;
}

// 16-bit activation storage (PwParams::io16 / WgParams::io16): the combinations the block executor issues in "act16" mode.
// false = no instantiation for this (modes, mask) or the shape does not fit the buffer-addressed kernels.
bool pw_launch_gemm_bf16_io(const PwParams& p, hipStream_t st) {
    const int P = pw_pitch16(p.T, p.B);
    if (pw_launch_gemm_lat(p, st)) return true;
    if (p.io16 & PW_IO_F16) {
        // inference at precision "fp16" with fp16-stored hidden tensors: the two eval-mode GEMMs of a block (256-row tiles; 128 for M < 256)
        if (!((p.K & 1) == 0 && (long)(p.K + 64) * P * 4 < 0x7fffffffL && (long)(p.M + 128) * p.K * 2 < 0x7fffffffL &&
              (long)p.B * p.M * P * 4 < 0x7fffff00L && p.x_mode == 0)) return false;
        const bool big16 = p.M >= 256;
        PwParams pb = p;
        pb.n_mtiles = (p.M + (big16 ? 255 : 127)) / (big16 ? 256 : 128);
        const dim3 grid((unsigned)((long)pb.n_mtiles * p.n_ttiles * p.B));
#define XF(EP, IOV)                                                                                                                  \
        if (p.epi_mode == EP && p.io16 == (IOV)) {                                                                                  \
            if (big16) V100_GGL((pw_gemm_bf16_fast_kernel<0, EP, 256, true, false, (IOV)>), grid, dim3(512), 0, st, pb);            \
            else V100_GGL((pw_gemm_bf16_fast_kernel<0, EP, 128, true, false, (IOV)>), grid, dim3(256), 0, st, pb);                  \
            return true;                                                                                                            \
        }
#if PW_WS
        // the eval-mode project GEMM (fp16-stored h2 in, fp32 block output): the wave-specialised kernel, as at precision "bf16"
        if (big16 && (PW_WS & 1) && p.epi_mode == 3 && p.io16 == (PW_IO_X | PW_IO_F16) && (p.K & 63) == 0 && p.K >= PW_WS_MINK && p.K <= WS_MAXK && (p.M & 255) == 0) {
            V100_GGL((pw_gemm_bf16_ws_kernel<0, 3, (PW_IO_X | PW_IO_F16)>), grid, dim3(768), 0, st, pb);
            return true;
        }
#endif
        XF(2, PW_IO_Y | PW_IO_F16) XF(3, PW_IO_X | PW_IO_F16)
#undef XF
        return false;
    }
    const bool big = p.M >= 256 && p.K > PW_BM128_MAXK;
    PwParams pb = p;
    pb.n_mtiles = (p.M + (big ? 255 : 127)) / (big ? 256 : 128);
    const dim3 grid((unsigned)((long)pb.n_mtiles * p.n_ttiles * p.B));
#define X(XM, EP, IOV)                                                                                                          \
    if (p.x_mode == XM && p.epi_mode == EP && p.io16 == (IOV)) {                                                                \
        if (big) V100_GGL((pw_gemm_bf16_fast_kernel<XM, EP, 256, false, false, (IOV)>), grid, dim3(512), 0, st, pb);  \
        else V100_GGL((pw_gemm_bf16_fast_kernel<XM, EP, 128, false, false, (IOV)>), grid, dim3(256), 0, st, pb);      \
        return true;                                                                                                            \
    }
#if PW_OV
    // epilogue under the next tile's main loop: plain bf16 X, bf16 Y, whole 256-row tiles, K = 256 / 512, >= 2 tiles per workgroup
    if (big && p.x_mode == 0 && (p.K == 256 || p.K == 512) && (p.M & 255) == 0 && !p.bias && (long)p.B * p.K * P * 2 < 0x7fffffffL &&
        (long)p.B * p.M * P * 2 < 0x7ffffff0L) {
        const long nt_ = (long)pb.n_mtiles * p.n_ttiles * p.B;
        const unsigned gridp = 256;                    // one workgroup per CU walks tiles v = blockIdx.x, + 256, ... (any count)
#if PW_SL
        if (nt_ >= gridp) {
            if (p.epi_mode == 1 && p.io16 == (PW_IO_X | PW_IO_Y) && ((PW_SL >> (p.K == 512 ? 0 : 1)) & 1)) {
                if (p.K == 512) V100_GGL((pw_gemm_bf16_sl_kernel<1, 8>), dim3(gridp), dim3(768), 0, st, pb);
                else V100_GGL((pw_gemm_bf16_sl_kernel<1, 4>), dim3(gridp), dim3(768), 0, st, pb);
                return true;
            }
            if (p.epi_mode == 4 && p.io16 == (PW_IO_X | PW_IO_R | PW_IO_Y) && ((PW_SL >> (p.K == 512 ? 2 : 3)) & 1)) {
                if (p.K == 512) V100_GGL((pw_gemm_bf16_sl_kernel<4, 8>), dim3(gridp), dim3(768), 0, st, pb);
                else V100_GGL((pw_gemm_bf16_sl_kernel<4, 4>), dim3(gridp), dim3(768), 0, st, pb);
                return true;
            }
        }
#endif
        if (nt_ >= 2 * gridp) {
            if (p.epi_mode == 1 && p.io16 == (PW_IO_X | PW_IO_Y)) {
                if (p.K == 512) V100_GGL((pw_gemm_bf16_ov_kernel<1, (PW_IO_X | PW_IO_Y), 8>), dim3(gridp), dim3(512), 0, st, pb);
                else V100_GGL((pw_gemm_bf16_ov_kernel<1, (PW_IO_X | PW_IO_Y), 4>), dim3(gridp), dim3(512), 0, st, pb);
                return true;
            }
            if (PW_OV_MASK && p.epi_mode == 4 && p.io16 == (PW_IO_X | PW_IO_R | PW_IO_Y)) {
                if (p.K == 512) V100_GGL((pw_gemm_bf16_ov_kernel<4, (PW_IO_X | PW_IO_R | PW_IO_Y), 8>), dim3(gridp), dim3(512), 0, st, pb);
                else V100_GGL((pw_gemm_bf16_ov_kernel<4, (PW_IO_X | PW_IO_R | PW_IO_Y), 4>), dim3(gridp), dim3(512), 0, st, pb);
                return true;
            }
        }
    }
#endif
#if PW_WS
    // wave-specialised kernel: bf16-stored X operands, 256-row tiles, whole tiles in M (rows past M are handled by the epilogue)
#define XS(XM, EP, IOV)                                                                                                         \
    if (big && ((PW_WS >> (XM)) & 1) && p.x_mode == XM && p.epi_mode == EP && p.io16 == (IOV) && (p.K & 63) == 0 && p.K >= PW_WS_MINK && p.K <= WS_MAXK && (p.M & 255) == 0) { \
        V100_GGL((pw_gemm_bf16_ws_kernel<XM, EP, (IOV)>), grid, dim3(768), 0, st, pb);                                           \
        return true;                                                                                                            \
    }
    XS(1, 1, PW_IO_X) XS(1, 1, PW_IO_X | PW_IO_Y)                       // project forward
    XS(2, 5, PW_IO_X | PW_IO_X2) XS(2, 0, PW_IO_X | PW_IO_X2)           // expand backward-data
    XS(0, 4, PW_IO_X | PW_IO_R | PW_IO_Y)                               // project backward-data
    XS(0, 1, PW_IO_X | PW_IO_Y)                                         // expand forward on the bf16 shadow
    XS(0, 3, PW_IO_X)                                                   // eval-mode project (K = the hidden width): y = bn3(W3 h2) (+ x)
    XS(0, 5, PW_IO_X) XS(0, 0, PW_IO_X)                                 // expand backward-data on the finished gradient da1 (round 5): plain bf16 X, fp32 dx (+ dy)
    XS(0, 5, PW_IO_X | PW_IO_R | PW_IO_Y) XS(0, 5, PW_IO_X | PW_IO_Y) XS(0, 5, PW_IO_X | PW_IO_R) XS(0, 0, PW_IO_X | PW_IO_Y)     // ... with the gradient stream between blocks in bf16 (round 6): dy in and / or dx out
#undef XS
#endif
#if PW_PERSIST
    // short-K GEMMs with several tiles per CU: persistent workgroups (grid = tiles / 2 or / 4 when that divides evenly)
#define XP(XM, EP, IOV)                                                                                                         \
    if (big && p.x_mode == XM && p.epi_mode == EP && p.io16 == (IOV) && p.K <= 512) {                                           \
        const long nt_ = (long)pb.n_mtiles * p.n_ttiles * p.B;                                                                  \
        const long per = (nt_ % 1024 == 0 && nt_ >= 1024) ? 4 : ((nt_ % 512 == 0 && nt_ >= 512) ? 2 : 1);                       \
        if (per > 1) {                                                                                                          \
            V100_GGL((pw_gemm_bf16_fast_kernel<XM, EP, 256, false, false, (IOV), true>), dim3((unsigned)(nt_ / per)),  \
                               dim3(512), 0, st, pb);                                                                           \
            return true;                                                                                                        \
        }                                                                                                                       \
    }
    XP(0, 1, PW_IO_Y)          // expand forward: 61 -> 56.5 us at 2048 x 512 x (32 x 512)
    XP(0, 1, PW_IO_X | PW_IO_Y)    // ... reading the bf16 shadow of the block input (act16 level 4)
    // (the project backward-data GEMM, same shape, does not fit: its mask epilogue keeps 96 registers of R / coefficient /
    //  statistics values beside the two staging stages -- 59 VGPRs spilled, 69 -> 118 us)
#undef XP
#endif
    // project backward-data at K <= 256 (4 k-tiles per block tile: prologue and epilogue are most of a tile's time): 128-row tiles,
    // two workgroups per CU, so one's epilogue runs under the other's main loop (30.7 -> 27.5 us at 1024 x 256 x (32 x 512); the
    // expand forward GEMM of that shape loses, 23 -> 30 us: its fp32 X operand is then staged twice as often)
    if (p.x_mode == 0 && p.epi_mode == 4 && p.io16 == (PW_IO_X | PW_IO_R | PW_IO_Y) && p.K <= 256 && big) {
        PwParams ps = p;
        ps.n_mtiles = (p.M + 127) / 128;
        V100_GGL((pw_gemm_bf16_fast_kernel<0, 4, 128, false, false, (PW_IO_X | PW_IO_R | PW_IO_Y)>),
                           dim3((unsigned)((long)ps.n_mtiles * p.n_ttiles * p.B)), dim3(256), 0, st, ps);
        return true;
    }
    X(0, 1, PW_IO_Y)                      // expand forward: a1 out
    X(0, 1, PW_IO_X | PW_IO_Y)            // ... X = bf16 shadow of the block input
    X(1, 1, PW_IO_X)                      // project forward: a2 in (BN2 + ReLU6 on load)
    X(0, 4, PW_IO_R)                      // project backward-data: ReLU6 mask / BN2-backward sums from a2
    X(2, 5, PW_IO_X2) X(2, 0, PW_IO_X2)   // expand backward-data: BN1-backward affine of (dz1, a1)
    X(0, 4, PW_IO_R | PW_IO_Y)            // ... with the hidden gradients stored as bf16 too
    X(2, 5, PW_IO_X | PW_IO_X2) X(2, 0, PW_IO_X | PW_IO_X2)
    X(1, 1, PW_IO_X | PW_IO_Y)            // ... and the project output a3 / its gradient da3
    X(0, 4, PW_IO_X | PW_IO_R | PW_IO_Y)
    X(0, 2, PW_IO_Y)                      // eval-mode expand: h1 = relu6(bn1(W1 x)) stored as bf16 (inference, block executor)
    X(0, 3, PW_IO_X)                      // eval-mode project: y = bn3(W3 h2) (+ x), h2 bf16 in
    X(0, 5, PW_IO_X) X(0, 0, PW_IO_X)     // expand backward-data on da1 (plain bf16 X), fp32 out (+ residual gradient)
    X(0, 5, PW_IO_X | PW_IO_R | PW_IO_Y) X(0, 5, PW_IO_X | PW_IO_Y) X(0, 5, PW_IO_X | PW_IO_R) X(0, 0, PW_IO_X | PW_IO_Y)     // ... bf16 residual gradient in / bf16 dx out (the stack's 16-bit gradient stream)
#undef X
    return false;
}

bool pw_launch_wgrad_bf16_io(const WgParams& p, dim3 grid, hipStream_t st) {
    const int P = pw_pitch16(p.T, p.B);
    if (!((long)(p.M + 256) * P * 4 < 0x7fffffffL && (long)(p.K + 256) * P * 4 < 0x7fffffffL)) return false;
#if PW_WG_WIDE
    // 256-row tile on the plain operand, 128 rows on the transformed one (pw_wgrad_bf16_wide_kernel)
#define XW(GM, XM, IOV, GR, XR, NS)                                                                                                    \
    if (p.g_mode == GM && p.x_mode == XM && p.io16 == (IOV) && p.M >= GR && p.K >= XR) {                                            \
        WgParams pw = p;                                                                                                            \
        pw.n_mtiles = (p.M + GR - 1) / GR;                                                                                          \
        pw.n_ktiles = (p.K + XR - 1) / XR;                                                                                          \
        const dim3 gw((unsigned)(pw.n_mtiles * pw.n_ktiles * p.S));                                                                 \
        if (p.T % BF_BK == 0) V100_GGL((pw_wgrad_bf16_wide_kernel<GM, XM, false, (IOV), GR, XR, NS>), gw, dim3(512), 0, st, pw); \
        else V100_GGL((pw_wgrad_bf16_wide_kernel<GM, XM, true, (IOV), GR, XR, NS>), gw, dim3(512), 0, st, pw);                \
        return true;                                                                                                                \
    }
#if PW_WG_WS
    // wave-specialised form: full tiles, every operand bf16-stored
#define XS(GM, XM, IOV, GR, XR, NSW)                                                                                                  \
    if (p.g_mode == GM && p.x_mode == XM && p.io16 == (IOV) && p.M % GR == 0 && p.K % XR == 0) {                                    \
        WgParams pw = p;                                                                                                            \
        pw.n_mtiles = p.M / GR;                                                                                                     \
        pw.n_ktiles = p.K / XR;                                                                                                     \
        const dim3 gw((unsigned)(pw.n_mtiles * pw.n_ktiles * p.S));                                                                 \
        if (p.T % BF_BK == 0) V100_GGL((pw_wgrad_bf16_ws_kernel<GM, XM, false, (IOV), GR, XR, NSW>), gw, dim3(512 + 64 * NSW), 0, st, pw); \
        else V100_GGL((pw_wgrad_bf16_ws_kernel<GM, XM, true, (IOV), GR, XR, NSW>), gw, dim3(512 + 64 * NSW), 0, st, pw);        \
        return true;                                                                                                                \
    }
    // (measured, profiles/r03_ws_gemm.txt: the project gradient -13 % at 512 channels, the expand gradient -12 %; eight staging
    //  waves instead of four: slower)
    if constexpr (PW_WG_WS & 2) { XS(2, 0, WG_IO_G | WG_IO_G2 | WG_IO_X, 128, 256, (PW_WG_WS & 8 ? 8 : 4)) }
    if constexpr (PW_WG_WS & 1) { XS(0, 1, WG_IO_G | WG_IO_X, 256, 128, (PW_WG_WS & 4 ? 8 : 4)) }
    if constexpr (PW_WG_WS & 2) { XS(0, 0, WG_IO_G | WG_IO_X, 128, 256, 4) }       // expand gradient on the finished gradient da1 (round 5)
#undef XS
#endif
    XW(2, 0, WG_IO_G | WG_IO_G2 | WG_IO_X, 128, 256, 2)      // ... X = bf16 shadow of the block input: copied as loaded, 32 registers a stage
    XW(2, 0, WG_IO_G | WG_IO_G2, 128, 256, PW_WG_EXPAND_NST)      // expand: G = affine2(dz1, a1), X = block input (plain fp32)
    XW(0, 1, WG_IO_G | WG_IO_X, 256, 128, 2)       // project: G = da3 (plain bf16, copied), X = relu6(bn2(a2))
    XW(0, 0, WG_IO_G | WG_IO_X, 128, 256, 2)       // expand on da1: both operands plain bf16, copied as loaded
    XW(0, 0, WG_IO_G, 128, 256, 1)                 // ... X = the block input in fp32 (no shadow)
#undef XW
#endif
#define X(GM, XM, IOV)                                                                                                              \
    if (p.g_mode == GM && p.x_mode == XM && p.io16 == (IOV)) {                                                                      \
        if (p.T % BF_BK == 0) V100_GGL((pw_wgrad_bf16_fast_kernel<GM, XM, false, false, (IOV)>), grid, dim3(256), 0, st, p); \
        else V100_GGL((pw_wgrad_bf16_fast_kernel<GM, XM, true, false, (IOV)>), grid, dim3(256), 0, st, p);                \
        return true;                                                                                                                \
    }
    X(2, 0, WG_IO_G2)                     // expand backward-weight: G = BN1-backward affine of (dz1, a1), X = block input
    X(0, 1, WG_IO_X)                      // project backward-weight: X = relu6(bn2(a2))
    X(2, 0, WG_IO_G | WG_IO_G2)           // ... with dz1 stored as bf16
    X(2, 0, WG_IO_G | WG_IO_G2 | WG_IO_X) // ... and X = bf16 shadow of the block input
    X(0, 1, WG_IO_G | WG_IO_X)            // ... with da3 stored as bf16
    X(0, 0, WG_IO_G | WG_IO_X) X(0, 0, WG_IO_G)      // expand gradient on the finished gradient da1 (bf16), X = bf16 shadow / fp32
#undef X
    return false;
}

// Tap-addressed X operand (PwParams / WgParams): plain store (+bias) or +R epilogue, no prologues.  false = the shape
// does not fit the buffer-addressed kernels (the caller falls back to an explicit im2col copy).
bool pw_taps_fit_bf16(int B, int M, int cx, int ntap, int T, int Tx) {
    const long K = (long)ntap * cx;
    return cx % BF_BK == 0 && ntap >= 1 && ntap <= 8 && (long)(cx + 64) * Tx * 4 < 0x7fffff00L && (M + 128L) * K * 2 < 0x7fffffffL &&
           (long)B * M * T * 4 < 0x7fffff00L && (M + 128L) * Tx * 4 < 0x7fffff00L;
}

bool pw_launch_gemm_taps_bf16(const PwParams& p, dim3 grid, hipStream_t st) {
    if (!pw_taps_fit_bf16(p.B, p.M, p.cx, p.ntap, p.T, p.Tx)) return false;
    const bool big = p.M >= 256;
    PwParams pb = p;
    pb.n_mtiles = (p.M + 255) / 256;
    const dim3 gridb((unsigned)((long)pb.n_mtiles * p.n_ttiles * p.B));
#define X(EP, F16)                                                                                                      \
    if (p.epi_mode == EP && (p.fmt == 2) == F16) {                                                                      \
        if (big) V100_GGL((pw_gemm_bf16_fast_kernel<0, EP, 256, F16, true>), gridb, dim3(512), 0, st, pb);   \
        else V100_GGL((pw_gemm_bf16_fast_kernel<0, EP, 128, F16, true>), grid, dim3(256), 0, st, p);          \
        return true;                                                                                                    \
    }
    X(0, false) X(5, false) X(0, true)
#undef X
    return false;
}

bool pw_launch_wgrad_taps_bf16(const WgParams& p, dim3 grid, hipStream_t st) {
    if (!((p.M + 128L) * p.Tg * 4 < 0x7fffff00L && (p.cx + 128L) * p.Tx * 4 < 0x7fffff00L && p.ntap >= 1 && p.ntap <= 8)) return false;
    if (p.T % BF_BK == 0) V100_GGL((pw_wgrad_bf16_fast_kernel<0, 0, false, true>), grid, dim3(256), 0, st, p);
    else V100_GGL((pw_wgrad_bf16_fast_kernel<0, 0, true, true>), grid, dim3(256), 0, st, p);
    return true;
}
