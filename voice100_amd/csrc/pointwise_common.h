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
#pragma once
#include <type_traits>
#include "common.h"

enum { PW_X_NONE = 0, PW_X_AFFINE_RELU6 = 1, PW_X_AFFINE2 = 2 };
enum { PW_EPI_STORE = 0, PW_EPI_STATS = 1, PW_EPI_AFFINE_RELU6 = 2, PW_EPI_AFFINE_RES = 3, PW_EPI_MASK_STATS = 4, PW_EPI_ADD = 5 };

#define PW_BM 128
#define PW_BN 128

// Timing-only ablation builds (tools/ab_variants.sh PW_ABLATE ...): bit 0 drops the X loads, bit 1 the A loads (zero-record
// buffer descriptors: same instruction stream, no traffic), bit 2 the epilogue's global stores.  0 in the product.
#ifndef PW_ABLATE
#define PW_ABLATE 0
#endif
#ifndef PW_EPI_FAST
#define PW_EPI_FAST 2      /* 1: lean epilogue for interior tiles; 2: also for the partial last t-tile (PT) */
#endif
#ifndef PW_LAT
#define PW_LAT 1             /* latency form of the inference GEMMs when the matrix has too few 256 x 128 tiles to fill the chip */
#endif
#ifndef PW_LAT_MAXTILES
#define PW_LAT_MAXTILES 64
#endif
#ifndef PW_LAT_NSTG
#define PW_LAT_NSTG 4
#endif
#ifndef PW_PERSIST
#define PW_PERSIST 1         /* persistent workgroups with cross-tile prefetch for the short-K training GEMMs */
#endif
#ifndef PW_WG_EXPAND_NST
#define PW_WG_EXPAND_NST 3    /* register staging of the expand backward-weight kernel: 1 one stage, 2 two (spills), 3 two for G + one for X */
#endif
#ifndef PW_WG_WIDE
#define PW_WG_WIDE 1         /* 256 x 128 backward-weight tiles for the act16 combinations (0: the 128 x 128 kernel everywhere) */
#endif
#ifndef PW_WS
#define PW_WS 7              /* wave-specialised NN GEMM (pw_gemm_bf16_ws_kernel): bit x_mode set = that prologue family uses it.  With
                                PW_WS_MINK = 1024 that is: project forward and expand backward-data (transform on load, K = the hidden
                                width) in training, the eval-mode project GEMM (plain bf16 h2, K = the hidden width) in inference */
#endif
#ifndef PW_OV
#define PW_OV 1              /* short-K GEMMs with the epilogue under the next tile's main loop (pw_gemm_bf16_ov_kernel): expand forward */
#endif
#ifndef PW_OV_MASK
#define PW_OV_MASK 1         /* ... and project backward-data (mask epilogue) */
#endif
#ifndef PW_SL
#define PW_SL 4              /* split-role short-K GEMM (pw_gemm_bf16_sl_kernel) instead of the overlapped-epilogue one: bit 0 expand forward
                                K = 512, bit 1 K = 256, bit 2 project backward-data K = 512, bit 3 K = 256 */
#endif
#ifndef PW_SL_PRIO
#define PW_SL_PRIO 3
#endif
#ifndef PW_SL_RAWBAR
#define PW_SL_RAWBAR 1
#endif
#ifndef PW_OV_RAWBAR
#define PW_OV_RAWBAR 0       /* the overlapped-epilogue kernel's k-tile barrier as s_waitcnt lgkmcnt(0) + s_barrier (no vmcnt(0)) */
#endif
#ifndef PW_SL_DBG
#define PW_SL_DBG 0
#endif
#ifndef PW_OV_XTR
#define PW_OV_XTR 1          /* round 5: the overlapped-epilogue kernel keeps its X image [k][t] as loaded (16-byte copies) and reads the B
                                fragments with ds_read_b64_tr_b16 (0: [t][k] image built with byte permutes + 8-byte column stores) */
#endif
#ifndef PW_OV_TOUCH
#define PW_OV_TOUCH 0        /* overlapped-epilogue kernel: L2 touch of a tile's R lines (bit 0) / Y lines (bit 1) at the start of its k-loop */
#endif
#ifndef PW_OV_CP_Y
#define PW_OV_CP_Y 0         /* cache policy (buffer aux bits) of the overlapped-epilogue kernel's Y stores: 1 sc0, 2 nt, 3 both */
#endif
#ifndef PW_OV_ABL
#define PW_OV_ABL 0          /* timing-only builds of pw_gemm_bf16_ov_kernel (WRONG results): 1 no epilogue, 2 no X staging, 4 no A staging,
                                8 no barrier, 16 no fragment reads, 32 no Y stores, 64 no statistics (partial sums through LDS), 128 no R loads */
#endif
#ifndef PW_OV_GAP
#define PW_OV_GAP 8
#endif
#ifndef PW_WG_ABL
#define PW_WG_ABL 0          /* timing-only builds of pw_wgrad_bf16_ws_kernel: 1 / 2 every plain / transformed load re-reads the first tile, 4 no
                                transform, 8 staging waves: loads only, 16 no MFMA */
#endif
#ifndef PW_WG_WS
#define PW_WG_WS 3           /* wave-specialised backward-weight kernel (pw_wgrad_bf16_ws_kernel), all-bf16 act16 combinations: bit 0 the project
                                gradient, bit 1 the expand gradient; bits 2 / 3: eight staging waves instead of four for the former / latter */
#endif
#ifndef PW_WS_ABL
#define PW_WS_ABL 0
#endif
#ifndef PW_WS_MINK
#define PW_WS_MINK 1024      /* ... at K >= this (shorter K: the persistent / 128-row forms of the 8-wave kernel win, profiles/r03_ws_gemm.txt) */
#endif
#ifndef PW_BM128_MAXK
#define PW_BM128_MAXK 0      /* A/B knob: 128-row tiles (two workgroups per CU) for GEMMs with K <= this */
#endif

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
    int fmt;              // 16-bit operand format of the bf16-path kernels: 1 bf16, 2 fp16 (inference combinations only)
    // Tap-addressed X operand (dense k-tap convolutions as ONE GEMM, no im2col copy): when ntap > 0 the contraction
    // index is k = tap * cx + c and row k is row c of a zero-padded tensor X [B][cx][Tx] read from column
    // t + shift(tap); shifts are >= 0 (the padding absorbs the negative taps), 4 bits each in `shifts`.  K = ntap * cx.
    int ntap, cx, Tx;
    unsigned shifts;
    // 16-bit storage of the big hidden tensors (bf16 mode, "act16"): bit mask of PW_IO_* -- those tensors are bf16 [B][C][P]
    // with the row pitch P = pw_pitch16(T) (a multiple of 8 elements, so every 4- or 8-element access is 8 / 16-byte aligned
    // whatever T is); X / X2 / Y / R then point at bf16 data.  0 = everything fp32 (pitch T).
    int io16;
};
// PW_IO_F16: the 16-bit tensors named by the other bits hold IEEE fp16, not bf16 (inference at precision "fp16": the hidden activations
// are stored in the operand format of the GEMMs that read them)
enum { PW_IO_X = 1, PW_IO_X2 = 2, PW_IO_Y = 4, PW_IO_R = 8, PW_IO_F16 = 16 };
enum { WG_IO_G = 1, WG_IO_G2 = 2, WG_IO_X = 4 };
__host__ __device__ __forceinline__ int pw_pitch16(int T, int B) { return v100_pitch16(T, B); }
// element q of a run of bf16 values held as dwords (two per dword, low half first)
template <class V>
__device__ __forceinline__ float pw_bf16_at(const V& r, int q) {
    const unsigned w = r[q >> 1];
    return __builtin_bit_cast(float, (q & 1) ? (w & 0xffff0000u) : (w << 16));
}

__device__ __forceinline__ int pw_tap_shift(unsigned shifts, int tap) { return (int)((shifts >> (4 * tap)) & 15u); }

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
    // RAW load: the caller masks invalid elements later (mask4), at the point of use -- a select right
    // here would be a use of the loaded value and pull the vmcnt wait up to the issue point.
    f32x4 v;
    if constexpr (VEC) {
        const bool ok = row_ok && i0 < n;
        v = *reinterpret_cast<const f32x4*>(base + (ok ? row_off + i0 : 0));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = row_ok && (i0 + e) < n;
            v[e] = base[ok ? row_off + i0 + e : 0];
        }
    }
    return v;
}

__device__ __forceinline__ f32x4 mask4(f32x4 v, int i0, int n, bool row_ok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (row_ok && (i0 + e) < n) ? v[e] : 0.f;
    return v;
}

// per-channel coefficient, address clamped; the value is only meaningful when ok (callers zero the
// element it multiplies otherwise)
__device__ __forceinline__ float ldc(const float* __restrict__ c, int i, bool ok, float dflt) {
    (void)dflt;
    return c[ok ? i : 0];
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
template <int EPI_>
__device__ __forceinline__ void pw_epilogue(const PwParams& p, f32x16 (&acc)[2][2], int b, int m0, int t0, int tt, int wm, int wn,
                                            int lane, float (*red)[2][64][2]) {
    const int epi = EPI_ >= 0 ? EPI_ : p.epi_mode;
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

// Sum over the 32 lanes that share lane>>5, on the DPP path (5 VALU ops, no LDS crossbar).  The total is
// valid in the upper 16 lanes of each half (lanes 16-31 and 48-63).
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));  // row_bcast:15 into rows 1,3
    return v;
}

// Fast epilogue for a tile that lies wholly inside the tensor (every row < M, every column < T, no bias): the general
// epilogue below spends ~60 VALU instructions per pass on bounds selects, 64-bit address arithmetic and per-pass branches and
// is bound by instruction ISSUE (~1000 VALU per wave and tile: 3-5 us of the 16 us a 256 x 128 x 512 tile takes), not by
// memory.  Here: buffer addressing (one per-lane offset, the row advance is a scalar), no masks, R / coefficient loads of
// all 16 passes in flight before the accumulators go through LDS, statistics written once at the end.
// PT: the tile's rows all lie inside the tensor but its last columns do not (t0 + 128 > T: the last t-tile of a time-stretched length):
// same path with a per-lane column test -- lanes past T neither store nor count; a lane that straddles T masks per element.
// cache policy of the epilogue's big streams (A/B: -DPW_EPI_CP_Y=2 / -DPW_EPI_CP_R=2 = nontemporal): Y is written once and read by
// the NEXT kernel, R is read once
#ifndef PW_EPI_CP_Y
#define PW_EPI_CP_Y 0
#endif
#ifndef PW_EPI_STATS_LDS
#define PW_EPI_STATS_LDS 1    /* lean epilogue: BatchNorm partial sums reduced through the parked tile's LDS rows, not by DPP per pass */
#endif
#ifndef PW_EPI_CP_R
#define PW_EPI_CP_R 0
#endif
template <int EPI_, int BM, int IO, bool PT = false>
struct PwEpilogueFull {
    static constexpr bool YB = (IO & PW_IO_Y) != 0, RB = (IO & PW_IO_R) != 0, YF16 = (IO & PW_IO_F16) != 0;
    static_assert(!(YF16 && RB), "fp16 storage: inference combinations only (no R tensor in 16 bits)");
    typedef unsigned int epi_u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int epi_u32x4 __attribute__((ext_vector_type(4)));
    static constexpr int epi = EPI_;
    static constexpr bool do_stats = (epi == PW_EPI_STATS || epi == PW_EPI_MASK_STATS);
    static constexpr bool use_e = (epi == PW_EPI_AFFINE_RELU6 || epi == PW_EPI_AFFINE_RES || epi == PW_EPI_MASK_STATS);
    static constexpr bool may_r = (epi == PW_EPI_MASK_STATS || epi == PW_EPI_ADD || epi == PW_EPI_AFFINE_RES);
    static constexpr int RPP = BM / 16;             // rows per pass: one row per half-wave
    static constexpr int EY = YB ? 2 : 4, ER = RB ? 2 : 4;
    using RReg = std::conditional_t<RB, epi_u32x2, epi_u32x4>;
    RReg rpre[may_r ? 16 : 1];
    float eav[use_e ? 16 : 1], ebv[use_e ? 16 : 1];
    bool use_r;
    int voY, stepY, mrow, tcol;

    // part 1: request the R tile and the per-row coefficients of all 16 passes
    __device__ __forceinline__ void issue(const PwParams& p, int b, int m0, int t0, int tid) {
        use_r = (epi == PW_EPI_MASK_STATS || epi == PW_EPI_ADD || (epi == PW_EPI_AFFINE_RES && p.R != nullptr));
        const int P16 = pw_pitch16(p.T, p.B);
        const int lane = tid & 63, col = lane & 31, half = lane >> 5, wave = tid >> 6;
        mrow = m0 + wave * 2 + half;            // this lane's row in pass 0
        tcol = t0 + col * 4;                    // this lane's first column
        const int PY = YB ? P16 : p.T, PR = RB ? P16 : p.T;
        const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(use_r ? p.R : p.X), 0,
                                                                           use_r ? (int)((size_t)p.B * p.M * PR * ER) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rEa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(use_e ? p.ea : p.X), 0, use_e ? p.M * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rEb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(use_e ? p.eb : p.X), 0, use_e ? p.M * 4 : 0, 0x00020000);
        voY = ((b * p.M + mrow) * PY + t0 + col * 4) * EY;
        stepY = RPP * PY * EY;
        const int voR = ((b * p.M + mrow) * PR + t0 + col * 4) * ER;
        const int stepR = RPP * PR * ER;
        if constexpr (may_r) {
            if (use_r) {
#pragma unroll
                for (int pass = 0; pass < 16; ++pass) {
                    if constexpr (RB) rpre[pass] = __builtin_amdgcn_raw_buffer_load_b64(rR, voR, pass * stepR, PW_EPI_CP_R);
                    else rpre[pass] = __builtin_amdgcn_raw_buffer_load_b128(rR, voR, pass * stepR, PW_EPI_CP_R);
                }
            }
        }
        if constexpr (use_e) {
#pragma unroll
            for (int pass = 0; pass < 16; ++pass) {
                eav[pass] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rEa, mrow * 4, pass * RPP * 4, 0));
                ebv[pass] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rEb, mrow * 4, pass * RPP * 4, 0));
            }
        }
    }

    // part 2: accumulators through LDS, 16 row passes, statistics
    __device__ __forceinline__ void finish(const PwParams& p, f32x16 (&acc)[2][2], float* ct, int b, int tt, int wm, int wn, int tid) {
        const int P16 = pw_pitch16(p.T, p.B);
        const int lane = tid & 63, col = lane & 31, half = lane >> 5, wave = tid >> 6;
        const int PY = YB ? P16 : p.T;
        const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)((size_t)p.B * p.M * PY * EY), 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    ct[row * 128 + wn * 64 + j * 32 + col] = acc[i][j][r];
                }
        __syncthreads();
        float sv0[do_stats ? 16 : 1], sv1[do_stats ? 16 : 1];
        const float* crow = ct + (wave * 2 + half) * 128 + col * 4;
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(crow + pass * RPP * 128);
            f32x4 rv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (may_r) {
                if (use_r) {
                    if constexpr (RB) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) rv[e] = pw_bf16_at(rpre[pass], e);
                    } else {
                        rv = __builtin_bit_cast(f32x4, rpre[pass]);
                    }
                }
            }
            f32x4 v;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = !PT || tcol + e < p.T;          // PT only: columns past T contribute nothing
                float x = a[e];
                const float r = (PT && !ok) ? 0.f : rv[e];
                if constexpr (epi == PW_EPI_STATS) {
                    if (ok) { s0 += x; s1 = fmaf(x, x, s1); }
                } else if constexpr (epi == PW_EPI_AFFINE_RELU6) {
                    x = relu6f(fmaf(x, eav[pass], ebv[pass]));
                } else if constexpr (epi == PW_EPI_AFFINE_RES) {
                    x = fmaf(x, eav[pass], ebv[pass]) + r;
                } else if constexpr (epi == PW_EPI_MASK_STATS) {
                    const float pre = fmaf(r, eav[pass], ebv[pass]);
                    x = (pre > 0.f && pre < 6.f) ? x : 0.f;
                    if (ok) { s0 += x; s1 = fmaf(x, r, s1); }
                } else if constexpr (epi == PW_EPI_ADD) {
                    x += r;
                }
                v[e] = x;
            }
            if constexpr (!(PW_ABLATE & 4)) {
                if constexpr (YB) {
                    // (PT: the pitch keeps a straddling lane's 8 bytes inside the row; columns past T are padding)
                    if (!PT || tcol < p.T) {
                        const epi_u32x2 o2 = {pack16<YF16>(v[0], v[1]), pack16<YF16>(v[2], v[3])};
                        __builtin_amdgcn_raw_buffer_store_b64(o2, rY, voY, pass * stepY, PW_EPI_CP_Y);
                    }
                } else if (PT && tcol + 3 >= p.T) {
                    float* yq = reinterpret_cast<float*>(reinterpret_cast<char*>(p.Y) + (size_t)pass * stepY + (unsigned)voY);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (tcol + e < p.T) yq[e] = v[e];
                } else {
                    // NOT a buffer store: `buffer_store_dwordx4 v[96:99], v65, s[0:3], s4 offen` directly followed by a VALU write of
                    // v99 stored the NEW v99 on gfx950 (sporadic wrong 4th elements) -- hipcc's hazard recognizer assumes that a
                    // > 64-bit MUBUF store with an SGPR soffset needs no wait states before its data registers are overwritten.
                    // For FLAT / global stores it pads.
                    f32x4u* yq = reinterpret_cast<f32x4u*>(reinterpret_cast<char*>(p.Y) + (size_t)pass * stepY + (unsigned)voY);
                    if constexpr (PW_EPI_CP_Y != 0) __builtin_nontemporal_store((f32x4u)v, yq);
                    else *yq = v;
                }
            }
            if constexpr (do_stats) {
                if constexpr (PW_EPI_STATS_LDS) {
                    // this lane's partial sums -> the (consumed) first two floats of its own slot in the row just read
                    *reinterpret_cast<float2*>(const_cast<float*>(crow) + pass * RPP * 128) = make_float2(s0, s1);
                } else {
                    sv0[pass] = half_wave_sum_dpp(s0);
                    sv1[pass] = half_wave_sum_dpp(s1);
                }
            }
        }
        if constexpr (do_stats) {
            const size_t part = (size_t)b * p.n_ttiles + tt;
            float* sp = p.stats + (part * p.M + mrow) * 2;
            if constexpr (PW_EPI_STATS_LDS) {
                // Row sums WITHOUT a cross-lane reduction per pass (14 DPP instructions a pass = 224 of a wave's ~800 epilogue
                // instructions, and the epilogue is bound by VALU issue): the 32 lanes of a half-wave parked their partials in the
                // half-wave's own 16 rows; now lane l sums 16 of the 32 partials of row (l & 15) -- a rotated start per row keeps
                // the 16 rows' reads on 16 different bank groups --, the two halves meet with one exchange, lanes 0-15 store one
                // row's statistics each.  Fixed order: deterministic.
                const int c = col & 15, base = col & 16;
                const float* rp = ct + (wave * 2 + half + c * RPP) * 128;
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float2 v2 = *reinterpret_cast<const float2*>(rp + (base + ((i + c) & 15)) * 4);
                    t0 += v2.x; t1 += v2.y;
                }
                t0 += __shfl_xor(t0, 16, 64);
                t1 += __shfl_xor(t1, 16, 64);
                if (col < 16) {
                    const epi_u32x2 o2 = {__builtin_bit_cast(unsigned, t0), __builtin_bit_cast(unsigned, t1)};
                    *reinterpret_cast<epi_u32x2*>(sp + c * RPP * 2) = o2;
                }
            } else if (col == 31) {
#pragma unroll
                for (int pass = 0; pass < 16; ++pass) {
                    const epi_u32x2 o2 = {__builtin_bit_cast(unsigned, sv0[pass]), __builtin_bit_cast(unsigned, sv1[pass])};
                    *reinterpret_cast<epi_u32x2*>(sp + pass * RPP * 2) = o2;
                }
            }
        }
    }
};

// block-uniform: does the tile at (m0, t0) take the lean epilogue?
__device__ __forceinline__ bool pw_tile_is_full(const PwParams& p, int BM, int m0, int t0) {
    return PW_EPI_FAST != 0 && t0 + 128 <= p.T && m0 + BM <= p.M && p.bias == nullptr;
}
// ... or at least in its rows (the last t-tile of a length that is not a multiple of 128)
__device__ __forceinline__ bool pw_tile_rows_full(const PwParams& p, int BM, int m0) {
    return PW_EPI_FAST >= 2 && m0 + BM <= p.M && p.bias == nullptr;
}

// Epilogue through LDS: the 128x128 fp32 accumulator tile is parked in the (now idle) 64 KB staging buffers,
// then every half-wave streams one output row per pass as 16-byte accesses (R read, Y write: 512 B contiguous
// per row) and reduces the row's BatchNorm partial sums with DPP.  `ct` = 128*128 floats of LDS.
template <int EPI_, int BM, int IO = 0>
__device__ __forceinline__ void pw_epilogue_lds(const PwParams& p, f32x16 (&acc)[2][2], float* ct, int b, int m0, int t0, int tt,
                                                int wm, int wn, int tid) {
    constexpr bool YB = (IO & PW_IO_Y) != 0, RB = (IO & PW_IO_R) != 0, YF16 = (IO & PW_IO_F16) != 0;
    typedef unsigned int epi_u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int epi_u32x4 __attribute__((ext_vector_type(4)));
    const int P16 = pw_pitch16(p.T, p.B);
    constexpr int epi = EPI_;
    if (pw_tile_is_full(p, BM, m0, t0)) {
        PwEpilogueFull<EPI_, BM, IO> ef;
        ef.issue(p, b, m0, t0, tid);
        ef.finish(p, acc, ct, b, tt, wm, wn, tid);
        return;
    }
    if (pw_tile_rows_full(p, BM, m0)) {
        PwEpilogueFull<EPI_, BM, IO, true> ef;
        ef.issue(p, b, m0, t0, tid);
        ef.finish(p, acc, ct, b, tt, wm, wn, tid);
        return;
    }
    const int lane = tid & 63, col = lane & 31, half = lane >> 5;
    constexpr bool do_stats = (epi == PW_EPI_STATS || epi == PW_EPI_MASK_STATS);
    constexpr bool use_e = (epi == PW_EPI_AFFINE_RELU6 || epi == PW_EPI_AFFINE_RES || epi == PW_EPI_MASK_STATS);
    const bool use_r = (epi == PW_EPI_MASK_STATS || epi == PW_EPI_ADD || (epi == PW_EPI_AFFINE_RES && p.R != nullptr));
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(use_r ? p.R : p.X), 0,
                                                                       (int)(RB ? (size_t)p.B * p.M * P16 * 2 : (size_t)p.B * p.M * p.T * 4),
                                                                       0x00020000);
    const int wave = tid >> 6;
    const int t = t0 + col * 4;
    const size_t part = (size_t)b * p.n_ttiles + tt;
    constexpr int RPP = BM / 16;             // rows per pass: one row per half-wave, BM/64*2 waves
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                ct[row * 128 + wn * 64 + j * 32 + col] = acc[i][j][r];
            }
    __syncthreads();
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int row = pass * RPP + wave * 2 + half;
        const int m = m0 + row;
        const bool mv = m < p.M;
        const f32x4 a = *reinterpret_cast<const f32x4*>(ct + row * 128 + col * 4);
        const float ea = use_e ? p.ea[mv ? m : 0] : 1.f;
        const float eb = use_e ? p.eb[mv ? m : 0] : 0.f;
        const float bs = p.bias ? p.bias[mv ? m : 0] : 0.f;
        const size_t o = ((size_t)b * p.M + m) * p.T + t;
        f32x4 rv = {0.f, 0.f, 0.f, 0.f};
        const size_t o16 = ((size_t)b * p.M + m) * P16 + t;        // element offset in a bf16 (pitched) tensor
        if (use_r) {
            if constexpr (RB) {
                const epi_u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rR, (mv && t < p.T) ? (int)(o16 * 2) : 0x7ffffff0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[e] = pw_bf16_at(r2, e);
            } else {
                rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, (mv && t < p.T) ? (int)(o * 4) : 0x7ffffff0, 0, 0));
            }
        }
        f32x4 v;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = mv && (t + e) < p.T;
            float x = a[e] + bs;
            const float r = ok ? rv[e] : 0.f;
            if constexpr (epi == PW_EPI_STATS) {
                if (ok) { s0 += x; s1 = fmaf(x, x, s1); }
            } else if constexpr (epi == PW_EPI_AFFINE_RELU6) {
                x = relu6f(fmaf(x, ea, eb));
            } else if constexpr (epi == PW_EPI_AFFINE_RES) {
                x = fmaf(x, ea, eb) + r;
            } else if constexpr (epi == PW_EPI_MASK_STATS) {
                const float pre = fmaf(r, ea, eb);
                x = (pre > 0.f && pre < 6.f) ? x : 0.f;
                if (ok) { s0 += x; s1 = fmaf(x, r, s1); }
            } else if constexpr (epi == PW_EPI_ADD) {
                x += r;
            }
            v[e] = x;
        }
        if (mv && !(PW_ABLATE & 4)) {
            if constexpr (YB) {
                // 4 bf16 = one 8-byte store; the pitch keeps it aligned, columns past T inside the pitch are padding
                if (t < p.T) {
                    const epi_u32x2 o2 = {pack16<YF16>(v[0], v[1]), pack16<YF16>(v[2], v[3])};
                    *reinterpret_cast<epi_u32x2*>(reinterpret_cast<u16*>(p.Y) + o16) = o2;
                }
            } else if (t + 3 < p.T) *reinterpret_cast<f32x4u*>(p.Y + o) = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t + e < p.T) p.Y[o + e] = v[e];
            }
        }
        if constexpr (do_stats) {
            s0 = half_wave_sum_dpp(s0);
            s1 = half_wave_sum_dpp(s1);
            if (col == 31 && mv) {
                p.stats[(part * p.M + m) * 2 + 0] = s0;
                p.stats[(part * p.M + m) * 2 + 1] = s1;
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

// the same for a virtual tile index v of `total` (persistent workgroups: v = blockIdx.x + i * gridDim.x; gridDim.x % 8 == 0
// keeps every tile of a workgroup on the workgroup's own XCD)
__device__ __forceinline__ void pw_work_v(const PwParams& p, int v, int total, int& b, int& tt, int& mt) {
    const int w = xcd_remap(v, total);
    mt = w % p.n_mtiles;
    const int rest = w / p.n_mtiles;
    tt = rest % p.n_ttiles;
    b = rest / p.n_ttiles;
}

struct WgParams {
    const float* G;  const float* G2;  const float* ga; const float* gb; const float* gc;   // A operand [B][M][T], coeffs [M]
    const float* X;  const float* xa; const float* xb;                                        // B operand [B][K][T], coeffs [K]
    float* partial;  // [S][M][K]
    int B, M, K, T, S, g_mode, x_mode, n_mtiles, n_ktiles;
    // tap-addressed X rows (see PwParams): column k = tap * cx + c of dW is contracted against row c of the padded
    // X [B][cx][Tx] shifted by shift(tap); G may itself sit in a padded buffer: row pitch Tg, first column g_off.
    int ntap, cx, Tx, Tg, g_off;
    unsigned shifts;
    int io16;        // WG_IO_* mask: G / G2 / X are bf16 [B][rows][pw_pitch16(T)] (see PwParams::io16)
};


// The contraction range of split s.  S <= B: a run of whole utterances (all their t-steps).  S > B (round 5: small weight
// matrices -- the 64 -> 256 opener, heads -- have so few (m, k) tiles that even one utterance per workgroup leaves most of the
// chip idle): S = B * TS, split s = (utterance s / TS, chunk s % TS of that utterance's nt t-steps).  Step i of the split is
// utterance b_lo + i / ntl at t-step t_first + i % ntl; nsteps = nb * ntl.  ntl >= 1 always (an empty split has nb = 0).
struct WgSpan { int b_lo, nb, t_first, ntl; };
__device__ __forceinline__ WgSpan wg_span(const WgParams& p, int s, int nt) {
    WgSpan sp;
    if (p.S <= p.B) {
        const int bper = (p.B + p.S - 1) / p.S;
        sp.b_lo = s * bper;
        const int b_hi = min(p.B, sp.b_lo + bper);
        sp.nb = b_hi > sp.b_lo ? b_hi - sp.b_lo : 0;
        sp.t_first = 0;
        sp.ntl = nt > 0 ? nt : 1;
        if (nt <= 0) sp.nb = 0;
    } else {
        const int TS = p.S / p.B;                     // the launcher guarantees S % B == 0
        const int c = s % TS;
        sp.b_lo = s / TS;
        const int lo = (int)((long)c * nt / TS), hi = (int)((long)(c + 1) * nt / TS);
        sp.t_first = lo;
        sp.ntl = hi > lo ? hi - lo : 1;
        sp.nb = (hi > lo && sp.b_lo < p.B) ? 1 : 0;
    }
    return sp;
}

// work item -> (split, m-tile, k-tile), k-tile fastest: one split's tiles sit on one XCD
__device__ __forceinline__ void wg_work(const WgParams& p, int& s, int& mt, int& kt) {
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    kt = w % p.n_ktiles;
    const int rest = w / p.n_ktiles;
    mt = rest % p.n_mtiles;
    s = rest / p.n_mtiles;
}


// Mode template parameters: >= 0 fixes the mode at compile time (branch-free staging, the fast
// kernels); -1 reads it from the params at run time (the generic kernel for odd shapes).
#define PW_MODE(TPL, RUNTIME) ((TPL) >= 0 ? (TPL) : (RUNTIME))
