// K2 on the matrix cores: stride-1 depthwise Conv1d as a banded-Toeplitz product, fp32 result.
//
// Why: the register-window kernel (dwconv_kernel) needs 2*K flop per output on the fp32 VALU; for K >= 51 that is
// the same time as (or more than) streaming the row from HBM, so the layer cannot reach the memory roofline whatever
// the schedule.  The matrix pipe has 16x the VALU's rate in bf16, and an fp32 value is
// EXACTLY the sum of three bf16 values (8 + 8 + 8 mantissa bits, split by truncation), so
//     x * w = sum_{i,j} x_i * w_j,   every bf16 x bf16 product exact in the fp32 accumulator,
// and keeping the six terms with i + j <= 4 drops only terms below 2^-24 of the product: fp32 accuracy at 6 MFMAs per
// 16x16x32 tile step.  For K = 83 that is 1.5 matrix-pipe cycles per output per SIMD against ~4 on the VALU, and the
// VALU is left with the staging work only, so every layer of the encoder becomes HBM-bound.
//
// Formulation.  One wave owns a (batch row, 512-output tile) item of its channel.  The staged input span is kept in LDS as
// NT bf16 "digit" images img_t[i] (i = input position - in0a, in0a = tile start - pad rounded down to a multiple of 4 so
// that the global float4 loads and the 8-byte LDS stores stay aligned).  With off = (-pad) & 3,
//     y[out0 + 256*sub + 16*n + m] = sum_kk A[m][kk] * B[kk][n],   A[m][kk] = w[kk - m - off],   B[kk][n] = img[256*sub + 16*n + kk]
// for m, n in [0, 16), kk in [0, 32*STEPS): per (sub, step) one v_mfma_f32_16x16x32_bf16 per digit pair.  A (the channel's
// taps) is built once per workgroup and lives in registers; B fragments are conflict-free ds_read_b128 of the images
// (lane (n, q) reads 16 bytes at 32*n + 16*q + 64*step + 512*sub); the 16x16 result has 4 consecutive outputs per lane,
// so a wave stores 1 KiB contiguous per instruction.  Same prologues / epilogues / per-channel partial sums as dwconv_kernel.
//
// WG (fused backward, stride 1): the backward-weight of the forward conv rides on the backward-data pass, also on the
// matrix pipe.  In that pass the images hold the upstream gradient g' and every lane loads the forward conv's
// pre-activation input a1 at its output positions (for the ReLU6 mask), so xin = relu6(bn1(a1)) is at hand.  With
// e = 16*p + r the position inside the tile,
//     dWf[jf] = sum_e xin[e] * g'img[off + e + jf] = sum_r E[r + jf + off][r],   E[i][r] = sum_p g'img[16*p + i] * ximg[16*p + r]:
// E is a plain matrix product over the block index p of the two images viewed as [32 x 16] row-major tiles, i.e. both
// MFMA operands are COLUMN reads of row-major bf16 tiles -- ds_read_b64_tr_b16 (address = image + 8 bytes * lane, the row
// shift 16*ib of the g' operand is 32 bytes: aligned).  One contraction step covers the whole 512-position tile; E
// ((K+18) x 16 fp32) stays in the accumulators for all rows of the workgroup and its diagonals are summed once at the end.
// K+15 over K more work than the K*T products needed, no extra HBM traffic, no K accumulators per lane.
#pragma once
#include <type_traits>

typedef float dwm_f32x4 __attribute__((ext_vector_type(4)));
typedef short dwm_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int dwm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int dwm_u32x2 __attribute__((ext_vector_type(2)));
typedef short dwm_s16x4 __attribute__((ext_vector_type(4)));

// Row stride (floats) of the E tiles' exchange buffer.  Thread jf sums the diagonal E[r + jf + off][r]: with 16-float rows the lanes of a
// ds_read_b32 sit 16 dwords apart -- two banks for 32 lanes, a 16-way conflict on every one of the 64 reads per thread (round 6: this
// was most of the fused backward's 39-49 % SQ_LDS_BANK_CONFLICT) -- with 17 they walk the banks.
#ifndef DWM_EP
#define DWM_EP 17
#endif

// 8 bf16 of one MFMA operand fragment as two transposing LDS reads (rows 4G..4G+3 and 16+4G..16+4G+3 of a [32 x 16] tile,
// G = lane >> 4, column lane & 15); `tile` points at element 0 of the tile, 8-byte aligned.  EXEC must be all ones.
__device__ __forceinline__ dwm_bf16x8 dwm_tr_fragment(const unsigned short* tile, int lane) {
    typedef __attribute__((address_space(3))) dwm_s16x4 lds_s16x4;
    const dwm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + 4 * lane));
    const dwm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + 256 + 4 * lane));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// acc += sum over the digit pairs (i, j), i < NA, j < NB, i + j <= max(NA, NB) - 1, of a[i] x b[j], smallest terms first
// (3 x 3 digits: six products, fp32-exact; 3 x 1: three products, exact for a bf16-valued b)
// F16: the fragments hold IEEE fp16 digits (inference at precision "fp16": fp16-stored activations against fp16 digits of the taps)
template <int NA, int NB, bool F16 = false>
__device__ __forceinline__ dwm_f32x4 dwm_mfma_digits(const dwm_bf16x8 (&a)[NA], const dwm_bf16x8 (&b)[NB], dwm_f32x4 acc) {
    constexpr int LIM = (NA > NB ? NA : NB) - 1;
#pragma unroll
    for (int s = LIM; s >= 0; --s)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int j = s - i;
            if (j >= 0 && j < NB) {
                if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc, 0, 0, 0);
            }
        }
    return acc;
}
// fp16 digits of an fp32 value (round to nearest): v ~ d0 + d1 + d2, 11 + 11 (+ what fp16's range leaves) mantissa bits; as 16-bit patterns
__device__ __forceinline__ void dwm_split_f16(float v, unsigned (&d)[3]) {
    const _Float16 h0 = (_Float16)v;
    const float r1 = v - (float)h0;
    const _Float16 h1 = (_Float16)r1;
    const float r2 = r1 - (float)h1;
    const _Float16 h2 = (_Float16)r2;
    d[0] = __builtin_bit_cast(unsigned short, h0);
    d[1] = __builtin_bit_cast(unsigned short, h1);
    d[2] = __builtin_bit_cast(unsigned short, h2);
}

// OFFMAX = 3: image element 0 sits at a position that is a multiple of 4 (4-sample loads); 7: a multiple of 8 (8-sample, 16-byte
// loads of bf16 rows).  off = (-pad) & OFFMAX is folded into the Toeplitz block, which therefore spans K + 15 + OFFMAX columns.
template <int K, int OFFMAX = 3>
struct DwMfmaGeom {
    static constexpr int SUBS = 2;
    static constexpr int TILE = 256 * SUBS;                       // outputs per wave item
    static constexpr int STEPS = (K + 15 + OFFMAX + 31) / 32;     // contraction steps of 32
    static constexpr int IMG = 256 * (SUBS - 1) + 240 + 32 * STEPS;   // image elements the B reads touch
    static constexpr int NV = (IMG + 255) / 256;                  // 4-sample loads per lane
    static constexpr int NV8 = (IMG + 511) / 512;                 // 8-sample loads per lane (all inputs bf16)
    static constexpr int IMGP = NV * 256 > NV8 * 512 ? NV * 256 : NV8 * 512;   // image elements written (all finite: zero or real samples)
    static constexpr int WPAD = 15 + OFFMAX;                      // zero taps in front of w[0] (m + off <= WPAD)
    static constexpr int WLEN = WPAD + 32 * STEPS;
    static constexpr int IB = (K + 15 + OFFMAX + 15) / 16;        // 16-row blocks of E (fused backward-weight)
};

// bf16 digit t of an fp32 value by truncation: v = d0 + d1 + d2 exactly (24 mantissa bits), each digit's low 16 bits zero
__device__ __forceinline__ void dwm_split(float v, unsigned (&d)[3]) {
    const unsigned u0 = __builtin_bit_cast(unsigned, v);
    d[0] = u0 & 0xffff0000u;
    const float r1 = v - __builtin_bit_cast(float, d[0]);
    const unsigned u1 = __builtin_bit_cast(unsigned, r1);
    d[1] = u1 & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, d[1]);
    d[2] = __builtin_bit_cast(unsigned, r2) & 0xffff0000u;
}
// two fp32 -> one dword of two bf16, round to nearest even, by plain casts (hipcc emits v_cvt_pk_bf16_f32 and -- unlike for an
// inline-asm statement -- its hazard recognizer pads the MFMA-result -> VALU-read wait states: common.h's pack_bf16 asm read
// the accumulators too early here and stored stale registers)
typedef __bf16 dwm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dwm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dwm_pack_rne(float lo, float hi) {
    // a VECTOR conversion: one v_cvt_pk_bf16_f32 (two scalar casts compile to two conversions and a v_perm)
    const dwm_f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, dwm_bf16x2));
}
// the NT digits of a TAP: NT >= 2 the exact truncation split; NT == 1 (16-bit activation storage, round 6) the tap ROUNDED to bf16 to
// nearest-even -- a truncated single digit would shrink every tap by 2^-9 on average, a gain error BatchNorm hides but a float64
// check does not; rounded it is the operand F.conv1d sees under bf16 autocast
template <int NT>
__device__ __forceinline__ void dwm_split_taps(float v, unsigned (&d)[3]) {
    if constexpr (NT == 1) { d[0] = dwm_pack_rne(v, 0.f) << 16; d[1] = 0u; d[2] = 0u; }
    else dwm_split(v, d);
}
// two digits (high halves of lo / hi) -> one dword of two bf16
__device__ __forceinline__ unsigned dwm_pack(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// element e (0..3) of a 4-sample run held either as f32x4 or as two dwords of packed bf16
__device__ __forceinline__ float dwm_elem(const f32x4& r, int e) { return r[e]; }
__device__ __forceinline__ float dwm_elem(const dwm_u32x2& r, int e) {
    const unsigned w = r[e >> 1];
    return __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
}
__device__ __forceinline__ float dwm_elem8(const dwm_u32x4& r, int e) {      // 8 bf16 in four dwords
    const unsigned w = r[e >> 1];
    return __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
}
// DWM_CP: cache policy of the bf16-stored row streams of the act16 kernels (2 = nontemporal; in-step A/B of the general kernel: +-0, profiles/r03_ab_misc.txt; the streaming kernels take theirs as a template argument;
// +16 % on the rotating-working-set micro-benchmark of the streaming kernels, profiles/r03_dw_stream_ab.txt)
#ifndef DWM_CP
#define DWM_CP 0
#endif
template <int CP = 0>
__device__ __forceinline__ dwm_u32x4 dwm_load_run8(__amdgpu_buffer_rsrc_t r, int voff_elems, unsigned row_elems) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff_elems < 0x20000000 ? voff_elems * 2 : 0x7ffffff0, (int)(row_elems * 2u), CP);
}
template <bool B16>
struct DwmRun { typedef f32x4 type; };
template <>
struct DwmRun<true> { typedef dwm_u32x2 type; };
template <bool B16, int CP = 0>
__device__ __forceinline__ typename DwmRun<B16>::type dwm_load_run(__amdgpu_buffer_rsrc_t r, int voff_elems, unsigned row_elems) {
    // voff_elems: element offset inside the row, or a huge value for "out of range" (hardware returns zero)
    if constexpr (B16) return __builtin_amdgcn_raw_buffer_load_b64(r, voff_elems < 0x20000000 ? voff_elems * 2 : 0x7ffffff0, (int)(row_elems * 2u), CP);
    else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_elems < 0x20000000 ? voff_elems * 4 : 0x7ffffff0, (int)(row_elems * 4u), 0));
}

#ifndef DWM_SINGLE
#define DWM_SINGLE 1
#endif

template <int K, int IM, int OM, int NT, bool WG = false, int IO = 0>
__global__ __launch_bounds__(256, WG ? 3 : 4) void dwconv_mfma_kernel(DwParams p) {
    constexpr bool XB = (IO & DW_IO_X) != 0, X2B = (IO & DW_IO_X2) != 0, AUXB = (IO & DW_IO_AUX) != 0, YB = (IO & DW_IO_Y) != 0;
    // every input stream bf16: a lane stages 8 consecutive samples per load (16-byte loads, 16-byte LDS stores)
    constexpr bool W8 = XB && (IM != DW_IN_AFFINE2 || X2B);
    constexpr int OFFMAX = W8 ? 7 : 3;
    // digits of the DATA operand (the staged images): with 16-bit activation storage the conv input is a bf16 tensor by
    // definition (as under autocast), so the transformed sample is rounded once (RNE) to ONE digit; the taps keep NT digits
    constexpr int NX = IO != 0 ? 1 : NT;
    using G_ = DwMfmaGeom<K, OFFMAX>;
    constexpr int SUBS = G_::SUBS, TILE = G_::TILE, STEPS = G_::STEPS, NV = G_::NV, IMGP = G_::IMGP, WPAD = G_::WPAD, WLEN = G_::WLEN;
    constexpr int IB = G_::IB;
    constexpr bool TWO = IM == DW_IN_AFFINE2;
    constexpr bool STATS = (OM == DW_OUT_RAW_STATS || OM == DW_OUT_MASK_STATS);
    static_assert(NT >= 1 && NT <= 3, "one to three bf16 digits per operand");
    static_assert(!WG || OM == DW_OUT_MASK_STATS, "the fused backward-weight rides on the backward-data pass");

    // per wave: NT digit images of the staged span (+ NT digit images of xin over the tile when WG); the E tiles of the
    // fused backward-weight are spilled over the same bytes at the very end
    constexpr int WAVE_U16 = NX * IMGP + (WG ? NX * TILE : 0);
    constexpr int E_FLOATS = WG ? 16 * IB * DWM_EP : 0;
    constexpr int LDS_BYTES = (4 * WAVE_U16 * 2 > 4 * E_FLOATS * 4) ? 4 * WAVE_U16 * 2 : 4 * E_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    __shared__ float lds_w[WLEN];
    __shared__ float lds_red[4][2];
    __shared__ float lds_coef[3];

    const int c = blockIdx.x;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_ = lane & 15, q_ = lane >> 4;

    for (int i = threadIdx.x; i < WLEN; i += 256) {
        const int j = i - WPAD;
        lds_w[i] = (j >= 0 && j < K) ? p.w[(size_t)c * K + (p.flip ? (K - 1 - j) : j)] : 0.f;
    }
    // one group (G == 1): this workgroup is the only consumer of channel c's input BatchNorm, so its first wave finalises it from
    // the producing GEMM's slab of partial sums (DwPre) while the others stage the taps
    if (p.pre.f.mode != 0 && wave == 0) dw_finalize_parts(p.pre, p.C, c, lane, lds_coef);
    __syncthreads();

    // A fragments: lane (m = n_, q_) holds A[m][32*s + 8*q + jj] = w[32*s + 8*q + jj - m - off], jj = 0..7, as NT digit packs
    const int off = (-p.pad) & OFFMAX;
    dwm_bf16x8 afr[STEPS][NT];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        unsigned pk[NT][4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            unsigned d0[3], d1[3];
            dwm_split_taps<NT>(lds_w[WPAD + 32 * s + 8 * q_ + 2 * jp - n_ - off], d0);
            dwm_split_taps<NT>(lds_w[WPAD + 32 * s + 8 * q_ + 2 * jp + 1 - n_ - off], d1);
#pragma unroll
            for (int t = 0; t < NT; ++t) pk[t][jp] = dwm_pack(d0[t], d1[t]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const dwm_u32x4 v = {pk[t][0], pk[t][1], pk[t][2], pk[t][3]};
            afr[s][t] = __builtin_bit_cast(dwm_bf16x8, v);
        }
    }

    float ca = 1.f, cb = 0.f, cc = 0.f, oa = 1.f, ob = 0.f;
    if (p.pre.f.mode != 0) { ca = lds_coef[0]; cb = lds_coef[1]; cc = lds_coef[2]; }
    else {
        if constexpr (IM != DW_IN_NONE) { ca = p.in_a[c]; cb = p.in_b[c]; }
        if constexpr (IM == DW_IN_AFFINE2) cc = p.in_c[c];
    }
    if constexpr (OM == DW_OUT_AFFINE_RELU6 || OM == DW_OUT_MASK_STATS) { oa = p.out_a[c]; ob = p.out_b[c]; }

    const int Tin = p.Tin, Tout = p.Tout;
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int ntiles = (Tout + TILE - 1) / TILE;

    float s0 = 0.f, s1 = 0.f;
    // row pitches (elements): bf16-stored tensors are padded to a multiple of 8 samples per row
    const int PinX = XB ? dw_pitch16(Tin, p.B) : Tin, PinX2 = X2B ? dw_pitch16(Tin, p.B) : Tin;
    const int PoutA = AUXB ? dw_pitch16(Tout, p.B) : Tout, PoutY = YB ? dw_pitch16(Tout, p.B) : Tout;
    constexpr int NV8 = G_::NV8;
    typename DwmRun<XB>::type rawx[W8 ? 1 : NV];
    typename DwmRun<X2B>::type rawx2[(TWO && !W8) ? NV : 1];
    dwm_u32x4 raw8[W8 ? NV8 : 1], raw8b[(W8 && TWO) ? NV8 : 1];
    const __amdgpu_buffer_rsrc_t rx = dw_make_rsrc(p.x, (unsigned)((size_t)p.B * p.C * PinX * (XB ? 2 : 4)));
    const __amdgpu_buffer_rsrc_t rx2 = TWO ? dw_make_rsrc(p.x2, (unsigned)((size_t)p.B * p.C * PinX2 * (X2B ? 2 : 4))) : rx;
    const __amdgpu_buffer_rsrc_t raux = dw_make_rsrc(OM == DW_OUT_MASK_STATS ? p.aux : p.x,
                                                     OM == DW_OUT_MASK_STATS ? (unsigned)((size_t)p.B * p.C * PoutA * (AUXB ? 2 : 4)) : 0u);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned short* img = reinterpret_cast<unsigned short*>(lds_raw) + wave_u * WAVE_U16;
    unsigned short* ximg = img + NX * IMGP;                  // WG only
    const unsigned short* bsrc = img + 16 * n_ + 8 * q_;
    dwm_f32x4 eacc[WG ? IB : 1];
#pragma unroll
    for (int ib = 0; ib < (WG ? IB : 1); ++ib) eacc[ib] = dwm_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = 0; tile < ntiles; ++tile) {
        const int out0 = tile * TILE;
        const int in0a = (out0 - p.pad) & ~OFFMAX;            // input position of image element 0 (multiple of 4, or of 8)
        // per-lane load positions of this tile (row-invariant)
        constexpr int EPL = W8 ? 8 : 4, NVL = W8 ? NV8 : NV;
        int vo[NVL];
#pragma unroll
        for (int v = 0; v < NVL; ++v) {
            const int ia = in0a + EPL * (lane + 64 * v);
            // in0a is a multiple of EPL: a run lies wholly before the row or starts inside it
            const bool ok = ia >= 0 && ia < Tin && EPL * (lane + 64 * v) < G_::IMG;
            vo[v] = ok ? ia : 0x7ffffff0;                      // element offset inside the row
        }
        // Rows that fit ONE tile (T <= 512: the benchmark's T' = 512) with all streams bf16: a lane stages exactly one run, the run
        // at positions 8 * lane ... + 7, and the image's zero padding on both sides of the row is written once per wave instead of
        // being re-derived from out-of-range loads on every row (half the staging instructions, no bounds selects).
        // Forward only: A/B on one box, 9-layer totals -- forward 167.4 -> 160.0 us, the fused backward (two input streams, xin images;
        // its own staging is not what bounds it) 347-358 -> 354-372 us.
        const bool single = DWM_SINGLE && W8 && !WG && ntiles == 1 && Tin <= TILE && (Tin & 7) == 0;        // kernel-uniform
        const int lpad = -in0a;
        if constexpr (W8) {
            if (single) {
                for (int c8 = lane; c8 < IMGP / 8; c8 += 64)
                    if (c8 < lpad / 8 || c8 >= (lpad + Tin) / 8) {
#pragma unroll
                        for (int t = 0; t < NX; ++t) *reinterpret_cast<dwm_u32x4*>(img + t * IMGP + 8 * c8) = dwm_u32x4{0u, 0u, 0u, 0u};
                    }
                asm volatile("" ::: "memory");
                vo[0] = 8 * lane < Tin ? 8 * lane : 0x7ffffff0;
#pragma unroll
                for (int v = 1; v < NVL; ++v) vo[v] = 0x7ffffff0;
            }
        }
        auto issue_loads = [&](unsigned row) {
#pragma unroll
            for (int v = 0; v < NVL; ++v) {
                if constexpr (W8) {
                    raw8[v] = dwm_load_run8<DWM_CP>(rx, vo[v], row * (unsigned)PinX);
                    if constexpr (TWO) raw8b[v] = dwm_load_run8<DWM_CP>(rx2, vo[v], row * (unsigned)PinX2);
                } else {
                    rawx[v] = dwm_load_run<XB>(rx, vo[v], row * (unsigned)PinX);
                    if constexpr (TWO) rawx2[v] = dwm_load_run<X2B>(rx2, vo[v], row * (unsigned)PinX2);
                }
            }
        };
        int bi = wave_u;
        if (bi < nb) issue_loads((unsigned)((b0 + bi) * p.C + c));
        for (; bi < nb; bi += 4) {
            const int b = b0 + bi;
            float auxv[SUBS][4];
            if constexpr (OM == DW_OUT_MASK_STATS) {
#pragma unroll
                for (int sub = 0; sub < SUBS; ++sub) {
                    const int t0 = out0 + 256 * sub + 16 * n_ + 4 * q_;
                    if constexpr (AUXB) {
                        const dwm_u32x2 a2 = dwm_load_run<true, W8 ? DWM_CP : 0>(raux, t0 < Tout ? t0 : 0x7ffffff0, (unsigned)(b * p.C + c) * (unsigned)PoutA);
#pragma unroll
                        for (int r = 0; r < 4; ++r) auxv[sub][r] = (t0 + r < Tout) ? dwm_elem(a2, r) : 0.f;
                    } else {
                        dw_load_run<4, true>(auxv[sub], p.aux + ((size_t)b * p.C + c) * Tout + t0, t0, Tout);
                    }
                }
            }
            // ---- stage: transform, zero outside the row, split into bf16 digits, 8- / 16-byte LDS stores ----
            if (W8 && single) {
                if constexpr (W8) {
                    float vals[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if constexpr (IM == DW_IN_AFFINE_RELU6) vals[e] = relu6f(fmaf(dwm_elem8(raw8[0], e), ca, cb));
                        else if constexpr (IM == DW_IN_AFFINE2) vals[e] = fmaf(dwm_elem8(raw8[0], e), ca, fmaf(dwm_elem8(raw8b[0], e), cb, cc));
                        else vals[e] = dwm_elem8(raw8[0], e);
                    }
                    const dwm_u32x4 w4 = {dwm_pack_rne(vals[0], vals[1]), dwm_pack_rne(vals[2], vals[3]), dwm_pack_rne(vals[4], vals[5]),
                                          dwm_pack_rne(vals[6], vals[7])};
                    if (8 * lane < Tin) *reinterpret_cast<dwm_u32x4*>(img + lpad + 8 * lane) = w4;     // the padding stays zero
                }
            } else if constexpr (W8) {
#pragma unroll
                for (int v = 0; v < NV8; ++v) {
                    const int ia = in0a + 8 * (lane + 64 * v);
                    float vals[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float val;
                        if constexpr (IM == DW_IN_AFFINE_RELU6) val = relu6f(fmaf(dwm_elem8(raw8[v], e), ca, cb));
                        else if constexpr (IM == DW_IN_AFFINE2) val = fmaf(dwm_elem8(raw8[v], e), ca, fmaf(dwm_elem8(raw8b[v], e), cb, cc));
                        else val = dwm_elem8(raw8[v], e);
                        vals[e] = (ia + e >= 0 && ia + e < Tin) ? val : 0.f;
                    }
                    const dwm_u32x4 w4 = {dwm_pack_rne(vals[0], vals[1]), dwm_pack_rne(vals[2], vals[3]), dwm_pack_rne(vals[4], vals[5]),
                                          dwm_pack_rne(vals[6], vals[7])};
                    *reinterpret_cast<dwm_u32x4*>(img + 8 * (lane + 64 * v)) = w4;
                }
            } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int ia = in0a + 4 * (lane + 64 * v);
                unsigned dg[4][3];
                float vals[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float val;
                    if constexpr (IM == DW_IN_AFFINE_RELU6) val = relu6f(fmaf(dwm_elem(rawx[v], e), ca, cb));
                    else if constexpr (IM == DW_IN_AFFINE2) val = fmaf(dwm_elem(rawx[v], e), ca, fmaf(dwm_elem(rawx2[v], e), cb, cc));
                    else val = dwm_elem(rawx[v], e);
                    // zero padding applies to the TRANSFORMED tensor; a float4 that straddles the row end (Tin % 4 != 0) holds the
                    // next row's first samples, so the plain copy needs the mask too (ia is a multiple of 4: ia >= 0 covers e)
                    val = (ia >= 0 && ia + e < Tin) ? val : 0.f;
                    vals[e] = val;
                    if constexpr (NX > 1) dwm_split(val, dg[e]);
                }
                if constexpr (NX == 1) {
                    const dwm_u32x2 w2 = {dwm_pack_rne(vals[0], vals[1]), dwm_pack_rne(vals[2], vals[3])};
                    *reinterpret_cast<dwm_u32x2*>(img + 4 * (lane + 64 * v)) = w2;
                } else {
#pragma unroll
                    for (int t = 0; t < NX; ++t) {
                        const dwm_u32x2 w2 = {dwm_pack(dg[0][t], dg[1][t]), dwm_pack(dg[2][t], dg[3][t])};
                        *reinterpret_cast<dwm_u32x2*>(img + t * IMGP + 4 * (lane + 64 * v)) = w2;
                    }
                }
            }
            }
            if constexpr (WG) {
                // xin = relu6(bn1(a1)) at this lane's 2 x 4 output positions -> digit images of the tile
#pragma unroll
                for (int sub = 0; sub < SUBS; ++sub) {
                    const int t0 = out0 + 256 * sub + 16 * n_ + 4 * q_;
                    unsigned dg[4][3];
                    float xv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        xv[r] = (t0 + r < Tout) ? relu6f(fmaf(auxv[sub][r], oa, ob)) : 0.f;
                        if constexpr (NX > 1) dwm_split(xv[r], dg[r]);
                    }
                    if constexpr (NX == 1) {
                        const dwm_u32x2 w2 = {dwm_pack_rne(xv[0], xv[1]), dwm_pack_rne(xv[2], xv[3])};
                        *reinterpret_cast<dwm_u32x2*>(ximg + 256 * sub + 16 * n_ + 4 * q_) = w2;
                    } else {
#pragma unroll
                        for (int t = 0; t < NX; ++t) {
                            const dwm_u32x2 w2 = {dwm_pack(dg[0][t], dg[1][t]), dwm_pack(dg[2][t], dg[3][t])};
                            *reinterpret_cast<dwm_u32x2*>(ximg + t * TILE + 256 * sub + 16 * n_ + 4 * q_) = w2;
                        }
                    }
                }
            }
            // cross-lane hand-off through LDS inside one wave (hardware keeps a wave's LDS operations in order; the
            // compiler must too)
            asm volatile("" ::: "memory");

            // prefetch the next row of this tile while this one computes
            if (bi + 4 < nb) issue_loads((unsigned)((b + 4) * p.C + c));

            if constexpr (WG) {
                // E[16*ib + m][r] += sum_p g'img[16*(p + ib) + m] * ximg[16*p + r]: one contraction step = the whole tile
                dwm_bf16x8 xfr[NX];
#pragma unroll
                for (int t = 0; t < NX; ++t) xfr[t] = dwm_tr_fragment(ximg + t * TILE, lane);
#pragma unroll
                for (int ib = 0; ib < IB; ++ib) {
                    dwm_bf16x8 gfr[NX];
#pragma unroll
                    for (int t = 0; t < NX; ++t) gfr[t] = dwm_tr_fragment(img + t * IMGP + 16 * ib, lane);
                    eacc[ib] = dwm_mfma_digits<NX, NX>(gfr, xfr, eacc[ib]);
                }
            }

#pragma unroll
            for (int sub = 0; sub < SUBS; ++sub) {
                if (out0 + 256 * sub >= Tout) break;      // wave-uniform: a time-stretched row that ends inside the first half
                const int t0 = out0 + 256 * sub + 16 * n_ + 4 * q_;
                const size_t oo = ((size_t)b * p.C + c) * Tout + t0;
                dwm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < STEPS; ++s) {
                    dwm_bf16x8 bfr[NX];
#pragma unroll
                    for (int t = 0; t < NX; ++t)
                        bfr[t] = *reinterpret_cast<const dwm_bf16x8*>(bsrc + t * IMGP + 256 * sub + 32 * s);
                    acc = dwm_mfma_digits<NT, NX>(afr[s], bfr, acc);
                }
                // ---- epilogue: 4 consecutive outputs per lane (FULL: the 256 outputs of this half all lie inside the row) ----
                float outv[4];
                auto epi = [&](auto full_t) {
                    constexpr bool FULL = decltype(full_t)::value;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool valid = FULL || t0 + r < Tout;
                        float yv = acc[r];
                        if constexpr (OM == DW_OUT_RAW_STATS) {
                            if (valid) { s0 += yv; s1 = fmaf(yv, yv, s1); }
                        } else if constexpr (OM == DW_OUT_AFFINE_RELU6) {
                            yv = relu6f(fmaf(yv, oa, ob));
                        } else if constexpr (OM == DW_OUT_MASK_STATS) {
                            const float pre = fmaf(auxv[sub][r], oa, ob);
                            yv = (pre > 0.f && pre < 6.f) ? yv : 0.f;
                            if (valid) { s0 += yv; s1 = fmaf(yv, auxv[sub][r], s1); }
                        }
                        outv[r] = yv;
                    }
                };
                if (out0 + 256 * sub + 256 <= Tout) epi(std::true_type{}); else epi(std::false_type{});
                if constexpr (YB) {
                    // 4 bf16 = one 8-byte store (the pitch keeps it aligned; samples past Tout land in the row's padding)
                    if (t0 < Tout) {
                        const dwm_u32x2 o2 = {dwm_pack_rne(outv[0], outv[1]), dwm_pack_rne(outv[2], outv[3])};
                        dwm_u32x2* yq = reinterpret_cast<dwm_u32x2*>(reinterpret_cast<unsigned short*>(p.y) + ((size_t)b * p.C + c) * PoutY + t0);
                        if constexpr (W8 && DWM_CP != 0) __builtin_nontemporal_store(o2, yq);
                        else *yq = o2;
                    }
                } else if (t0 + 3 < Tout) {
                    const f32x4 o = {outv[0], outv[1], outv[2], outv[3]};
                    *reinterpret_cast<f32x4u*>(p.y + oo) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t0 + r < Tout) p.y[oo + r] = outv[r];
                }
            }
            asm volatile("" ::: "memory");     // the next row's LDS stores stay behind this row's fragment reads
        }
    }

    if constexpr (WG) {
        // dWf[jf] = sum over waves and r of E[r + jf + off][r]; the E tiles go through LDS (over the dead images)
        __syncthreads();
        float* ebuf = reinterpret_cast<float*>(lds_raw);
#pragma unroll
        for (int ib = 0; ib < IB; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) ebuf[(wave * 16 * IB + 16 * ib + 4 * q_ + r) * DWM_EP + n_] = eacc[ib][r];
        __syncthreads();
        for (int jf = threadIdx.x; jf < K; jf += 256) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float sw = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sw += ebuf[(w * 16 * IB + r + jf + off) * DWM_EP + r];
                sum += sw;
            }
            p.wpartial[((size_t)g * p.C + c) * K + (K - 1 - jf)] = sum;      // flipped taps here = forward taps K-1-jf
        }
        __syncthreads();
    }
    if constexpr (STATS) {
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) { lds_red[wave][0] = s0; lds_red[wave][1] = s1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t0s = (lds_red[0][0] + lds_red[1][0]) + (lds_red[2][0] + lds_red[3][0]);
            const float t1s = (lds_red[0][1] + lds_red[1][1]) + (lds_red[2][1] + lds_red[3][1]);
            p.stats[((size_t)g * p.C + c) * 2 + 0] = t0s;
            p.stats[((size_t)g * p.C + c) * 2 + 1] = t1s;
            if (p.fin.mode != 0) dw_finalize(p.fin, c, t0s, t1s);      // G == 1: these ARE the channel's sums
        }
    }
}
