// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the Voice100 hot path.
// Wave = 64 lanes everywhere in this tree; nothing here is portable on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define V100_OK 0
#define V100_ERR_SHAPE 1       // invalid / unsupported shape or argument
#define V100_ERR_LAUNCH 2      // hipGetLastError() after a launch
#define V100_ERR_NULL 3        // required pointer is null

#define V100_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 4-byte-aligned variant: global memory on gfx950 takes dword-aligned dwordx4 accesses, so rows whose length is
// not a multiple of 4 can still be streamed with 16-byte loads/stores (the tail is masked by the caller)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// every kernel launch of the library goes through these two: a process-wide launch counter (v100_launch_count(), read by bench.py
// to report launches per step from THIS run rather than from a stored profile)
#include <atomic>
#include <hip/hip_ext.h>
inline std::atomic<long long> g_v100_launches{0};
// ... and, when bench.py's step-level pass asks for it (v100_timing_enable with the "other" bit), every launch that is not timed under
// a family tag of its own carries an event pair in its dispatch packet (hipExtLaunchKernelGGL), so that the step's WHOLE kernel time
// is read inside the run (roofline_step.kernel_ms) and not from a stored profile.  Off: one relaxed load per launch.
extern "C" int v100_timing_other(void** a, void** b);      // timing.hip: 1 and an event pair when the "other" tag is on
#define V100_GGL(kernel, grid, block, shmem, st, ...) do {                                                                 \
        g_v100_launches.fetch_add(1, std::memory_order_relaxed);                                                             \
        void *ta_ = nullptr, *tb_ = nullptr;                                                                                 \
        if (v100_timing_other(&ta_, &tb_)) hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, (hipEvent_t)ta_, (hipEvent_t)tb_, 0, ##__VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, st, ##__VA_ARGS__);                                              \
    } while (0)
#define V100_EXT_GGL(...) do { g_v100_launches.fetch_add(1, std::memory_order_relaxed); hipExtLaunchKernelGGL(__VA_ARGS__); } while (0)

static inline int v100_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? V100_OK : V100_ERR_LAUNCH;
}

__device__ __forceinline__ float relu6f(float v) {
    return __builtin_fminf(__builtin_fmaxf(v, 0.0f), 6.0f);   // v_med3-able clamp
}

// full-wave sum, result in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Full-wave sum on the DPP path: 6 VALU adds, no LDS crossbar (a __shfl_xor is a ds_bpermute: an LDS-pipe round trip
// per step).  The total is valid in lanes 48-63 only.
__device__ __forceinline__ float wave_sum_dpp_hi(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));  // row_bcast:15 into rows 1,3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true));  // row_bcast:31 into rows 2,3
    return v;
}

#ifndef PACK_BF16_ASM
#define PACK_BF16_ASM 0
#endif
// fp32 -> bf16 round-to-nearest-even; plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ u16 f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(u16, h);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 (RNE, NaN-preserving).  As a VECTOR conversion, not inline asm: hipcc emits the same instruction, can
    // schedule it, and pads the MFMA-result -> VALU-read hazard that it does not see through an asm statement
#if PACK_BF16_ASM
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#else
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    const pk_f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, pk_bf16x2));
#endif
}

// 16-bit operand formats of the MFMA GEMMs: bf16 (training + inference) or IEEE fp16 (inference), fp32 accumulate
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ unsigned pack16(float lo, float hi) {
    if constexpr (F16) {
        const f16x2 h = {(_Float16)lo, (_Float16)hi};            // round to nearest even (v_cvt_f16_f32 x2 + v_pack_b32_f16)
        return __builtin_bit_cast(unsigned, h);
    } else {
        return pack_bf16(lo, hi);
    }
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ u16 f2h(float f) {
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(u16, h);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Row pitch, in elements, of a 16-bit-stored activation tensor [B][C][P] ("act16") -- THE one place the rule lives
// (round 6; it used to be written out as (T + 7) & ~7 at ~50 sites).  Every row starts on a 16-byte boundary (P % 8 == 0: the kernels'
// 8- and 16-byte accesses stay aligned whatever T a time-stretch produced); rows of 256 samples or more start on a 128-BYTE boundary
// (P % 64 == 0): a 1136-byte row (T' = 563, the stretch-110 % step) shares its first and last cache line with its neighbours in memory --
// other channels, i.e. other workgroups on other XCDs -- and every shared line is fetched twice: measured 1.07-1.10 x algorithmic HBM
// traffic on the depthwise launches of those steps and 8-10 % of their time (profiles/r06_dw_ab.txt: T = 568 -> 51.4 %, 576 -> 55.4 % of
// 8 TB/s).  A tensor of ONE row per channel (B == 1: the channel-major inference matrices [C][B P], which the GEMMs see as B = 1,
// T = B P) keeps the 8-sample rule: its row length is the caller's column count and must not be re-padded.
#ifndef V100_PITCH_LINES
#define V100_PITCH_LINES 1     /* 0: the rounds 2-5 rule, (T + 7) & ~7 everywhere (A/B builds) */
#endif
__host__ __device__ __forceinline__ int v100_pitch16(int T, int B) {
    return (V100_PITCH_LINES && B > 1 && T >= 256) ? ((T + 63) & ~63) : ((T + 7) & ~7);
}
// (exported to the host side as v100_row_pitch16, block.hip)
