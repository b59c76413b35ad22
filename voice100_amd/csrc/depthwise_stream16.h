// K2 forward for the benchmark's case, built around BYTES IN FLIGHT: training-mode depthwise forward with bf16-stored
// activations (BN1 + ReLU6 on load, raw output + BN2 sums; asr.py:49) on rows that fit one 512-output tile (T <= 512, T % 8 == 0).
//
// Why a second kernel.  dwconv_mfma_kernel keeps ONE row of loads in flight per wave and issues its first load only after the
// tap prologue (taps -> LDS -> barrier -> Toeplitz fragments).  Inside a training step its input and output stream from / to HBM
// (tools/bench_dw_regimes.py: the stand-alone number that round 2 quoted has both tensors resident in the 256 MB Infinity Cache), a
// miss costs ~2 us under load, and a 2048-channel layer gives a wave only 8 rows: the kernel was latency-bound (PMC: 52-56 % of
// wave cycles parked on s_waitcnt, MFMA 5-9 %, VALU active 21-28 %) at 16 KB of loads in flight per CU, where Little's law wants
// >= 32 KB (MI355X_MICROARCH.md, "streaming").  Here a wave requests its first D rows at kernel entry -- the loads need nothing
// but the row index -- so the whole prologue runs under the memory latency, and stays D rows (D x 16 B per lane = 4 VGPRs each)
// ahead from then on: D = 4 gives 64 KB in flight per CU at 4 waves per SIMD.
//
// The main loop is branch-free (a conditional load or store makes hipcc's s_waitcnt insertion take the worst-case count at the
// join, i.e. it waits for the prefetches just issued): rows past the wave's last row load through an out-of-range buffer offset
// (the hardware returns zeros and moves no bytes), run with zeroed BatchNorm coefficients (so they add nothing to the sums) and
// store through an out-of-range offset (dropped).  The first group of D rows is peeled so that the loop is entered in the same
// queue state its back edge produces (otherwise the merged state again costs the prefetch depth).
//
// Arithmetic, LDS image, fragment reads, statistics and the in-kernel BatchNorm finalisation (G == 1) are those of
// dwconv_mfma_kernel<K, AFFINE_RELU6, RAW_STATS, NT, false, X | Y>: results are bit-identical to it.
#pragma once

template <int K, int NT, int D>
__global__ __launch_bounds__(256, 4) void dwconv_fwd16_stream_kernel(DwParams p) {
    using G_ = DwMfmaGeom<K, 7>;
    constexpr int STEPS = G_::STEPS, IMGP = G_::IMGP, WPAD = G_::WPAD, WLEN = G_::WLEN;
    __shared__ __attribute__((aligned(16))) unsigned short lds_img[4 * IMGP];
    __shared__ float lds_w[WLEN];
    __shared__ float lds_red[4][2];

    const int c = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_ = lane & 15, q_ = lane >> 4;
    const int T = p.Tin;                                     // == Tout, a multiple of 8, <= 512: the row pitch is T
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int nrows = wave < nb ? (nb - wave + 3) >> 2 : 0;  // this wave's rows: b0 + wave + 4 r

    const unsigned tbytes = (unsigned)((size_t)p.B * p.C * T * 2);
    const __amdgpu_buffer_rsrc_t rx = dw_make_rsrc(p.x, tbytes);
    const __amdgpu_buffer_rsrc_t ry = dw_make_rsrc(p.y, tbytes);
    const int vo_in = 8 * lane < T ? 16 * lane : 0x7ffffff0;
    auto row_bytes = [&](int r) -> unsigned { return (unsigned)((b0 + wave + 4 * r) * p.C + c) * (unsigned)T * 2u; };
    auto issue = [&](int r) -> dwm_u32x4 {
        const bool ok = r < nrows;                           // wave-uniform
        return __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? vo_in : 0x7ffffff0, ok ? (int)row_bytes(r) : 0, 0);
    };
    // the taps FIRST: vmcnt retires in order, so a tap load issued behind the row requests would wait for all of them
    static_assert(WLEN <= 256, "one tap slot per thread");
    const int tj = (int)threadIdx.x - WPAD;
    const float tapv = p.w[(size_t)c * K + min(max(tj, 0), K - 1)];      // unconditional (clamped) load, selected below
    dwm_u32x4 raw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) raw[d] = issue(d);

    // ---- prologue, under the latency of those loads: taps, Toeplitz fragments, coefficients, the image's zero padding ----
    if (threadIdx.x < WLEN) lds_w[threadIdx.x] = (tj >= 0 && tj < K) ? tapv : 0.f;
    const int off = (-p.pad) & 7;
    const int in0a = (-p.pad) & ~7;                          // input position of image element 0
    const int lpad = -in0a;
    unsigned short* img = lds_img + wave * IMGP;
    for (int c8 = lane; c8 < IMGP / 8; c8 += 64)
        if (c8 < lpad / 8 || c8 >= (lpad + T) / 8) *reinterpret_cast<dwm_u32x4*>(img + 8 * c8) = dwm_u32x4{0u, 0u, 0u, 0u};
    const float ca = p.in_a[c], cb = p.in_b[c];
    __syncthreads();
    dwm_bf16x8 afr[STEPS][NT];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        unsigned pk[NT][4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            unsigned d0[3], d1[3];
            dwm_split(lds_w[WPAD + 32 * s + 8 * q_ + 2 * jp - n_ - off], d0);
            dwm_split(lds_w[WPAD + 32 * s + 8 * q_ + 2 * jp + 1 - n_ - off], d1);
#pragma unroll
            for (int t = 0; t < NT; ++t) pk[t][jp] = dwm_pack(d0[t], d1[t]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const dwm_u32x4 v = {pk[t][0], pk[t][1], pk[t][2], pk[t][3]};
            afr[s][t] = __builtin_bit_cast(dwm_bf16x8, v);
        }
    }
    const unsigned short* bsrc = img + 16 * n_ + 8 * q_;
    unsigned short* stg = img + lpad + 8 * lane;
    const bool stg_ok = 8 * lane < T;
    float s0 = 0.f, s1 = 0.f;

    // one row: stage raw[d] (BN1 + ReLU6, one bf16 digit), request row r + D into the freed registers, Toeplitz MFMAs, store
    auto row = [&](dwm_u32x4& rw, int r) {
        const bool ok = r < nrows;                           // wave-uniform; rows past the end compute zeros and store nothing
        const float ra = ok ? ca : 0.f, rb = ok ? cb : 0.f;
        float vals[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vals[e] = relu6f(fmaf(dwm_elem8(rw, e), ra, rb));
        const dwm_u32x4 w4 = {dwm_pack_rne(vals[0], vals[1]), dwm_pack_rne(vals[2], vals[3]), dwm_pack_rne(vals[4], vals[5]),
                              dwm_pack_rne(vals[6], vals[7])};
        if (stg_ok) *reinterpret_cast<dwm_u32x4*>(stg) = w4;              // the padding stays zero
        asm volatile("" ::: "memory");                                    // wave-local hand-off through LDS: program order
        rw = issue(r + D);
        const unsigned yb = row_bytes(r);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int t0 = 256 * sub + 16 * n_ + 4 * q_;
            dwm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                dwm_bf16x8 bfr[1];
                bfr[0] = *reinterpret_cast<const dwm_bf16x8*>(bsrc + 256 * sub + 32 * s);
                acc = dwm_mfma_digits<NT, 1>(afr[s], bfr, acc);
            }
            // positions >= T hold exact zeros only when the whole tail of the image is zero padding: mask them out of the sums
            const bool in_row = t0 < T;                                    // T % 8 == 0 and t0 % 4 == 0: all four or none
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yv = in_row ? acc[e] : 0.f;
                s0 += yv;
                s1 = fmaf(yv, yv, s1);
            }
            const dwm_u32x2 o2 = {dwm_pack_rne(acc[0], acc[1]), dwm_pack_rne(acc[2], acc[3])};
            __builtin_amdgcn_raw_buffer_store_b64(o2, ry, (ok && in_row) ? 2 * t0 : 0x7ffffff0, ok ? (int)yb : 0, 0);
        }
        asm volatile("" ::: "memory");                                    // the next row's LDS store stays behind these fragment reads
    };

#pragma unroll
    for (int d = 0; d < D; ++d) row(raw[d], d);                           // peeled first group
    for (int r0 = D; r0 < nrows; r0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) row(raw[d], r0 + d);
    }

    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) { lds_red[wave][0] = s0; lds_red[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t0s = (lds_red[0][0] + lds_red[1][0]) + (lds_red[2][0] + lds_red[3][0]);
        const float t1s = (lds_red[0][1] + lds_red[1][1]) + (lds_red[2][1] + lds_red[3][1]);
        p.stats[((size_t)g * p.C + c) * 2 + 0] = t0s;
        p.stats[((size_t)g * p.C + c) * 2 + 1] = t1s;
        if (p.fin.mode != 0) dw_finalize(p.fin, c, t0s, t1s);
    }
}
