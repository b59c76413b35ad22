// K2 for the benchmark's case, built around how the rows reach HBM: training-mode depthwise forward and fused backward with
// bf16-stored activations (asr.py:49 and its autograd) on rows that fit ONE wave item -- up to 512 outputs (NS = 2 sub-tiles of 256:
// the nominal T' = 512 and every shorter time-stretched length) or up to 768 (NS = 3: the longer time-stretched lengths, <= 763).
//
// Why a second set of kernels.  dwconv_mfma_kernel issues a row's first load only after the tap prologue (taps -> LDS -> barrier ->
// Toeplitz fragments), re-derives the image's zero padding from out-of-range loads on every row, loads a1 (backward: mask / xin) at
// the top of the row that needs it -- a full miss on the critical path of every row -- and walks rows longer than 512 outputs as two
// tiles.  Inside a training step its input and output stream from / to HBM (tools/bench_dw_regimes.py: the stand-alone number that
// round 2 quoted has both tensors resident in the 256 MB Infinity Cache), and what decides the rate there is measured in
// profiles/r03_stream_pattern_probe.txt and profiles/r03_dw_stream_ab.txt:
//   * a pure copy with this kernel's access pattern ([B][C][T] rows of 1 KB, one workgroup per channel) tops out at 4.5 - 5.1 TB/s --
//     and goes DOWN with more rows in flight per wave (D = 4: 4.46, D = 1: 5.10 TB/s): a wave that requests rows b, b + 4, b + 8, ...
//     at once spreads the chip's accesses over many 2 MB-apart regions, while all workgroups walking b in step sweep a few regions
//     sequentially.  Little's law was the wrong model: DRAM locality is the limit.  So D = 1: the row after the current one;
//   * the loads need nothing but the row index, so a wave requests its first rows at kernel entry and the whole prologue runs under
//     their latency; all of a row's streams (backward: dz2, a2 AND a1) are requested together, one row ahead;
//   * rows are read once and written once: nontemporal (CP = 2) is worth +9 % on a working set that rotates through HBM;
//   * one (row) item per wave: no second tile for 513 .. 768 outputs, the zero padding of the LDS image is written once.
// Measured, 8 layers, rotating working set: forward 189 -> 159 us, fused backward 391 -> 324 us; in the step forward 0.49 -> 0.54
// of 8 TB/s on the same box.
//
// The main loop is branch-free (a conditional load or store makes hipcc's s_waitcnt insertion take the worst-case count at the
// join, i.e. it waits for the prefetch just issued): rows past the wave's last row load through an out-of-range buffer offset
// (the hardware returns zeros and moves no bytes), run with zeroed coefficients (so they add nothing to the sums) and store
// through an out-of-range offset (dropped).  The first group of D rows is peeled so that the loop is entered in the queue state
// its back edge produces.
//
// Arithmetic, LDS image, fragment reads, statistics and the in-kernel BatchNorm finalisation (G == 1) are those of
// dwconv_mfma_kernel<K, ..., NT, WG, IO>: the NS = 2 kernels are bit-identical to it; NS = 3 sums a row's outputs in one pass
// instead of two tiles (same values, a different order of the fp32 partial sums).
#pragma once

// Round-6 knobs of the streaming kernels (A/B builds: tools/ab_variants.sh; the defaults are what ships)
#ifndef DWS_XSWZ
#define DWS_XSWZ 1            /* xin image of the fused backward stored with an XOR swizzle of its 8-byte chunks (see dws_xoff) */
#endif
#ifndef DWS_TAIL
#define DWS_TAIL 1            /* DA1: finish BatchNorm 1's backward and store the kept rows BEFORE the weight-gradient tiles go through LDS */
#endif
#ifndef DWS_XCD
#define DWS_XCD 2             /* blockIdx -> channel: 0 identity, 1 C/8 consecutive channels per XCD, 2 groups of DWS_XCD_GROUP channels per XCD */
#endif
#ifndef DWS_XCD_GROUP
#define DWS_XCD_GROUP 16      /* a power of two */
#endif

// blockIdx.x -> channel.  Workgroups are dealt to the 8 XCDs round-robin; with DWS_XCD the channels c, c + 1, ... (adjacent 1 KB rows
// of every utterance) go to the same XCD: rows whose pitch is not a whole number of 128-byte lines then share their boundary lines in
// ONE L2 instead of fetching them twice.
__device__ __forceinline__ int dws_chan(int bid, int C) {
#if DWS_XCD == 1
    return (C & 7) == 0 ? (bid & 7) * (C >> 3) + (bid >> 3) : bid;
#elif DWS_XCD == 2
    // groups of 16 consecutive channels per XCD, the groups dealt round-robin: the 16 channels whose partial sums share one 128-byte
    // line of a producer's slab [parts][C][2] (and whose BatchNorm parameters share a line) then run on ONE XCD, back to back, so the
    // consumer-side finalisation (dw_pre_issue) fetches each slab line into one L2 once instead of into all eight (measured: +9 / +18 MB
    // of reads per 1024- / 2048-channel launch, 1.14 x the algorithmic bytes)
    // (round 6, 8 layers on a rotating working set: forward 177 -> 170 us, plain fused backward 326 -> 301, kept-rows form 366 -> 340 --
    // the remap pays without any finalisation in the kernel too: an XCD's 32 CUs sweep 16 KB runs of each utterance instead of every
    // eighth 1 KB row; HBM bytes of the forward launches 1.132 -> 1.018 x algorithmic with the finalisation on)
    constexpr int GR = DWS_XCD_GROUP;
    if (C & (8 * GR - 1)) return bid;
    const int xcd = bid & 7, idx = bid >> 3;
    return ((idx / GR) * 8 + xcd) * GR + (idx & (GR - 1));
#else
    (void)C;
    return bid;
#endif
}

// Element offset, inside a 256-position half of the xin image, of the 4-sample chunk q of tile row n (row-major [16 x 16] bf16 tiles,
// 32-byte rows).  Plain: 16 n + 4 q.  A ds_write_b64 is served in four groups of 16 consecutive lanes on 32 banks: with q fixed in a
// group the plain addresses are 8 dwords apart -- 4 lanes per bank, a 4-way conflict on both stores of every row.  XOR-ing the chunk
// index with bits 2..3 of the row spreads each group over all 32 banks; the transposing read supplies one address per (row, chunk), so
// it simply asks for the swizzled chunk (dws_tr_fragment_x) and still covers 256 contiguous bytes per 32 lanes: conflict-free both ways.
__device__ __forceinline__ int dws_xoff(int n, int q) {
#if DWS_XSWZ
    return 16 * n + 4 * (q ^ ((n >> 2) & 3));
#else
    return 16 * n + 4 * q;
#endif
}
__device__ __forceinline__ dwm_bf16x8 dws_tr_fragment_x(const unsigned short* tile, int lane) {
#if DWS_XSWZ
    typedef __attribute__((address_space(3))) dwm_s16x4 lds_s16x4;
    const int o = dws_xoff(lane >> 2, lane & 3);
    const dwm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o));
    const dwm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + 256 + o));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#else
    return dwm_tr_fragment(tile, lane);
#endif
}

template <int K, int NS>
struct DwStreamGeom {
    using G_ = DwMfmaGeom<K, 7>;
    static constexpr int STEPS = G_::STEPS, WPAD = G_::WPAD, WLEN = G_::WLEN, IB = G_::IB;
    static constexpr int TMAX = 256 * NS;                                   // outputs (= inputs) per row
    static constexpr int NL = (TMAX + 511) / 512;                           // 16-byte loads per lane and stream
    // image elements: B-fragment reads touch [0, 256 (NS-1) + 240 + 32 STEPS), the weight-gradient fragments [0, 512 NL + 16 (IB-1))
    static constexpr int FWD_IMG = ((256 * (NS - 1) + 240 + 32 * STEPS + 7) / 8) * 8;
    static constexpr int BWD_IMG = ((512 * NL + 16 * IB + 7) / 8) * 8;
    static constexpr int XIMG = 512 * NL;                                   // xin image (weight gradient): whole 512-position steps
};

// CP: cache-policy bits of the row loads / stores (0 default, 2 = nontemporal: the rows are read once and written once)
// EV (eval mode, inference): the input is the already-activated h1 (no transform on load), the output relu6(acc * out_a + out_b)
// with the folded BatchNorm-2 coefficients, no statistics -- ConvBNActivate's "dw" stage with frozen statistics (asr.py:27-37, 49)
// F16 (EV only): the stored tensors hold IEEE fp16 (inference at precision "fp16"): the loaded words ARE the matrix operand -- no
// conversion while staging --, the taps are split into fp16 digits, the output is rounded to fp16
#ifndef DWS_FWD_MINW
#define DWS_FWD_MINW 4        /* A/B: workgroups per CU the register allocation must allow (8: all 2048 channels of a wide layer resident at once) */
#endif
#ifndef DWS_FWD_PADLDS
#define DWS_FWD_PADLDS 0      /* A/B: bytes of unused LDS per workgroup (caps the workgroups per CU: 26000 -> 4, 18000 -> 5) */
#endif
template <int K, int NT, int D, int CP = 0, int NS = 2, bool EV = false, bool F16 = false>
__global__ __launch_bounds__(256, (NS == 2 && !EV) ? DWS_FWD_MINW : 4) void dwconv_fwd16_stream_kernel(DwParams p) {
    static_assert(!F16 || EV, "fp16 storage: inference only");
    using S_ = DwStreamGeom<K, NS>;
    constexpr int STEPS = S_::STEPS, WPAD = S_::WPAD, WLEN = S_::WLEN, NL = S_::NL;
    constexpr int IMGP = S_::FWD_IMG > 512 * NL + 64 ? S_::FWD_IMG : 512 * NL + 64;   // + the staged runs of lanes past the row
    __shared__ __attribute__((aligned(16))) unsigned short lds_img[4 * IMGP + ((NS == 2 && !EV) ? DWS_FWD_PADLDS / 2 : 0)];
    __shared__ float lds_w[256];                             // WLEN used; every thread stores one slot (no lane-masked branch)
    __shared__ float lds_red[4][2];

    const int c = dws_chan(blockIdx.x, p.C), g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_ = lane & 15, q_ = lane >> 4;
    const int T = p.Tin;                                     // == Tout <= 256 NS; rows are stored with pitch P (a multiple of 8 samples)
    const int P = dw_pitch16(T, p.B);
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    // segment packing (EV only): SEGN utterances side by side in one item, SS image positions apart; SEGN = 1: one row per item
    const int SEGN = (EV && p.segn > 1) ? p.segn : 1, SS = (EV && p.segn > 1) ? p.segs : 0;
    const int nitems = (nb + SEGN - 1) / SEGN;
    const int nrows = wave < nitems ? (nitems - wave + 3) >> 2 : 0;  // this wave's items: utterances b0 + (wave + 4 r) SEGN ...

    const unsigned tbytes = (unsigned)((size_t)p.B * p.C * P * 2);
    const __amdgpu_buffer_rsrc_t rx = dw_make_rsrc(p.x, tbytes);
    const __amdgpu_buffer_rsrc_t ry = dw_make_rsrc(p.y, tbytes);
    const int seg_bytes = (p.cm ? P : p.C * P) * 2;          // byte distance between the rows of consecutive utterances of this channel
    int vo_in[NL], seg_in[NL], w_in[NL];
#pragma unroll
    for (int v = 0; v < NL; ++v) {
        const int rp = 8 * (lane + 64 * v);                  // image position of this lane's run
        seg_in[v] = SEGN > 1 ? rp / SS : 0;
        w_in[v] = rp - seg_in[v] * SS;                       // position inside the segment's row
        vo_in[v] = (seg_in[v] < SEGN && w_in[v] < T) ? seg_in[v] * seg_bytes + 2 * w_in[v] : 0x7ffffff0;
    }
    auto row_bytes = [&](int r) -> unsigned { return dw_row_index(p, b0 + (wave + 4 * r) * SEGN, c) * (unsigned)P * 2u; };
    struct Row { dwm_u32x4 x[NL]; };
    // (the scalar row offset is formed unconditionally -- only the per-lane offset takes part in the bounds check, so an
    // out-of-range voffset alone makes the access a no-op -- and the voffset by a select: no branch for hipcc to build around a load)
    auto issue = [&](int r, Row& rw) {
        const bool ok = r < nrows;                           // wave-uniform
        const int left = nb - (wave + 4 * r) * SEGN;         // utterances left from this item's first one (the last item may be short)
#pragma unroll
        for (int v = 0; v < NL; ++v)
            rw.x[v] = __builtin_amdgcn_raw_buffer_load_b128(rx, (ok && seg_in[v] < left) ? vo_in[v] : 0x7ffffff0, (int)row_bytes(r), CP);
    };
    // the taps FIRST: vmcnt retires in order, so a tap load issued behind the row requests would wait for all of them
    static_assert(WLEN <= 256, "one tap slot per thread");
    const int tj = (int)threadIdx.x - WPAD;
    const float tapv = p.w[(size_t)c * K + min(max(tj, 0), K - 1)];      // unconditional (clamped) load, selected below
    // BatchNorm 1 finalised HERE (DwPre, one group): its slab and parameter reads go out in front of the rows (dw_pre_issue)
    DwPreRegs prer;
    const bool pre_fast = !EV && p.pre.f.mode != 0 && p.pre.parts <= DW_PRE_MAXPARTS;
    if constexpr (!EV) dw_pre_issue(p.pre, p.C, c, lane, pre_fast && wave == 0, p.w, prer);
    __builtin_amdgcn_sched_barrier(0);
    Row raw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, raw[d]);
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue, under the latency of those loads: taps, Toeplitz fragments, coefficients, the image's zero padding ----
    // (every thread stores: inside an `if (threadIdx.x < WLEN)` hipcc sinks the tap load into the branch, behind the row requests)
    lds_w[threadIdx.x] = (tj >= 0 && tj < K) ? tapv : 0.f;
    const int off = (-p.pad) & 7;
    const int in0a = (-p.pad) & ~7;                          // input position of image element 0
    const int lpad = -in0a;
    unsigned short* img = lds_img + wave * IMGP;
    for (int c8 = lane; c8 < IMGP / 8; c8 += 64)
        if (SEGN > 1 || c8 < lpad / 8 || c8 >= (lpad + P) / 8) *reinterpret_cast<dwm_u32x4*>(img + 8 * c8) = dwm_u32x4{0u, 0u, 0u, 0u};
    // BatchNorm 1 of this channel: coefficients from the finaliser launch -- or, with one group (this workgroup is then the only
    // consumer of channel c), finalised HERE from the expand GEMM's slab of partial sums by the first wave (DwPre), under the latency
    // of the row requests above: one dependent launch less per block
    __shared__ float lds_coef[3];
    __syncthreads();                                         // taps and zero padding in place
    // (the coefficient reads are unconditional -- a harmless address when the in-kernel finalisation supplies them -- and go out before
    //  the fragment build, which hides them)
    const bool pre_on = !EV && p.pre.f.mode != 0;
    float ca = 1.f, cb = 0.f, oa = 1.f, ob = 0.f;
    if constexpr (EV) { oa = p.out_a[c]; ob = p.out_b[c]; }
    else { ca = *(pre_on ? p.w : p.in_a + c); cb = *(pre_on ? p.w : p.in_b + c); }
    dwm_bf16x8 afr[STEPS][NT];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        unsigned pk[NT][4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            unsigned d0[3], d1[3];
            if constexpr (F16) {
                dwm_split_f16(lds_w[WPAD + 32 * s + 8 * q_ + 2 * jp - n_ - off], d0);
                dwm_split_f16(lds_w[WPAD + 32 * s + 8 * q_ + 2 * jp + 1 - n_ - off], d1);
#pragma unroll
                for (int t = 0; t < NT; ++t) pk[t][jp] = d0[t] | (d1[t] << 16);
            } else {
                dwm_split_taps<NT>(lds_w[WPAD + 32 * s + 8 * q_ + 2 * jp - n_ - off], d0);
                dwm_split_taps<NT>(lds_w[WPAD + 32 * s + 8 * q_ + 2 * jp + 1 - n_ - off], d1);
#pragma unroll
                for (int t = 0; t < NT; ++t) pk[t][jp] = dwm_pack(d0[t], d1[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const dwm_u32x4 v = {pk[t][0], pk[t][1], pk[t][2], pk[t][3]};
            afr[s][t] = __builtin_bit_cast(dwm_bf16x8, v);
        }
    }
    // the in-kernel finalisation AFTER the fragment build: its slab reads (issued at entry) have had the whole prologue to arrive
    if constexpr (!EV) {
        if (pre_on) {
            if (wave == 0) {
                if (pre_fast) dw_pre_finish(p.pre, c, lane, prer, lds_coef);
                else dw_finalize_parts(p.pre, p.C, c, lane, lds_coef);
            }
            __syncthreads();
            ca = lds_coef[0]; cb = lds_coef[1];
        }
    }
    const unsigned short* bsrc = img + 16 * n_ + 8 * q_;
    // samples this lane stages per load (0 .. 8): positions 8 (lane + 64 v) + e < T.  Zero padding applies to the TRANSFORMED tensor,
    // and the pitch padding of a stored row holds arbitrary bits: everything at or past T is forced to zero.  A lane whose run lies
    // wholly past the row writes its zeros INTO the zero padding (an unconditional store: a lane-masked store is a branch, and the
    // wait for the row's data inside it spoils the counted waits after the join)
    int nval[NL];
    unsigned short* stg[NL];
#pragma unroll
    for (int v = 0; v < NL; ++v) {
        nval[v] = seg_in[v] < SEGN ? min(max(T - w_in[v], 0), 8) : 0;      // (runs between two segments: zeros, the rows' padding)
        stg[v] = img + lpad + 8 * (lane + 64 * v);
    }
    float s0 = 0.f, s1 = 0.f;

    // one row: stage raw[d] (BN1 + ReLU6, one bf16 digit), request row r + D into the freed registers, Toeplitz MFMAs, store
    auto row = [&](Row& rw, int r) {
        const bool ok = r < nrows;                           // wave-uniform; rows past the end compute zeros and store nothing
        const float ra = ok ? ca : 0.f, rb = ok ? cb : 0.f;
        (void)ra; (void)rb;
#pragma unroll
        for (int v = 0; v < NL; ++v) {
            if constexpr (F16) {
                // the stored fp16 words are the operand: only the positions at or past T are cleared (by halfword)
                dwm_u32x4 w4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned m = (2 * j < nval[v] ? 0x0000ffffu : 0u) | (2 * j + 1 < nval[v] ? 0xffff0000u : 0u);
                    w4[j] = rw.x[v][j] & m;
                }
                *reinterpret_cast<dwm_u32x4*>(stg[v]) = w4;
                continue;
            }
            float vals[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xin = EV ? dwm_elem8(rw.x[v], e) : relu6f(fmaf(dwm_elem8(rw.x[v], e), ra, rb));
                vals[e] = e < nval[v] ? xin : 0.f;
            }
            const dwm_u32x4 w4 = {dwm_pack_rne(vals[0], vals[1]), dwm_pack_rne(vals[2], vals[3]), dwm_pack_rne(vals[4], vals[5]),
                                  dwm_pack_rne(vals[6], vals[7])};
            *reinterpret_cast<dwm_u32x4*>(stg[v]) = w4;
        }
        asm volatile("" ::: "memory");                                    // wave-local hand-off through LDS: program order
        issue(r + D, rw);
        const unsigned yb = row_bytes(r);
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
            const int o0 = 256 * sub + 16 * n_ + 4 * q_;     // output position in the image; its segment and position in that row:
            const int oseg = SEGN > 1 ? o0 / SS : 0;
            const int t0 = o0 - oseg * SS;
            dwm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                dwm_bf16x8 bfr[1];
                bfr[0] = *reinterpret_cast<const dwm_bf16x8*>(bsrc + 256 * sub + 32 * s);
                acc = dwm_mfma_digits<NT, 1, F16>(afr[s], bfr, acc);
            }
            // outputs at positions >= T are not part of the row (they are not zero: the taps still reach real samples): out of the sums
            const bool in_row = t0 < T && oseg < SEGN && oseg < nb - (wave + 4 * r) * SEGN;
            if constexpr (EV) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = relu6f(fmaf(acc[e], oa, ob));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float yv = t0 + e < T ? acc[e] : 0.f;
                    s0 += yv;
                    s1 = fmaf(yv, yv, s1);
                }
            }
            // 4 bf16 = one 8-byte store (the pitch keeps it aligned; samples past T land in the row's padding)
            const dwm_u32x2 o2 = {F16 ? pack16<true>(acc[0], acc[1]) : dwm_pack_rne(acc[0], acc[1]),
                                  F16 ? pack16<true>(acc[2], acc[3]) : dwm_pack_rne(acc[2], acc[3])};
            __builtin_amdgcn_raw_buffer_store_b64(o2, ry, (ok && in_row) ? oseg * seg_bytes + 2 * t0 : 0x7ffffff0, (int)yb, CP);
        }
        asm volatile("" ::: "memory");                                    // the next row's LDS store stays behind these fragment reads
        __builtin_amdgcn_sched_barrier(0);                                // rows are not interleaved (register pressure; LDS image reuse)
    };

#pragma unroll
    for (int d = 0; d < D; ++d) row(raw[d], d);                           // peeled first group
    for (int r0 = D; r0 < nrows; r0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) row(raw[d], r0 + d);
    }

    if constexpr (EV) return;
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

// ---------------------------------------------------------------------------------------------------------------------
// The fused backward of the same stage (autograd of asr.py:49 with BatchNorm-2 backward applied on load, the ReLU6 mask /
// BatchNorm-1 backward sums on the way out and the depthwise weight gradient riding on the pass; every tensor bf16-stored), for
// rows that fit one tile -- the streaming form of dwconv_mfma_kernel<K, AFFINE2, MASK_STATS, NT, true, X | X2 | AUX | Y>.
// Per row a wave reads THREE 1 KB streams (dz2, a2 for the affine; a1 at its output positions for the mask and xin) and writes one:
// the general kernel keeps one row of the first two in flight and loads a1 at the top of the row it is needed in (a full miss on
// the critical path of every row).  Here all three are requested D rows ahead, from kernel entry on (12 VGPRs per row in flight).
// Same arithmetic and order of accumulation: bit-identical results (E tiles, sums, stored gradient).
// DA1 (round 5): the kernel finalises BatchNorm 1's backward itself (one group: it owns the channel's sums after its last row), so it
// can hand its consumers -- the expand weight gradient and the expand backward-data GEMM -- the FINISHED gradient
// da1 = p dz1 + q a1 + r instead of dz1, which both of them re-derive today from two 67 MB tensors with a transform on load (the
// backward-data GEMM runs 62 us so against 43 on a plain operand).  A wave keeps the up to MAXR = 8 rows it produced -- dz1 as
// stored (bf16) and the a1 it loaded for the mask: 4 NS registers a row -- instead of storing them, and after the channel's
// coefficients are known writes fmaf(dz1, p, fmaf(a1, q, r)) rounded to bf16: exactly the value (same operations, same order) the
// consumers' on-load transform produced, so every downstream result is bit-identical.  The row loop is unrolled MAXR times (rows past
// a wave's last one run as the branch-free kernel's dummy rows); B <= 4 MAXR.
#ifndef DWS_BWD_MINW
#define DWS_BWD_MINW 3        /* A/B: 2 = the plain fused backward at the kept-rows form's occupancy */
#endif
#ifndef DWS_BWD_PADLDS
#define DWS_BWD_PADLDS 0
#endif
#ifndef DWS_DA1_K3
#define DWS_DA1_K3 0          /* kept-rows form: kernel sizes up to this keep three workgroups per CU (168 registers) */
#endif
// DWS_DA1_KEEP 0 (A/B): the rows are NOT kept -- the plain form (rolled row loop, dz1 stored as produced, three workgroups per CU), then each
// wave reads its own rows back (dz1 it wrote, a1 it read: L2 / Infinity Cache) and overwrites dz1 with the finished gradient.  Bit-identical
// (the stored bf16 dz1 IS what the kept register held) but measured SLOWER in the step: this kernel 0.498 ms against 0.448 kept-rows and
// 0.374 without DA1 (step 3.36 / 3.33 / 3.36 ms): the second pass costs more than the occupancy returns.
#ifndef DWS_DA1_KEEP
#define DWS_DA1_KEEP 1
#endif
template <int K, int NT, int D, int CP = 0, int NS = 2, bool DA1 = false, int MAXR = 8>
__global__ __launch_bounds__(256, (DA1 && DWS_DA1_KEEP && NS == 2 && K > DWS_DA1_K3) ? 2 : DWS_BWD_MINW) void dwconv_bwd16_stream_kernel(DwParams p) {
    // rows of up to 512 outputs: the wave KEEPS its rows in registers; rows of 513 .. 768 (NS = 3: 12 registers a row, the file is full)
    // take the read-back form -- dz1 stored as produced, then each wave reads its own rows of dz1 and a1 back (L2 / Infinity Cache) and
    // overwrites dz1 with the finished gradient.  Measured (round 6, stretch-110 step): this kernel 0.416 -> 0.610 ms, the two expand
    // GEMMs -0.19 ms: no net gain -- off by default (V100_IR_DA1_TMAX, depthwise_bwd_fused16g.hip)
    constexpr bool KEEP = DA1 && DWS_DA1_KEEP && NS == 2;
    using S_ = DwStreamGeom<K, NS>;
    constexpr int STEPS = S_::STEPS, WPAD = S_::WPAD, WLEN = S_::WLEN, IB = S_::IB, NL = S_::NL, XIMG = S_::XIMG;
    constexpr int IMG0 = S_::FWD_IMG > S_::BWD_IMG ? S_::FWD_IMG : S_::BWD_IMG;
    constexpr int IMGP = IMG0 > 512 * NL + 64 ? IMG0 : 512 * NL + 64;
    constexpr int WAVE_U16 = IMGP + XIMG;
    constexpr int E_FLOATS = 16 * IB * DWM_EP;
    constexpr int LDS_BYTES = (4 * WAVE_U16 * 2 > 4 * E_FLOATS * 4) ? 4 * WAVE_U16 * 2 : 4 * E_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES + DWS_BWD_PADLDS];      // (DWS_BWD_PADLDS: A/B of the occupancy alone)
    __shared__ float lds_w[256];
    __shared__ float lds_red[4][2];
    __shared__ float lds_fin[4];

    const int c = dws_chan(blockIdx.x, p.C), g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_ = lane & 15, q_ = lane >> 4;
    const int T = p.Tin;
    const int P = dw_pitch16(T, p.B);
    const int bper = (p.B + p.G - 1) / p.G;
    const int b0 = g * bper;
    const int nb = min(p.B, b0 + bper) - b0;
    const int nrows = wave < nb ? (nb - wave + 3) >> 2 : 0;

    const unsigned tbytes = (unsigned)((size_t)p.B * p.C * P * 2);
    const __amdgpu_buffer_rsrc_t rx = dw_make_rsrc(p.x, tbytes), rx2 = dw_make_rsrc(p.x2, tbytes), raux = dw_make_rsrc(p.aux, tbytes);
    const __amdgpu_buffer_rsrc_t ry = dw_make_rsrc(p.y, tbytes);
    int vo_in[NL], vo_aux[NS];
#pragma unroll
    for (int v = 0; v < NL; ++v) vo_in[v] = 8 * (lane + 64 * v) < T ? 16 * (lane + 64 * v) : 0x7ffffff0;
#pragma unroll
    for (int sub = 0; sub < NS; ++sub) {
        const int t0 = 256 * sub + 16 * n_ + 4 * q_;
        vo_aux[sub] = t0 < T ? 2 * t0 : 0x7ffffff0;
    }
    auto row_bytes = [&](int r) -> unsigned { return dw_row_index(p, b0 + wave + 4 * r, c) * (unsigned)P * 2u; };
    struct Row { dwm_u32x4 g[NL], g2[NL]; dwm_u32x2 a[NS]; };
    auto issue = [&](int r, Row& rw) {
        const bool ok = r < nrows;
        const int so = (int)row_bytes(r);
#pragma unroll
        for (int v = 0; v < NL; ++v) {
            rw.g[v] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? vo_in[v] : 0x7ffffff0, so, CP);
            rw.g2[v] = __builtin_amdgcn_raw_buffer_load_b128(rx2, ok ? vo_in[v] : 0x7ffffff0, so, CP);
        }
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) rw.a[sub] = __builtin_amdgcn_raw_buffer_load_b64(raux, ok ? vo_aux[sub] : 0x7ffffff0, so, CP);
    };
    static_assert(WLEN <= 256, "one tap slot per thread");
    const int tj = (int)threadIdx.x - WPAD;
    const int tjc = min(max(tj, 0), K - 1);
    const float tapv = p.w[(size_t)c * K + (p.flip ? (K - 1 - tjc) : tjc)];       // taps first (vmcnt retires in order)
    // DA1: the three per-channel inputs of BatchNorm 1's backward finalisation (saved mean, rstd, gamma) are requested HERE, with the
    // taps, and parked in LDS: fetched by thread 0 at the end of the kernel they were three dependent misses between the channel's last
    // row and the stores of all its kept rows
    float fmu = 0.f, frs = 0.f, fga = 0.f;
    if constexpr (DA1 && DWS_TAIL) { fmu = p.fin.a[c]; frs = p.fin.b[c]; fga = p.fin.gamma[c]; }
    // BatchNorm-2 backward finalised HERE (DwPre, one group): its slab and parameter reads go out in front of the rows
    DwPreRegs prer;
    const bool pre_fast = p.pre.f.mode != 0 && p.pre.parts <= DW_PRE_MAXPARTS;
    dw_pre_issue(p.pre, p.C, c, lane, pre_fast && wave == 0, p.w, prer);
    __builtin_amdgcn_sched_barrier(0);
    Row raw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, raw[d]);
    __builtin_amdgcn_sched_barrier(0);

    lds_w[threadIdx.x] = (tj >= 0 && tj < K) ? tapv : 0.f;
    if constexpr (DA1 && DWS_TAIL) { lds_fin[0] = fmu; lds_fin[1] = frs; lds_fin[2] = fga; }      // (every thread, the same values: no branch)
    const int off = (-p.pad) & 7;
    const int in0a = (-p.pad) & ~7;
    const int lpad = -in0a;
    unsigned short* img = reinterpret_cast<unsigned short*>(lds_raw) + wave * WAVE_U16;
    unsigned short* ximg = img + IMGP;
    for (int c8 = lane; c8 < IMGP / 8; c8 += 64)
        if (c8 < lpad / 8 || c8 >= (lpad + P) / 8) *reinterpret_cast<dwm_u32x4*>(img + 8 * c8) = dwm_u32x4{0u, 0u, 0u, 0u};
    // the xin image past the last sub-tile (NS = 3: positions 768 .. 1023 of the second 512-position contraction step) stays zero
    for (int c8 = 32 * NS + lane; c8 < XIMG / 8; c8 += 64) *reinterpret_cast<dwm_u32x4*>(ximg + 8 * c8) = dwm_u32x4{0u, 0u, 0u, 0u};
    // BatchNorm-2 backward coefficients (p, q, r): from the finaliser launch, or finalised here from the project backward-data GEMM's
    // slab by the first wave (DwPre; one group only), under the latency of the row requests
    __shared__ float lds_coef[3];
    __syncthreads();                                         // taps and zero padding in place
    const bool pre_on = p.pre.f.mode != 0;
    float ca = *(pre_on ? p.w : p.in_a + c), cb = *(pre_on ? p.w : p.in_b + c), cc = *(pre_on ? p.w : p.in_c + c);
    const float oa = p.out_a[c], ob = p.out_b[c];
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
    // the in-kernel finalisation AFTER the fragment build (see the forward kernel)
    if (pre_on) {
        if (wave == 0) {
            if (pre_fast) dw_pre_finish(p.pre, c, lane, prer, lds_coef);
            else dw_finalize_parts(p.pre, p.C, c, lane, lds_coef);
        }
        __syncthreads();
        ca = lds_coef[0]; cb = lds_coef[1]; cc = lds_coef[2];
    }
    const unsigned short* bsrc = img + 16 * n_ + 8 * q_;
    int nval[NL];
    unsigned short* stg[NL];
#pragma unroll
    for (int v = 0; v < NL; ++v) {
        nval[v] = min(max(T - 8 * (lane + 64 * v), 0), 8);       // samples of the row this lane stages (everything at or past T is zero)
        stg[v] = img + lpad + 8 * (lane + 64 * v);
    }
    dwm_f32x4 eacc[IB];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) eacc[ib] = dwm_f32x4{0.f, 0.f, 0.f, 0.f};
    float s0 = 0.f, s1 = 0.f;

    dwm_u32x2 keep_o[KEEP ? MAXR : 1][NS], keep_a[KEEP ? MAXR : 1][NS];
    auto row = [&](Row& rw, int r, auto rconst) {
        constexpr int RI = decltype(rconst)::value;  // DA1: the row's slot in the kept arrays (a compile-time index: registers)
        // (DA1: r is a constant in each of the eight unrolled copies, so `ok` and the five coefficient selects below would be hoisted
        //  out of all of them and live across the whole loop: 40 registers.  An opaque copy of nrows keeps them inside their row.)
        int nrows_l = nrows;
        if constexpr (KEEP) asm volatile("" : "+s"(nrows_l));
        const bool ok = r < nrows_l;                 // rows past the end: zero coefficients -> g' = 0, xin = 0, mask 0, nothing stored
        const float ra = ok ? ca : 0.f, rb = ok ? cb : 0.f, rc = ok ? cc : 0.f, roa = ok ? oa : 0.f, rob = ok ? ob : 0.f;
        float auxv[NS][4];
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
            if constexpr (KEEP) keep_a[RI][sub] = rw.a[sub];      // (before issue() below re-uses rw for the next row)
#pragma unroll
            for (int e = 0; e < 4; ++e) auxv[sub][e] = dwm_elem(rw.a[sub], e);
        }
#pragma unroll
        for (int v = 0; v < NL; ++v) {
            float vals[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) vals[e] = e < nval[v] ? fmaf(dwm_elem8(rw.g[v], e), ra, fmaf(dwm_elem8(rw.g2[v], e), rb, rc)) : 0.f;
            const dwm_u32x4 w4 = {dwm_pack_rne(vals[0], vals[1]), dwm_pack_rne(vals[2], vals[3]), dwm_pack_rne(vals[4], vals[5]),
                                  dwm_pack_rne(vals[6], vals[7])};
            *reinterpret_cast<dwm_u32x4*>(stg[v]) = w4;
        }
        // xin = relu6(bn1(a1)) at this lane's NS x 4 output positions -> the tile image of the weight-gradient product
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
            const int t0 = 256 * sub + 16 * n_ + 4 * q_;
            float xv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = (t0 + e < T) ? relu6f(fmaf(auxv[sub][e], roa, rob)) : 0.f;
            const dwm_u32x2 w2 = {dwm_pack_rne(xv[0], xv[1]), dwm_pack_rne(xv[2], xv[3])};
            *reinterpret_cast<dwm_u32x2*>(ximg + 256 * sub + dws_xoff(n_, q_)) = w2;
        }
        asm volatile("" ::: "memory");
        issue(r + D, rw);
        // E[16 ib + m][rr] += sum_p g'img[16 (p + ib) + m] * ximg[16 p + rr], 32 blocks p (512 positions) per contraction step
#pragma unroll
        for (int h = 0; h < NL; ++h) {
            dwm_bf16x8 xfr[1];
            xfr[0] = dws_tr_fragment_x(ximg + 512 * h, lane);
#pragma unroll
            for (int ib = 0; ib < IB; ++ib) {
                dwm_bf16x8 gfr[1];
                gfr[0] = dwm_tr_fragment(img + 512 * h + 16 * ib, lane);
                eacc[ib] = dwm_mfma_digits<1, 1>(gfr, xfr, eacc[ib]);
            }
        }
        const unsigned yb = row_bytes(r);
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
            const int t0 = 256 * sub + 16 * n_ + 4 * q_;
            dwm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                dwm_bf16x8 bfr[1];
                bfr[0] = *reinterpret_cast<const dwm_bf16x8*>(bsrc + 256 * sub + 32 * s);
                acc = dwm_mfma_digits<NT, 1>(afr[s], bfr, acc);
            }
            const bool in_row = t0 < T;
            float outv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pre = fmaf(auxv[sub][e], roa, rob);
                float yv = (pre > 0.f && pre < 6.f) ? acc[e] : 0.f;
                if (t0 + e < T) { s0 += yv; s1 = fmaf(yv, auxv[sub][e], s1); }
                outv[e] = yv;
            }
            const dwm_u32x2 o2 = {dwm_pack_rne(outv[0], outv[1]), dwm_pack_rne(outv[2], outv[3])};
            if constexpr (KEEP) keep_o[RI][sub] = o2;
            else __builtin_amdgcn_raw_buffer_store_b64(o2, ry, (ok && in_row) ? 2 * t0 : 0x7ffffff0, (int)yb, CP);
        }
        if constexpr (KEEP) (void)yb;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);          // rows are not interleaved: the sums of a row retire before the next row starts
    };

    using R0 = std::integral_constant<int, 0>;
    if constexpr (KEEP) {
        static_assert(D >= 1 && D <= 4, "the kept-rows form keeps one to four rows of loads in flight");
#define DWS_ROW(i_) if constexpr ((i_) < MAXR) row(raw[(i_) % D], (i_), std::integral_constant<int, ((i_) < MAXR ? (i_) : 0)>{});
        DWS_ROW(0) DWS_ROW(1) DWS_ROW(2) DWS_ROW(3) DWS_ROW(4) DWS_ROW(5) DWS_ROW(6) DWS_ROW(7)
#undef DWS_ROW
        static_assert(MAXR <= 8, "unrolled by hand up to 8 rows per wave");
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) row(raw[d], d, R0{});
        for (int r0 = D; r0 < nrows; r0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) row(raw[d], r0 + d, R0{});
        }
    }

    // dWf[jf] = sum over waves and rr of E[rr + jf + off][rr]; the E tiles go through LDS (over the dead images)
    auto e_exchange = [&]() {
        float* ebuf = reinterpret_cast<float*>(lds_raw);
#pragma unroll
        for (int ib = 0; ib < IB; ++ib)
#pragma unroll
            for (int e = 0; e < 4; ++e) ebuf[(wave * 16 * IB + 16 * ib + 4 * q_ + e) * DWM_EP + n_] = eacc[ib][e];
        __syncthreads();
        for (int jf = threadIdx.x; jf < K; jf += 256) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float sw = 0.f;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) sw += ebuf[(w * 16 * IB + rr + jf + off) * DWM_EP + rr];
                sum += sw;
            }
            p.wpartial[((size_t)g * p.C + c) * K + (K - 1 - jf)] = sum;
        }
    };
    constexpr bool TAIL_FIRST = DA1 && DWS_TAIL;      // the kept rows leave first; the weight-gradient tiles follow under their stores
    __syncthreads();
    if constexpr (!TAIL_FIRST) {
        e_exchange();
        __syncthreads();
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
        if constexpr (TAIL_FIRST) dw_finalize_bwd_pre(p.fin, c, (double)t0s, (double)t1s, lds_fin[0], lds_fin[1], lds_fin[2], lds_coef);
        else if constexpr (DA1) dw_finalize_d(p.fin, c, (double)t0s, (double)t1s, lds_coef);      // (p, q, r) of BatchNorm 1's backward -> LDS too
        else if (p.fin.mode != 0) dw_finalize(p.fin, c, t0s, t1s);
    }
    if constexpr (DA1) {
        __syncthreads();
        const float pa = lds_coef[0], qb = lds_coef[1], rc = lds_coef[2];
        if constexpr (KEEP) {
#pragma unroll
            for (int ri = 0; ri < MAXR; ++ri) {
                const bool ok = ri < nrows;
                const unsigned yb = row_bytes(ok ? ri : 0);
#pragma unroll
                for (int sub = 0; sub < NS; ++sub) {
                    const int t0 = 256 * sub + 16 * n_ + 4 * q_;
                    float dv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] = fmaf(dwm_elem(keep_o[ri][sub], e), pa, fmaf(dwm_elem(keep_a[ri][sub], e), qb, rc));
                    const dwm_u32x2 o2 = {dwm_pack_rne(dv[0], dv[1]), dwm_pack_rne(dv[2], dv[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(o2, ry, (ok && t0 < T) ? 2 * t0 : 0x7ffffff0, (int)yb, CP);
                }
            }
        } else {
            // the wave's own rows again, four at a time: 4 NS loads of dz1 (as stored above by this wave: the stores are older in its
            // memory queue) and of a1 in flight, then the finished rows over the unfinished ones
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int r0 = 0; r0 < nrows; r0 += 4) {
                dwm_u32x2 go[4][NS], ga[4][NS];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = r0 + i < nrows;
                    const unsigned yb = row_bytes(ok ? r0 + i : 0);
#pragma unroll
                    for (int sub = 0; sub < NS; ++sub) {
                        const int t0 = 256 * sub + 16 * n_ + 4 * q_;
                        go[i][sub] = __builtin_amdgcn_raw_buffer_load_b64(ry, (ok && t0 < T) ? 2 * t0 : 0x7ffffff0, (int)yb, 0);
                        ga[i][sub] = __builtin_amdgcn_raw_buffer_load_b64(raux, ok ? vo_aux[sub] : 0x7ffffff0, (int)yb, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = r0 + i < nrows;
                    const unsigned yb = row_bytes(ok ? r0 + i : 0);
#pragma unroll
                    for (int sub = 0; sub < NS; ++sub) {
                        const int t0 = 256 * sub + 16 * n_ + 4 * q_;
                        float dv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) dv[e] = fmaf(dwm_elem(go[i][sub], e), pa, fmaf(dwm_elem(ga[i][sub], e), qb, rc));
                        const dwm_u32x2 o2 = {dwm_pack_rne(dv[0], dv[1]), dwm_pack_rne(dv[2], dv[3])};
                        __builtin_amdgcn_raw_buffer_store_b64(o2, ry, (ok && t0 < T) ? 2 * t0 : 0x7ffffff0, (int)yb, CP);
                    }
                }
            }
        }
    }
    if constexpr (TAIL_FIRST) e_exchange();           // (every wave passed two barriers since its last image read: the images are dead)
}
