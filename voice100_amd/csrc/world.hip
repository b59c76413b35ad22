// WORLD synthesis on the device: pyworld.decode_aperiodicity + pyworld.synthesize as the reference calls them
// (voice100/vocoder.py:100-101), the last step of BASELINE configs[2] ("... -> WORLD features + vocoder").
//
// PARITY PARTIALLY PINNED (the reference's docs samples witness the noise path, time-base phase, placement, length, level; no
// input/output vector exists): the arithmetic lives in pyworld 0.3.2 (C++ WORLD), which is neither in the reference tree nor in the build
// image.  These kernels follow the published algorithm as restated in oracle/world_synth.py (Morise et al. 2016; D4C 2016) and
// are held to that restatement (<= 1e-4 of the waveform's peak, pulse instants bit-exact).
//
// Three launches per call, every utterance of the batch at once:
//   1. world_timebase_kernel   one workgroup per utterance.  F0 contour -> per-sample phase increment (linear interpolation over the
//                              frame grid, in DOUBLE and with contraction off: it must round like the C / numpy reference), then the
//                              running phase as ONE sequential double accumulation (lane 0: the unvoiced default of 500 Hz at
//                              16 kHz puts every crossing exactly on a sample, so which side of 2 pi a sample lands on is decided by
//                              the rounding of that particular summation order -- a parallel scan would move pulses by one sample),
//                              wrap + crossing detection + ordered compaction in parallel -> pulse sample index, fractional shift.
//   2. world_pulse_kernel      one WAVE per pulse: envelope / aperiodicity interpolated between the two frames, minimum-phase
//                              spectra (log -> real cepstrum -> fold -> spectrum: two 512-point real FFTs each, run as 256-point
//                              radix-4 complex FFTs through 2 KB of wave-local LDS), fractional delay, inverse FFT, DC removal for
//                              the periodic part; a zero-mean noise burst (WORLD's own xorshift sequence, a fixed table) coloured by
//                              the aperiodic minimum-phase spectrum -> a 512-sample response per pulse.
//   3. world_overlap_add_kernel  one thread per output sample gathers the responses that cover it, in pulse order (deterministic; no
//                              float atomics).
// fp32 except the time base.  fft_size 512 (16 kHz) runs the fp32 wave-per-pulse kernel, every other power of two in 64 ... 2048 (22.05 kHz /
// 1024) the fp64 workgroup-per-pulse kernel.  V100_ERR_SHAPE only when a time-base chunk's contour window fits neither the chained
// kernel (TBC_CH / TBC_NFR) nor the serial one (TBS_CH / TBS_NFR): fewer than ~34 samples per frame (frame periods below ~2.1 ms at 16 kHz).
#include "common.h"
#include "../../include/voice100_hip.h"
#include "world_f64.h"
#include <stdlib.h>

namespace {
constexpr int NF = 512, NH = 256, NB = 257;          // fft size, half, bins of the fp32 wave-per-pulse kernel (16 kHz)
constexpr double kPi = 3.1415926535897932384626433832795;
constexpr double kDefaultF0 = 500.0;

struct WorldParams {
    const float* f0; const float* sp; const float* ap; const int* frames;
    const float* randn; long long table_len;
    float* y; int* n_pulses;
    unsigned char* vuv; int* idx; float* xshift; float* resp;
    int B, T, fs, Ymax, Pcap;
    double frame_period;      // seconds
    double frame_period_ms;
    const float* tw256; const float* tw512; const float* dcr;
    const float* coded; int nb;   // optional: coded band aperiodicity [B][T][nb] (dB) decoded on the fly instead of `ap`
    int nf;                       // fft size (512: the fp32 wave-per-pulse kernel; other powers of two: world_pulse_f64_kernel)
};

// everything up to the matching "contract(fast)" must round like the C / numpy reference: no fused multiply-adds
#pragma clang fp contract(off)
__device__ __forceinline__ int world_ylen(int frames, double frame_period_ms, int fs) {
    return (int)((double)frames * frame_period_ms * (double)fs / 1000.0);
}

// ---------------------------------------------------------------------------------------------------------------------------
// 1. time base
constexpr int TBS_CH = 2048, TBS_NFR = 64;             // the SERIAL kernel (one workgroup per utterance): samples per chunk; contour frames staged per chunk
__global__ __launch_bounds__(256) void world_timebase_kernel(WorldParams p) {
    // Chunks of CH samples through LDS: (1) all threads interpolate the contour -> phase increments, (2) thread 0 accumulates them IN
    // ORDER (the one summation order, see the header; ~10 cycles a sample: the increments come from LDS, unrolled so the reads run ahead of
    // the dependent adds), (3) all threads wrap, detect the 2 pi crossings and compact them in order.  The running phase never goes to
    // HBM; only the per-sample voicing flag (1 byte) and the pulse list do.
    constexpr int CH = TBS_CH, NFR = TBS_NFR;            // samples per chunk; contour frames staged per chunk
    __shared__ double s_tot[CH + 1];                     // [0] = the last total of the previous chunk
    __shared__ double s_wrap[CH + 1];
    __shared__ double s_cf0[NFR], s_cv[NFR];
    __shared__ int s_cnt[4];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int T = p.frames ? min(max(p.frames[b], 0), p.T) : p.T;      // an utterance's own frame count, never past the row
    const int ylen = T > 0 ? world_ylen(T, p.frame_period_ms, p.fs) : 0;
    const float* f0 = p.f0 + (size_t)b * p.T;
    unsigned char* vuv = p.vuv + (size_t)b * p.Ymax;
    int* idx = p.idx + (size_t)b * p.Pcap;
    float* xs = p.xshift + (size_t)b * p.Pcap;
    const double fp = p.frame_period, fs = (double)p.fs;
    const double lowest = fs / (double)p.nf + 1.0;
    if (T < 2 || ylen < 2) {              // the reference extrapolates the contour from its last two frames: fewer is undefined there
        if (tid == 0) p.n_pulses[b] = 0;
        return;
    }
    // coarse contour value at frame j (j <= T: the extra point extrapolates linearly), unvoiced (below fs / fft + 1) = 0
    auto cf0 = [&](int j) -> double {
        if (j < T) { const double v = (double)f0[j]; return v < lowest ? 0.0 : v; }
        const double a = (double)f0[T - 1], c = (double)f0[T - 2];
        return (a < lowest ? 0.0 : a) * 2 - (c < lowest ? 0.0 : c);
    };
    auto cvuv = [&](int j) -> double {
        if (j < T) return ((double)f0[j] < lowest) ? 0.0 : 1.0;
        return (((double)f0[T - 1] < lowest) ? 0.0 : 1.0) * 2 - (((double)f0[T - 2] < lowest) ? 0.0 : 1.0);
    };
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_base = 0; s_tot[0] = 0.0; }
    double carry_all = 0.0;                               // every thread: the running phase (total of the last sample so far)
    for (int c0 = 0; c0 < ylen; c0 += CH) {
        const int n = min(CH, ylen - c0);
        // the contour frames this chunk's samples interpolate between, staged once (a chunk spans CH / (fs fp) + 2 frames <= NFR: checked
        // by the launcher)
        int kf = (int)(((double)c0 / fs) / fp) - 1;
        if (kf < 0) kf = 0;
        __syncthreads();                                  // the previous chunk's detection has read s_tot / the contour window
        if (tid == 0) s_tot[0] = carry_all;               // total of sample c0 - 1 (the crossing test of this chunk's first sample reads it)
        if (tid < NFR) {
            const int j = kf + tid;
            s_cf0[tid] = j <= T ? cf0(j) : 0.0;
            s_cv[tid] = j <= T ? cvuv(j) : 0.0;
        }
        __syncthreads();
        for (int j = tid; j < n; j += 256) {
            const int i = c0 + j;
            const double t = (double)i / fs;
            int k = (int)(t / fp) + 1;                   // x[k-1] <= t < x[k] with x[j] = j * fp, fixed up against the products themselves
            if (k < 1) k = 1;
            if (k > T) k = T;
            while (k > 1 && t < (double)(k - 1) * fp) --k;
            while (k < T && t >= (double)k * fp) ++k;
            const double x0 = (double)(k - 1) * fp, x1 = (double)k * fp;
            const double s = (t - x0) / (x1 - x0);
            const int kw = min(max(k - 1 - kf, 0), NFR - 2);         // inside the staged window (the launcher checked that a chunk's frames fit)
            const double fa = s_cf0[kw], fb = s_cf0[kw + 1], va = s_cv[kw], vb = s_cv[kw + 1];
            double fi = fa + s * (fb - fa);
            const double vi = va + s * (vb - va);
            const bool voiced = vi > 0.5;
            if (!voiced) fi = kDefaultF0;
            vuv[i] = voiced ? 1 : 0;
            s_tot[1 + j] = 2.0 * kPi * fi / fs;          // the increment; accumulated in place below
        }
        __syncthreads();
        // The running phase of this chunk.  The reference accumulates sample by sample, total[i] = fl(total[i-1] + inc[i]), and which side
        // of 2 pi a sample lands on can hinge on that exact rounding (header).  But INSIDE one binade of the running sum the rounding is
        // a rounding of the INCREMENT alone: with u = ulp(total), total = A u (A an integer) and inc = q u, the exact sum (A + q) u
        // rounds to (A + rn(q)) u whenever q is not exactly half-way between two integers and the result stays below 2^(e+1).  So the
        // sequential sum of a chunk equals A0 u + (prefix sum of the integers rn(q_i)) u -- an exact INTEGER scan, done in parallel.
        // Chunks in which the sum crosses a power of two, or in which some q_i is an exact tie (then the parity of the running sum
        // decides), take the sequential loop: the first chunk, ~6 crossings of an utterance, a few percent of the rest.
        __shared__ double s_scan[256];
        __shared__ int s_flag;
        const double A0 = carry_all;                      // total of sample c0 - 1, known to every thread (s_tot[0] of the previous round)
        bool serial = !(A0 >= 1.0);
        double u = 0.0, lim = 0.0;
        if (!serial) {
            int e;
            (void)frexp(A0, &e);                          // A0 = m 2^e, 0.5 <= m < 1: the binade is [2^(e-1), 2^e), ulp 2^(e-53)
            u = ldexp(1.0, e - 53);
            lim = ldexp(1.0, e);
        }
        if (tid == 0) s_flag = 0;
        __syncthreads();
        double loc[8];
        double sum = 0.0;
        bool bad = false;
        if (!serial) {
            const double inv_u = 1.0 / u;                 // a power of two: exact
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int j = 8 * tid + e8;
                double r = 0.0;
                if (j < n) {
                    const double q = s_tot[1 + j] * inv_u;
                    r = rint(q);
                    bad |= fabs(q - r) == 0.5;            // exact tie: the running sum's parity decides -> sequential
                }
                sum += r;                                 // integers below 2^53: exact
                loc[e8] = sum;
            }
            s_scan[tid] = sum;
            if (bad) s_flag = 1;
        }
        __syncthreads();
        if (!serial) {
            // exclusive prefix of the 256 per-thread totals (one wave does it: 4 values per lane, then a wave scan)
            if (tid < 64) {
                double v0 = s_scan[4 * tid], v1 = s_scan[4 * tid + 1], v2 = s_scan[4 * tid + 2], v3 = s_scan[4 * tid + 3];
                const double t = v0 + v1 + v2 + v3;
                double inc = t;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double up = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += up;
                }
                const double ex = inc - t;                // exclusive prefix of this lane's group
                s_scan[4 * tid] = ex; s_scan[4 * tid + 1] = ex + v0; s_scan[4 * tid + 2] = ex + v0 + v1; s_scan[4 * tid + 3] = ex + v0 + v1 + v2;
                if (tid == 63 && !(A0 + inc * u < lim)) s_flag = 1;       // the chunk's last total leaves the binade -> sequential
            }
        }
        __syncthreads();
        serial = serial || s_flag != 0;
        if (!serial) {
            const double base = s_scan[tid];
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int j = 8 * tid + e8;
                if (j < n) s_tot[1 + j] = A0 + (base + loc[e8]) * u;      // (integer) u and the sum: both exact
            }
        } else if (tid == 0) {
            double acc = A0;
            int j = 0;
            if (n >= 16) {
                double v[16], w[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = s_tot[1 + e];
                for (; j + 32 <= n; j += 16) {             // the next 16 increments are read while these 16 are added
#pragma unroll
                    for (int e = 0; e < 16; ++e) w[e] = s_tot[1 + j + 16 + e];
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc += v[e]; s_tot[1 + j + e] = acc; }
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = w[e];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc += v[e]; s_tot[1 + j + e] = acc; }
                j += 16;
            }
            for (; j < n; ++j) { acc += s_tot[1 + j]; s_tot[1 + j] = acc; }
        }
        __syncthreads();
        carry_all = s_tot[n];                             // every thread: the running phase after this chunk
        for (int j = tid; j <= n; j += 256) s_wrap[j] = fmod(s_tot[j], 2.0 * kPi);       // one fmod per sample (s_tot[0]: the previous chunk's last)
        __syncthreads();
        // crossings between samples i and i + 1, i = c0 - 1 + j for j = 0 .. n - 1 (i >= 0): |wrap[i + 1] - wrap[i]| > pi
        for (int j0 = 0; j0 < n; j0 += 256) {
            const int j = j0 + tid;
            const int i = c0 - 1 + j;
            bool hit = false;
            double w0 = 0.0, w1 = 0.0;
            if (j < n && i >= 0) {
                w0 = s_wrap[j];
                w1 = s_wrap[j + 1];
                hit = fabs(w1 - w0) > kPi;
            }
            const unsigned long long m = __ballot(hit);
            if (lane == 0) s_cnt[wave] = __popcll(m);
            __syncthreads();
            int before = s_base;
            for (int w = 0; w < wave; ++w) before += s_cnt[w];
            if (hit) {
                const int slot = before + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < p.Pcap) {
                    idx[slot] = i;
                    const double y1 = w0 - 2.0 * kPi;
                    xs[slot] = (float)(-y1 / (w1 - y1));  // fraction of a sample to the exact crossing, in [0, 1)
                }
            }
            __syncthreads();
            if (tid == 0) s_base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            __syncthreads();
        }
    }
    if (tid == 0) p.n_pulses[b] = s_base <= p.Pcap ? s_base : -1;       // -1: more pulses than the caller made room for
}

// 1b. the same time base with the chunks of an utterance on DIFFERENT workgroups (round 4, late): the only thing a chunk needs from
// its predecessor is the exact running phase at its first sample -- one double -- and, for its slice of the pulse list, the number of
// pulses before it -- one int.  Both travel down a chain of per-chunk mailboxes in global memory (value, then a flag with release /
// acquire at agent scope: workgroups of one utterance may sit on different XCDs).  Chunk numbers are handed out by a per-utterance
// ticket, so a chunk's predecessor is always held by a workgroup that is already running (no dependence on dispatch order).  What is on
// the chain per chunk is the exact integer scan alone (~2 us); interpolation before it and wrap / detect / compact after it overlap
// with the neighbours'.  A chunk in which the running sum crosses a power of two is scanned in two segments (ulp u before the crossing
// sample, the reference's own fl(total + inc) AT it, ulp 2u after); exact ties and the first chunk (total < 1) take the one-lane loop.
constexpr int TBC_CH = 8192, TBC_NFR = 128, TBC_NT = 1024;          // samples per chunk of the chained time base; contour frames staged per chunk
struct TbChain { int* ticket; int* cflag; double* cval; int* nflag; int* nval; int nchunks; };

#ifdef TBC_DEBUG
__device__ long long g_tbc_dbg[8 * 64];
#endif
__global__ __launch_bounds__(TBC_NT) void world_timebase_chain_kernel(WorldParams p, TbChain ch) {
    constexpr int CH = TBC_CH, NFR = TBC_NFR, NTH = TBC_NT, PER = CH / NTH, NW = NTH / 64, PL = NTH / 64;     // PL: per-thread sums a lane of the scanning wave takes
    // s_tot[TI(j)]: one pad slot per PER entries, so that the scans' per-thread runs of PER consecutive samples (thread stride PER + 1 doubles)
    // do not all start in the same banks; consecutive j stay consecutive.  j = 0 holds the previous chunk's last total.
    __shared__ double s_tot[CH + 1 + (CH + 1) / PER + 1];
#define TI(j) ((j) + ((j) / PER))
    __shared__ double s_cf0[NFR], s_cv[NFR];
    __shared__ double s_scan[NTH];
    __shared__ int s_cnt[NW];
    __shared__ int s_flag, s_chunk, s_first, s_base;
    __shared__ double s_carry, s_tk;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.frames ? min(max(p.frames[b], 0), p.T) : p.T;
    const int ylen = T > 0 ? world_ylen(T, p.frame_period_ms, p.fs) : 0;
    if (tid == 0) s_chunk = atomicAdd(ch.ticket + b, 1);
    __syncthreads();
    const int c = s_chunk, c0 = c * CH;
#ifdef TBC_DEBUG          /* -DTBC_DEBUG: wall-clock stamps of utterance 0's chunks (tools/tbc_timeline.py): how the 64-way bank conflict of the first version was found */
#define TBC_STAMP(slot) if (tid == 0 && b == 0) g_tbc_dbg[c * 8 + (slot)] = wall_clock64()
#else
#define TBC_STAMP(slot)
#endif
    TBC_STAMP(0);
    if (T < 2 || ylen < 2) {
        if (c == 0 && tid == 0) p.n_pulses[b] = 0;
        return;
    }
    if (c0 >= ylen) return;
    const int n = min(CH, ylen - c0);
    const bool last = c0 + n >= ylen;
    const float* f0 = p.f0 + (size_t)b * p.T;
    unsigned char* vuv = p.vuv + (size_t)b * p.Ymax;
    int* idx = p.idx + (size_t)b * p.Pcap;
    float* xs = p.xshift + (size_t)b * p.Pcap;
    int* cflag = ch.cflag + (size_t)b * ch.nchunks;
    double* cval = ch.cval + (size_t)b * ch.nchunks;
    int* nflag = ch.nflag + (size_t)b * ch.nchunks;
    int* nval = ch.nval + (size_t)b * ch.nchunks;
    const double fp = p.frame_period, fs = (double)p.fs;
    const double lowest = fs / (double)p.nf + 1.0;
    auto cf0 = [&](int j) -> double {
        if (j < T) { const double v = (double)f0[j]; return v < lowest ? 0.0 : v; }
        const double a = (double)f0[T - 1], cc = (double)f0[T - 2];
        return (a < lowest ? 0.0 : a) * 2 - (cc < lowest ? 0.0 : cc);
    };
    auto cvuv = [&](int j) -> double {
        if (j < T) return ((double)f0[j] < lowest) ? 0.0 : 1.0;
        return (((double)f0[T - 1] < lowest) ? 0.0 : 1.0) * 2 - (((double)f0[T - 2] < lowest) ? 0.0 : 1.0);
    };
    // ---- increments of this chunk (independent of every other chunk) ----
    int kf = (int)(((double)c0 / fs) / fp) - 1;
    if (kf < 0) kf = 0;
    if (tid < NFR) {
        const int j = kf + tid;
        s_cf0[tid] = j <= T ? cf0(j) : 0.0;
        s_cv[tid] = j <= T ? cvuv(j) : 0.0;
    }
    __syncthreads();
    for (int j = tid; j < n; j += NTH) {
        const int i = c0 + j;
        const double t = (double)i / fs;
        int k = (int)(t / fp) + 1;
        if (k < 1) k = 1;
        if (k > T) k = T;
        while (k > 1 && t < (double)(k - 1) * fp) --k;
        while (k < T && t >= (double)k * fp) ++k;
        const double x0 = (double)(k - 1) * fp, x1 = (double)k * fp;
        const double sft = (t - x0) / (x1 - x0);
        const int kw = min(max(k - 1 - kf, 0), NFR - 2);
        const double fa = s_cf0[kw], fb = s_cf0[kw + 1], va = s_cv[kw], vb = s_cv[kw + 1];
        double fi = fa + sft * (fb - fa);
        const double vi = va + sft * (vb - va);
        const bool voiced = vi > 0.5;
        if (!voiced) fi = kDefaultF0;
        vuv[i] = voiced ? 1 : 0;
        s_tot[TI(1 + j)] = 2.0 * kPi * fi / fs;
    }
    TBC_STAMP(1);
    // ---- the running phase at this chunk's first sample: from the predecessor's mailbox ----
    if (tid == 0) {
        double a0 = 0.0;
        if (c > 0) {
            // poll with RELAXED loads (an acquire per poll would invalidate this XCD's L2 each time, under everybody else's feet), one acquire
            // fence once the flag is up
            while (__hip_atomic_load(cflag + c - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            a0 = __hip_atomic_load(cval + c - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_carry = a0;
        s_tot[TI(0)] = a0;
        s_flag = 0;
        s_first = n;                                      // first sample at which the sum has left the binade (n: none)
    }
    __syncthreads();
    TBC_STAMP(2);
    const double A0 = s_carry;
    bool serial = !(A0 > 0.0) && c > 0;                  // (a non-positive running phase: degenerate input)
    double u = 0.0, lim = 0.0;
    // exact integer scan of rint(inc / unit) over samples [from, n) on top of `start`: s_tot[TI(1 + j)] = start + prefix * unit.  Returns
    // (through s_flag) whether a tie was met; (through s_first) the first sample whose total is >= limit.
    double incr[PER];                                     // this thread's run of increments, read once (the rounds below re-scan them)
#pragma unroll
    for (int e8 = 0; e8 < PER; ++e8) {
        const int j = PER * tid + e8;
        incr[e8] = j < n ? s_tot[TI(1 + j)] : 0.0;
    }
    auto scan = [&](int from, double start, double unit, double limit) {
        const double inv = 1.0 / unit;
        double loc[PER];
        double sum = 0.0;
        unsigned ties = 0;                                 // samples whose increment is EXACTLY half-way between two multiples of the unit
#pragma unroll
        for (int e8 = 0; e8 < PER; ++e8) {
            const int j = PER * tid + e8;
            double r = 0.0;
            if (j >= from && j < n) {
                const double q = incr[e8] * inv;
                r = rint(q);
                if (fabs(q - r) == 0.5) ties |= 1u << e8;
            }
            sum += r;
            loc[e8] = sum;
        }
        s_scan[tid] = sum;
        __syncthreads();
        if (tid < 64) {
            double v[PL];
            double t4 = 0.0;
#pragma unroll
            for (int q = 0; q < PL; ++q) { v[q] = s_scan[PL * tid + q]; t4 += v[q]; }
            double inc = t4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double up = __shfl_up(inc, o, 64);
                if (lane >= o) inc += up;
            }
            double ex = inc - t4;
#pragma unroll
            for (int q = 0; q < PL; ++q) { s_scan[PL * tid + q] = ex; ex += v[q]; }
        }
        __syncthreads();
        const double base = s_scan[tid];
        int firstbad = n;
#pragma unroll
        for (int e8 = 0; e8 < PER; ++e8) {
            const int j = PER * tid + e8;
            if (j >= from && j < n) {
                const double v = start + (base + loc[e8]) * unit;
                // a break point: the sum leaves the binade here, or this sample is a tie (round-half-even looks at the running sum's parity):
                // everything BEFORE it is exact, the sample itself is added as the reference adds it, the scan restarts behind it
                if ((!(v < limit) || ((ties >> e8) & 1u)) && j < firstbad) firstbad = j;
                loc[e8] = v;
            }
        }
        if (firstbad < n) atomicMin(&s_first, firstbad);
        __syncthreads();
        const int stop = s_first;
#pragma unroll
        for (int e8 = 0; e8 < PER; ++e8) {
            const int j = PER * tid + e8;
            if (j >= from && j < n && j < stop) s_tot[TI(1 + j)] = loc[e8];
        }
        __syncthreads();
    };
    // segments between the samples at which the running sum leaves its binade: each an exact integer scan; the crossing sample itself is
    // added as the reference adds it (one fp64 addition).  The utterance's first sample starts from 0 (0 + inc is exact).
    if (!serial || A0 == 0.0) {
        serial = false;
        int pos = 0;
        double cur = A0;
        if (cur == 0.0) {
            // the utterance's first samples: the sum runs through a binade every few samples and most increments are ties at this
            // scale -- one lane adds them in order until the sum has reached 64 (at most 512 samples), the segments take over from there
            if (tid == 0) {
                double acc = 0.0;
                int j = 0;
                const int m = n < 512 ? n : 512;
                for (; j < m && acc < 64.0; ++j) { acc += s_tot[TI(1 + j)]; s_tot[TI(1 + j)] = acc; }
                s_tk = acc;
                s_first = j;
            }
            __syncthreads();
            cur = s_tk;
            pos = s_first;
            __syncthreads();
        }
        int rounds = 0;
        while (pos < n && !serial) {
            if (++rounds > 96) { serial = true; break; }  // pathological contours (a tie on every sample of a long stretch): the one-lane loop
            int e;
            (void)frexp(cur, &e);
            u = ldexp(1.0, e - 53);
            lim = ldexp(1.0, e);
            if (tid == 0) s_first = n;
            __syncthreads();
            scan(pos, cur, u, lim);
            const int k = s_first;
            if (k >= n) break;
            if (tid == 0) {
                s_tk = s_tot[TI(k)] + s_tot[TI(1 + k)];            // s_tot[TI(k)] = total of sample k - 1 (exact), s_tot[TI(1 + k)] still the increment
                s_tot[TI(1 + k)] = s_tk;
            }
            __syncthreads();
            cur = s_tk;
            pos = k + 1;
        }
    }
    if (serial) {
        // one lane, in order.  The increments of samples the scans above already replaced by totals must be rebuilt: recompute nothing,
        // re-derive from the contour (cheap: serial chunks are the first one and the rare ties)
        __syncthreads();
        for (int j = tid; j < n; j += NTH) {
            const int i = c0 + j;
            const double t = (double)i / fs;
            int k = (int)(t / fp) + 1;
            if (k < 1) k = 1;
            if (k > T) k = T;
            while (k > 1 && t < (double)(k - 1) * fp) --k;
            while (k < T && t >= (double)k * fp) ++k;
            const double x0 = (double)(k - 1) * fp, x1 = (double)k * fp;
            const double sft = (t - x0) / (x1 - x0);
            const int kw = min(max(k - 1 - kf, 0), NFR - 2);
            double fi = s_cf0[kw] + sft * (s_cf0[kw + 1] - s_cf0[kw]);
            const double vi = s_cv[kw] + sft * (s_cv[kw + 1] - s_cv[kw]);
            if (!(vi > 0.5)) fi = kDefaultF0;
            s_tot[TI(1 + j)] = 2.0 * kPi * fi / fs;
        }
        __syncthreads();
        if (tid == 0) {
            double acc = A0;
            for (int j = 0; j < n; ++j) { acc += s_tot[TI(1 + j)]; s_tot[TI(1 + j)] = acc; }
        }
        __syncthreads();
    }
    TBC_STAMP(3);
    // ---- hand the running phase on, then the work that nobody waits for ----
    if (tid == 0 && !last) {
        __hip_atomic_store(cval + c, s_tot[TI(n)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cflag + c, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int j = tid; j <= n; j += NTH) s_tot[TI(j)] = fmod(s_tot[TI(j)], 2.0 * kPi);      // in place: nobody needs the unwrapped phase any more
    __syncthreads();
    TBC_STAMP(4);
    // crossings between samples i and i + 1, i = c0 - 1 + j (i >= 0): thread t owns the PER consecutive j of its run, so the pulse order
    // is thread order: count, one workgroup scan, place behind the predecessors' pulses
    unsigned hits = 0;
#pragma unroll
    for (int e8 = 0; e8 < PER; ++e8) {
        const int j = PER * tid + e8, i = c0 - 1 + j;
        if (j < n && i >= 0 && fabs(s_tot[TI(j + 1)] - s_tot[TI(j)]) > kPi) hits |= 1u << e8;
    }
    const int mine = __popc(hits);
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_cnt[wave] = inc;
    __syncthreads();
    int total = 0, mybase = inc - mine;
    for (int w = 0; w < NW; ++w) {
        if (w < wave) mybase += s_cnt[w];
        total += s_cnt[w];
    }
    if (tid == 0) {
        int before = 0;
        if (c > 0) {
            while (__hip_atomic_load(nflag + c - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            before = __hip_atomic_load(nval + c - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!last) {
            __hip_atomic_store(nval + c, before + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(nflag + c, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            p.n_pulses[b] = before + total <= p.Pcap ? before + total : -1;
        }
        s_base = before;
    }
    __syncthreads();
    int slot = s_base + mybase;
#pragma unroll
    for (int e8 = 0; e8 < PER; ++e8) {
        if ((hits >> e8) & 1u) {
            const int j = PER * tid + e8;
            if (slot < p.Pcap) {
                const double w0 = s_tot[TI(j)], w1 = s_tot[TI(j + 1)];
                idx[slot] = c0 - 1 + j;
                const double y1 = w0 - 2.0 * kPi;
                xs[slot] = (float)(-y1 / (w1 - y1));
            }
            ++slot;
        }
    }
    TBC_STAMP(5);
}
#undef TI

// frame pair and interpolation weight of the pulse at sample `id` (GetSpectralEnvelope / GetAperiodicRatio: floor / ceil of the frame
// position in double)
__device__ __forceinline__ void world_frame_mix(int id, int fs, double frame_period, int T, int& fl, int& ce, float& mix) {
    const double pos = ((double)id / (double)fs) / frame_period;
    const int f0_ = (int)floor(pos), c0_ = (int)ceil(pos);
    fl = f0_ < T - 1 ? f0_ : T - 1;
    ce = c0_ < T - 1 ? c0_ : T - 1;
    mix = fl == ce ? 0.f : (float)(pos - (double)fl);
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------------------------------------
// 2. one response per pulse
struct cplx { float re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx mul_mi(cplx a) { return {a.im, -a.re}; }          // a * (-i)
__device__ __forceinline__ void radix4(cplx (&a)[4]) {
    const cplx s0 = cadd(a[0], a[2]), s1 = csub(a[0], a[2]), s2 = cadd(a[1], a[3]), s3 = mul_mi(csub(a[1], a[3]));
    a[0] = cadd(s0, s2); a[2] = csub(s0, s2); a[1] = cadd(s1, s3); a[3] = csub(s1, s3);
}
__device__ __forceinline__ int rev4x4(int k) {
    return ((k & 3) << 6) | (((k >> 2) & 3) << 4) | (((k >> 4) & 3) << 2) | ((k >> 6) & 3);
}
#define WAVE_LDS_FENCE() asm volatile("" ::: "memory")     /* wave-local exchange through LDS: program order is enough */

// 256-point forward complex FFT of a[q] = input[lane + 64 q]; on return Z[k] sits at z[rev4x4(k)]  (mel.hip's transform)
__device__ __forceinline__ void fft256(cplx (&a)[4], float2* z, int lane, const cplx (&tw)[3][3]) {
    radix4(a);
#pragma unroll
    for (int qq = 1; qq < 4; ++qq) a[qq] = cmul(a[qq], tw[0][qq - 1]);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) z[lane + 64 * qq] = make_float2(a[qq].re, a[qq].im);
    WAVE_LDS_FENCE();
#pragma unroll
    for (int s = 1; s < 4; ++s) {
        const int quarter = 64 >> (2 * s);
        const int g = (lane / quarter) * (4 * quarter), j = lane & (quarter - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float2 v = z[g + j + q * quarter]; a[q] = {v.x, v.y}; }
        WAVE_LDS_FENCE();
        radix4(a);
        if (s < 3) {
#pragma unroll
            for (int qq = 1; qq < 4; ++qq) a[qq] = cmul(a[qq], tw[s][qq - 1]);
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) z[g + j + qq * quarter] = make_float2(a[qq].re, a[qq].im);
        WAVE_LDS_FENCE();
    }
}

// Real FFT of the 512 real samples held as r[0..511] in LDS: X[k], k = lane + 64 q (q = 0..3) into x[q], X[256] into x256 (all lanes).
// Forward transform, unnormalised.  Clobbers z.
__device__ __forceinline__ void rfft512(const float* r, float2* z, int lane, const cplx (&tw)[3][3], const cplx (&w512)[4], cplx (&x)[4],
                                        cplx& x256) {
    cplx a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float2 v = *reinterpret_cast<const float2*>(r + 2 * (lane + 64 * q)); a[q] = {v.x, v.y}; }
    WAVE_LDS_FENCE();
    fft256(a, z, lane, tw);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        const float2 zk = z[rev4x4(k)], zm = z[rev4x4((256 - k) & 255)];
        const cplx e = {0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
        const cplx d = {0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y)};
        x[q] = cadd(e, cmul(w512[q], mul_mi(d)));
    }
    const float2 z0 = z[0];
    x256 = {z0.x - z0.y, 0.f};                           // X[N/2] = E[0] - O[0] = Re Z[0] - Im Z[0]
    WAVE_LDS_FENCE();
}

// Inverse (backward, unnormalised: x[n] = sum_k X[k] e^{+2 pi i k n / 512} over the Hermitian extension) real FFT.  X[k] for k = lane + 64 q
// in x[q], X[256] in x256 (real).  The spectrum is first parked in LDS (sre / sim, 257 floats each) because bin 256 - k belongs to another
// lane; the 512 real samples come back in r[0..511].
__device__ __forceinline__ void irfft512(const cplx (&x)[4], cplx x256, float* sre, float* sim, float2* z, float* r, int lane,
                                         const cplx (&tw)[3][3], const cplx (&w512)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { sre[lane + 64 * q] = x[q].re; sim[lane + 64 * q] = x[q].im; }
    if (lane == 0) { sre[256] = x256.re; sim[256] = 0.f; }
    WAVE_LDS_FENCE();
    cplx a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        const cplx xk = {sre[k], sim[k]}, xm = {sre[256 - k], -sim[256 - k]};           // X[k], conj X[256 - k]
        const cplx e = {0.5f * (xk.re + xm.re), 0.5f * (xk.im + xm.im)};                 // spectrum of the even samples
        const cplx d = {0.5f * (xk.re - xm.re), 0.5f * (xk.im - xm.im)};
        const cplx o = cmul(d, cplx{w512[q].re, -w512[q].im});                           // ... of the odd samples: (X[k] - conj X[256-k]) / 2 * W^-k
        // Z[k] = E[k] + i O[k]; the inverse transform is conj(FFT(conj Z))
        a[q] = {e.re - o.im, -(e.im + o.re)};
    }
    WAVE_LDS_FENCE();
    fft256(a, z, lane, tw);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = lane + 64 * q;
        const float2 v = z[rev4x4(n)];
        *reinterpret_cast<float2*>(r + 2 * n) = make_float2(2.f * v.x, -2.f * v.y);      // x[2n], x[2n+1]; factor 2: 512-point sum from a 256-point one
    }
    WAVE_LDS_FENCE();
}

// minimum-phase spectrum exp(FFT(fold(IFFT(mirror(L))))) of the log-amplitude L[k] (k = lane + 64 q in l[q], L[256] in l256)
__device__ __forceinline__ void min_phase(const float (&l)[4], float l256, float* r, float2* z, int lane, const cplx (&tw)[3][3],
                                          const cplx (&w512)[4], cplx (&h)[4], cplx& h256) {
    // The mean of the mirrored log spectrum is taken out first and put back in the exponent at the end (exact: it only moves c[0]).  A
    // log-amplitude around -9 (aperiodicity 0.001 under a 1e-2 envelope) otherwise makes every butterfly carry sums of ~ -4700, whose
    // fp32 rounding (3e-4 absolute) spreads over all cepstral coefficients and comes back through the exponential as a 1e-4
    // relative error of the response; without the mean the sums are the spectrum's variation only.
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) part += (lane + 64 * q == 0) ? l[q] : 2.f * l[q];      // bins 1 .. 255 appear twice in the mirrored sequence
    const float mu = (wave_sum(part) + l256) * (1.f / 512.f);
    // mirrored real sequence of 512: r[j] = L[j] (j <= 256), L[512 - j] beyond
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        r[k] = l[q] - mu;
        if (k > 0) r[512 - k] = l[q] - mu;
    }
    if (lane == 0) r[256] = l256 - mu;
    WAVE_LDS_FENCE();
    cplx c[4], c256;
    rfft512(r, z, lane, tw, w512, c, c256);             // real cepstrum x 512 (imaginary parts are round-off: the input is even)
    // fold: c[0], 2 c[1..255], c[256], zeros beyond
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        r[k] = c[q].re * (k == 0 ? (1.f / 512.f) : (2.f / 512.f));
        r[256 + k] = (k == 0) ? c256.re * (1.f / 512.f) : 0.f;
    }
    WAVE_LDS_FENCE();
    cplx m[4], m256;
    rfft512(r, z, lane, tw, w512, m, m256);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float e = expf(m[q].re + mu);
        float sn, cs;
        sincosf(m[q].im, &sn, &cs);
        h[q] = {e * cs, e * sn};
    }
    h256 = {expf(m256.re + mu), 0.f};                       // the folded cepstrum is real: bin 256 of its spectrum is real
}

__global__ __launch_bounds__(256, 3) void world_pulse_kernel(WorldParams p) {     // 3 waves / SIMD (168 VGPRs, 100 B of scratch): 0.36 -> 0.32 ms; 4 spills 248 B and loses
    __shared__ float2 zbuf[4][256];
    __shared__ float rbuf[4][512];
    __shared__ float sbuf[4][2][260];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int P = p.n_pulses[b];
    if (P <= 0) return;
    float2* z = zbuf[wave];
    float* r = rbuf[wave];
    float* sre = sbuf[wave][0];
    float* sim = sbuf[wave][1];
    cplx tw[3][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int quarter = 64 >> (2 * s);
        const int j = lane & (quarter - 1);
        const int mult = 1 << (2 * s);
#pragma unroll
        for (int qq = 1; qq < 4; ++qq) {
            const int k = (mult * j * qq) & 255;
            tw[s][qq - 1] = {p.tw256[2 * k], p.tw256[2 * k + 1]};
        }
    }
    cplx w512[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int k = lane + 64 * q; w512[q] = {p.tw512[2 * k], p.tw512[2 * k + 1]}; }
    const int T = p.frames ? min(max(p.frames[b], 0), p.T) : p.T;      // an utterance's own frame count, never past the row
    const int* idx = p.idx + (size_t)b * p.Pcap;
    const float* xs = p.xshift + (size_t)b * p.Pcap;
    const unsigned char* vuv = p.vuv + (size_t)b * p.Ymax;
    const float* spb = p.sp + (size_t)b * p.T * NB;
    const float* apb = p.ap + (size_t)b * p.T * NB;
    float* respb = p.resp + (size_t)b * p.Pcap * NF;
    const int idx0 = idx[0];
    for (int pi = blockIdx.x * 4 + wave; pi < P; pi += gridDim.x * 4) {
        const int id = idx[pi];
        const int ns = idx[pi + 1 < P ? pi + 1 : P - 1] - id;           // samples to the next pulse (0 for the last one)
        const bool cur_v = vuv[id] != 0;
        int fl, ce;
        float mix;
        world_frame_mix(id, p.fs, p.frame_period, T, fl, ce, mix);
        // u = 1 - aperiodicity, carried instead of the aperiodicity itself: near 1 (every voiced frame above a few kHz) the fp32 value of a
        // has an absolute error of 6e-8, i.e. 1 - a^2 -- whose logarithm shapes the periodic response's minimum phase EVERYWHERE -- a
        // relative error of 1e-4 .. 1e-3; 1 - a of a stored fp32 a is exact, the interpolation of u is accurate to fp32's relative
        // precision, and 1 - a^2 = u (2 - u).  With `coded` the band values are decoded here (u = -expm1(dB ln10 / 20)), so nothing near 1 is
        // ever rounded to fp32.  WORLD's clip of a to [0.001, 1 - 1e-12] is u in [1e-12, 0.999].
        float env[4], rat[4], omr[4], env256, rat256, omr256;                    // omr = 1 - ratio
        {
            const float* s0 = spb + (size_t)fl * NB;
            const float* s1 = spb + (size_t)ce * NB;
            const float* a0 = apb + (size_t)fl * NB;
            const float* a1 = apb + (size_t)ce * NB;
            const float* c0 = p.coded ? p.coded + ((size_t)b * p.T + fl) * p.nb : nullptr;
            const float* c1 = p.coded ? p.coded + ((size_t)b * p.T + ce) * p.nb : nullptr;
            auto u_of = [&](const float* arow, const float* crow, int k) -> float {
                float u;
                if (crow) {
                    float mean = 0.f;
                    for (int i = 0; i < p.nb; ++i) mean += crow[i];
                    if (mean / (float)p.nb > -0.5f) return 1e-12f;              // unvoiced frame: a = 1 - 1e-12
                    const float f = (float)p.fs / (float)NF * (float)k;
                    int seg = (int)(f / 3000.f);
                    if (seg > p.nb) seg = p.nb;
                    const float x0 = 3000.f * seg, x1 = seg == p.nb ? 0.5f * p.fs : 3000.f * (seg + 1);
                    const float y0 = seg == 0 ? -60.f : crow[seg - 1], y1 = seg == p.nb ? -1e-12f : crow[seg];
                    const float db = y0 + (f - x0) / (x1 - x0) * (y1 - y0);
                    u = -expm1f(db * 0.11512925464970229f);                      // 1 - 10^(dB / 20)
                } else {
                    u = 1.f - arow[k];
                }
                return fminf(fmaxf(u, 1e-12f), 0.999f);
            };
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = lane + 64 * q;
                env[q] = (1.f - mix) * fabsf(s0[k]) + mix * fabsf(s1[k]);
                const float u = (1.f - mix) * u_of(a0, c0, k) + mix * u_of(a1, c1, k);
                rat[q] = (1.f - u) * (1.f - u);
                omr[q] = u * (2.f - u);
            }
            env256 = (1.f - mix) * fabsf(s0[256]) + mix * fabsf(s1[256]);
            const float u = (1.f - mix) * u_of(a0, c0, 256) + mix * u_of(a1, c1, 256);
            rat256 = (1.f - u) * (1.f - u);
            omr256 = u * (2.f - u);
        }
        const float rat0 = __shfl(rat[0], 0, 64);
        const bool periodic = cur_v && !(rat0 > 0.999f);
        // ---- periodic part ----
        float per[8];                                    // this lane's 8 samples of the (fftshifted) periodic response: n = 2 (lane + 64 q) + e
        if (periodic) {
            float l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) l[q] = 0.5f * logf(env[q] * omr[q] + 1e-12f);
            const float l256 = 0.5f * logf(env256 * omr256 + 1e-12f);
            cplx h[4], h256;
            min_phase(l, l256, r, z, lane, tw, w512, h, h256);
            // fractional delay of xs samples: multiply by cos - i |sin| of 2 pi x k / 512
            const float coef = 6.283185307179586f * xs[pi] / 512.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sn, cs;
                sincosf(coef * (float)(lane + 64 * q), &sn, &cs);
                h[q] = cmul(h[q], cplx{cs, -fabsf(sn)});
            }
            {
                float sn, cs;
                sincosf(coef * 256.f, &sn, &cs);
                h256 = {h256.re * cs, 0.f};              // the c2r transform ignores the imaginary part of bin N/2
            }
            irfft512(h, h256, sre, sim, z, r, lane, tw, w512);
            // fftshift + DC removal: dc = sum of the causal half (unshifted samples 0..255)
            float part = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) { const float2 v = *reinterpret_cast<const float2*>(r + 2 * (lane + 64 * q)); part += v.x + v.y; }
            const float dc = wave_sum(part);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 2 * (lane + 64 * q);       // position in the shifted response
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = n + e;
                    const float w = p.dcr[i];
                    per[2 * q + e] = i < NH ? -dc * w : r[i - NH] - dc * w;
                }
            }
            WAVE_LDS_FENCE();
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) per[e] = 0.f;
        }
        // ---- aperiodic part: zero-mean noise burst of ns samples, coloured ----
        cplx nz[4], nz256;
        {
            const long base = (long)id - idx0;
            const int nuse = ns < NF ? ns : NF;
            float sum = 0.f;
            for (int i = lane; i < ns; i += 64) sum += (base + i < p.table_len) ? p.randn[base + i] : 0.f;
            const float mean = ns > 0 ? wave_sum(sum) / (float)ns : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = lane + 64 * q;
                r[i] = (i < nuse && base + i < p.table_len) ? p.randn[base + i] - mean : 0.f;
            }
            WAVE_LDS_FENCE();
            rfft512(r, z, lane, tw, w512, nz, nz256);
        }
        {
            float l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) l[q] = 0.5f * logf(cur_v ? env[q] * rat[q] : env[q]);
            const float l256 = 0.5f * logf(cur_v ? env256 * rat256 : env256);
            cplx h[4], h256;
            min_phase(l, l256, r, z, lane, tw, w512, h, h256);
#pragma unroll
            for (int q = 0; q < 4; ++q) h[q] = cmul(h[q], nz[q]);
            h256 = {h256.re * nz256.re, 0.f};
            irfft512(h, h256, sre, sim, z, r, lane, tw, w512);
        }
        const float sq = sqrtf((float)ns);
        float* out = respb + (size_t)pi * NF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 2 * (lane + 64 * q);
            const int src = n < NH ? n + NH : n - NH;    // fftshift of the aperiodic response
            const float2 a = *reinterpret_cast<const float2*>(r + src);
            *reinterpret_cast<float2*>(out + n) = make_float2((per[2 * q] * sq + a.x) * (1.f / 512.f), (per[2 * q + 1] * sq + a.y) * (1.f / 512.f));
        }
        WAVE_LDS_FENCE();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------------------
// 2b. one response per pulse at ANY power-of-two fft size (22.05 kHz models: 1024), in fp64: one workgroup per pulse, the transforms as
// radix-2 FFTs on an LDS array (world_f64.h).  Same steps as world_pulse_kernel; in double none of its fp32 precautions (mean removal,
// carrying 1 - a) are needed.  Twiddles and the DC remover are computed here, no tables.
#pragma clang fp contract(off)
__device__ void min_phase_f64(const double* lg, cd* buf, cd* out, int N, int logN, const cd* tw) {
    const int H = N / 2;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) buf[brev(i, logN)] = {lg[i <= H ? i : N - i], 0.0};
    __syncthreads();
    fft_lds(buf, logN, tw, logN, false);
    double mine[8];                                       // folded cepstrum of this thread's positions (N <= 8 * blockDim.x)
    int c = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x, ++c) {
        const double v = buf[i].x / (double)N;
        mine[c] = i == 0 || i == H ? v : (i < H ? 2.0 * v : 0.0);
    }
    __syncthreads();
    c = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x, ++c) buf[brev(i, logN)] = {mine[c], 0.0};
    __syncthreads();
    fft_lds(buf, logN, tw, logN, false);
    for (int k = threadIdx.x; k <= H; k += blockDim.x) {
        const double e = exp(buf[k].x);
        out[k] = {e * cos(buf[k].y), e * sin(buf[k].y)};
    }
    __syncthreads();
}
// unnormalised inverse real transform of the half spectrum hs[0 .. N/2] (imaginary parts of bins 0 and N/2 ignored) -> buf[n].x
__device__ void irfft_f64(const cd* hs, cd* buf, int N, int logN, const cd* tw) {
    const int H = N / 2;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        cd v;
        if (i == 0 || i == H) v = {hs[i].x, 0.0};
        else if (i < H) v = hs[i];
        else v = {hs[N - i].x, -hs[N - i].y};
        buf[brev(i, logN)] = v;
    }
    __syncthreads();
    fft_lds(buf, logN, tw, logN, true);
}

__global__ __launch_bounds__(256) void world_pulse_f64_kernel(WorldParams p, int logN) {
    extern __shared__ double shd[];
    __shared__ double red[16];
    const int N = 1 << logN, H = N / 2, NBN = H + 1;
    cd* buf = (cd*)shd;                        // N
    cd* tw = buf + N;                          // H
    cd* hs = tw + H;                           // H + 2
    cd* nz = hs + H + 2;                       // H + 2
    double* env = (double*)(nz + H + 2);       // H + 2 each
    double* rat = env + H + 2;
    double* lg = rat + H + 2;
    double* per = lg + H + 2;                  // N
    const int b = blockIdx.y, P = p.n_pulses[b];
    if (P <= 0) return;
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        double sn, cs;
        sincospi(2.0 * (double)k / (double)N, &sn, &cs);
        tw[k] = {cs, -sn};
    }
    double hsum = 0.0;
    for (int i = threadIdx.x; i < H; i += blockDim.x) hsum += 0.5 - 0.5 * cos(2.0 * kPi * ((double)i + 1.0) / (1.0 + (double)N));
    hsum = 2.0 * block_sum(hsum, red);
    auto dcr = [&](int i) -> double {                     // GetDCRemover: a Hann window of unit sum, mirrored
        const int j = i < H ? i : N - 1 - i;
        return (0.5 - 0.5 * cos(2.0 * kPi * ((double)j + 1.0) / (1.0 + (double)N))) / hsum;
    };
    const int T = p.frames ? min(max(p.frames[b], 0), p.T) : p.T;
    const int* idx = p.idx + (size_t)b * p.Pcap;
    const float* xs = p.xshift + (size_t)b * p.Pcap;
    const unsigned char* vuv = p.vuv + (size_t)b * p.Ymax;
    const float* spb = p.sp + (size_t)b * p.T * NBN;
    const float* apb = p.ap + (size_t)b * p.T * NBN;
    float* respb = p.resp + (size_t)b * p.Pcap * N;
    const int idx0 = idx[0];
    for (int pi = blockIdx.x; pi < P; pi += gridDim.x) {
        const int id = idx[pi];
        const int ns = idx[pi + 1 < P ? pi + 1 : P - 1] - id;
        const bool cur_v = vuv[id] != 0;
        const double pos = ((double)id / (double)p.fs) / p.frame_period;
        const int f0_ = (int)floor(pos), c0_ = (int)ceil(pos);
        const int fl = f0_ < T - 1 ? f0_ : T - 1, ce = c0_ < T - 1 ? c0_ : T - 1;
        const double mix = fl == ce ? 0.0 : pos - (double)fl;
        const float* s0 = spb + (size_t)fl * NBN;
        const float* s1 = spb + (size_t)ce * NBN;
        const float* a0 = apb + (size_t)fl * NBN;
        const float* a1 = apb + (size_t)ce * NBN;
        const float* c0 = p.coded ? p.coded + ((size_t)b * p.T + fl) * p.nb : nullptr;
        const float* c1 = p.coded ? p.coded + ((size_t)b * p.T + ce) * p.nb : nullptr;
        auto a_of = [&](const float* arow, const float* crow, int k) -> double {
            double a;
            if (crow) {                                    // DecodeAperiodicity of this frame's bands, in double
                double mean = 0.0;
                for (int i = 0; i < p.nb; ++i) mean += (double)crow[i];
                if (mean / (double)p.nb > -0.5) a = 1.0 - 1e-12;
                else {
                    const double f = (double)k * (double)p.fs / (double)N;
                    int seg = (int)(f / 3000.0);
                    if (seg > p.nb) seg = p.nb;
                    const double x0 = 3000.0 * seg, x1 = seg == p.nb ? 0.5 * p.fs : 3000.0 * (seg + 1);
                    const double y0 = seg == 0 ? -60.0 : (double)crow[seg - 1], y1 = seg == p.nb ? -1e-12 : (double)crow[seg];
                    a = pow(10.0, (y0 + (f - x0) / (x1 - x0) * (y1 - y0)) / 20.0);
                }
            } else a = (double)arow[k];
            return fmin(fmax(a, 0.001), 0.999999999999);
        };
        __syncthreads();
        for (int k = threadIdx.x; k <= H; k += blockDim.x) {
            env[k] = (1.0 - mix) * fabs((double)s0[k]) + mix * fabs((double)s1[k]);
            const double a = (1.0 - mix) * a_of(a0, c0, k) + mix * a_of(a1, c1, k);
            rat[k] = a * a;
        }
        const double a00 = (1.0 - mix) * a_of(a0, c0, 0) + mix * a_of(a1, c1, 0);
        const bool periodic = cur_v && !(a00 * a00 > 0.999);
        __syncthreads();
        if (periodic) {
            for (int k = threadIdx.x; k <= H; k += blockDim.x) lg[k] = log(env[k] * (1.0 - rat[k]) + 1e-12) / 2.0;
            min_phase_f64(lg, buf, hs, N, logN, tw);
            const double coef = 2.0 * kPi * (double)xs[pi] / (double)N;
            for (int k = threadIdx.x; k <= H; k += blockDim.x) {
                const double re2 = cos(coef * (double)k), im2 = sqrt(fmax(0.0, 1.0 - re2 * re2));
                const cd h = hs[k];
                hs[k] = {h.x * re2 + h.y * im2, h.y * re2 - h.x * im2};          // h * (re2 - i im2)
            }
            irfft_f64(hs, buf, N, logN, tw);
            double part = 0.0;
            for (int i = threadIdx.x; i < H; i += blockDim.x) part += buf[i].x;     // the causal half: the shifted response's second half
            const double dc = block_sum(part, red);
            for (int i = threadIdx.x; i < N; i += blockDim.x) per[i] = i < H ? -dc * dcr(i) : buf[i - H].x - dc * dcr(i);
        } else {
            for (int i = threadIdx.x; i < N; i += blockDim.x) per[i] = 0.0;
        }
        // aperiodic part: zero-mean burst of WORLD's randn of ns samples, coloured by the minimum-phase envelope
        {
            const long base = (long)id - idx0;
            const int nuse = ns < N ? ns : N;
            double sum = 0.0;
            for (int i = threadIdx.x; i < ns; i += blockDim.x) sum += (base + i < p.table_len) ? (double)p.randn[base + i] : 0.0;
            const double mean = ns > 0 ? block_sum(sum, red) / (double)ns : 0.0;
            __syncthreads();
            for (int i = threadIdx.x; i < N; i += blockDim.x)
                buf[brev(i, logN)] = {(i < nuse && base + i < p.table_len) ? (double)p.randn[base + i] - mean : 0.0, 0.0};
            __syncthreads();
            fft_lds(buf, logN, tw, logN, false);
            for (int k = threadIdx.x; k <= H; k += blockDim.x) nz[k] = buf[k];
        }
        for (int k = threadIdx.x; k <= H; k += blockDim.x) lg[k] = log(cur_v ? env[k] * rat[k] : env[k]) / 2.0;
        min_phase_f64(lg, buf, hs, N, logN, tw);
        for (int k = threadIdx.x; k <= H; k += blockDim.x) {
            const cd h = hs[k], z = nz[k];
            hs[k] = {h.x * z.x - h.y * z.y, h.x * z.y + h.y * z.x};
        }
        irfft_f64(hs, buf, N, logN, tw);
        const double sq = sqrt((double)ns);
        float* out = respb + (size_t)pi * N;
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const double ap_ = buf[i < H ? i + H : i - H].x;                        // fftshift
            out[i] = (float)((per[i] * sq + ap_) / (double)N);
        }
        __syncthreads();
    }
}
#pragma clang fp contract(fast)

// 3. overlap-add: y[n] = sum over pulses with idx in [n - N/2, n + N/2 - 1] of resp[pulse][n - idx + N/2 - 1], pulse order
__global__ __launch_bounds__(256) void world_overlap_add_kernel(WorldParams p) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int T = p.frames ? min(max(p.frames[b], 0), p.T) : p.T;      // an utterance's own frame count, never past the row
    const int ylen = T > 0 ? world_ylen(T, p.frame_period_ms, p.fs) : 0;
    if (n >= p.Ymax) return;
    float* y = p.y + (size_t)b * p.Ymax;
    const int P = p.n_pulses[b];
    if (n >= ylen) { y[n] = 0.f; return; }
    if (P < 0) { y[n] = __builtin_nanf(""); return; }   // pulse list overflowed: fail loudly
    const int* idx = p.idx + (size_t)b * p.Pcap;
    const int nf = p.nf, nh = nf / 2;
    const float* resp = p.resp + (size_t)b * p.Pcap * nf;
    int lo = 0, hi = P;                                  // first pulse with idx >= n - 256
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (idx[mid] < n - nh) lo = mid + 1; else hi = mid;
    }
    float acc = 0.f;
    for (int q = lo; q < P; ++q) {
        const int id = idx[q];
        if (id > n + nh - 1) break;
        acc += resp[(size_t)q * nf + (n - id + nh - 1)];
    }
    y[n] = acc;
}

// decode_aperiodicity: coded [rows][nb] dB -> ap [rows][257]
__global__ void world_decode_ap_kernel(const float* __restrict__ coded, float* __restrict__ ap, long long rows, int nb, int fs, int nbins, int fft_size) {
    const long long row = blockIdx.x;
    if (row >= rows) return;
    const float* c = coded + row * nb;
    float mean = 0.f;
    for (int i = 0; i < nb; ++i) mean += c[i];
    mean /= (float)nb;
    float* o = ap + row * nbins;
    if (mean > -0.5f) {
        for (int k = threadIdx.x; k < nbins; k += blockDim.x) o[k] = 1.0f;               // 1 - 1e-12 rounds to 1 in fp32
        return;
    }
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
        const float f = (float)fs / (float)fft_size * (float)k;
        // coarse grid: 0 Hz (-60 dB), 3 kHz steps (coded values), fs / 2 (-1e-12 dB)
        int seg = (int)(f / 3000.f);
        if (seg > nb) seg = nb;
        const float x0 = 3000.f * seg, x1 = seg == nb ? 0.5f * fs : 3000.f * (seg + 1);
        const float y0 = seg == 0 ? -60.f : c[seg - 1], y1 = seg == nb ? -1e-12f : c[seg];
        const float db = y0 + (f - x0) / (x1 - x0) * (y1 - y0);
        o[k] = exp2f(db * (3.321928094887362f / 20.f));
    }
}
}   // namespace

// WORLD's randn() after randn_reseed(): sum of twelve xorshift128 draws (>> 4, x 2^-28) minus 6.  HOST function on a HOST buffer (the one
// entry point of the library that is not a launch): the sequence is the same for every utterance and call, so callers fill it once.
extern "C" int v100_world_randn_host(float* host_out, long long n) {
    if (!host_out) return V100_ERR_NULL;
    if (n < 0) return V100_ERR_SHAPE;
    uint32_t x = 123456789u, y = 362436069u, z = 521288629u, w = 88675123u;
    for (long long i = 0; i < n; ++i) {
        uint32_t tmp = 0;
        for (int j = 0; j < 12; ++j) {
            const uint32_t t = x ^ (x << 11);
            x = y; y = z; z = w;
            w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
            tmp += w >> 4;
        }
        host_out[i] = (float)(tmp / 268435456.0 - 6.0);
    }
    return V100_OK;
}

extern "C" int v100_world_decode_aperiodicity(const float* coded, float* ap, long long rows, int nb, int fs, int fft_size, void* stream) {
    if (!coded || !ap) return V100_ERR_NULL;
    if (rows <= 0 || fs <= 0 || fft_size < 2) return V100_ERR_SHAPE;
    const double lim = (fs / 2.0 - 3000.0) < 15000.0 ? (fs / 2.0 - 3000.0) : 15000.0;
    if (nb < 1 || nb != (int)(lim / 3000.0) || rows > 0x7fffffffL) return V100_ERR_SHAPE;
    V100_GGL(world_decode_ap_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, coded, ap, rows, nb, fs, fft_size / 2 + 1, fft_size);
    return v100_launch_status();
}

static int world_ymax(int T, int fs, double frame_period_ms) { return (int)((double)T * frame_period_ms * (double)fs / 1000.0); }

// bytes of the `workspace` argument of v100_world_synthesize
#ifdef TBC_DEBUG
extern "C" int v100_tbc_debug_read(long long* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tbc_dbg), sizeof(long long) * 8 * 64) == hipSuccess ? 0 : 2; }
#endif
extern "C" long long v100_world_synth_workspace_bytes(int B, int T, int fs, double frame_period_ms, int fft_size, int max_pulses) {
    if (B <= 0 || T <= 0 || fs <= 0 || fft_size < 64 || fft_size > 2048 || (fft_size & (fft_size - 1)) || max_pulses <= 0 || frame_period_ms <= 0) return -1;
    const long long Y = (world_ymax(T, fs, frame_period_ms) + 63) & ~63LL;
    const long long Pc = (max_pulses + 63) & ~63LL;
    const long long nch = (world_ymax(T, fs, frame_period_ms) + TBC_CH - 1) / TBC_CH + 1;   // time-base chain mailboxes (20 B per chunk of the CHAINED kernel, as v100_world_synthesize lays them out) + tickets
    return (long long)B * (Y + Pc * 4 + Pc * 4 + Pc * fft_size * 4) + 256 + (((long long)B * (4 + nch * 20) + 255) & ~255LL) + 256;
}

extern "C" int v100_world_synthesize(const float* f0, const float* sp, const float* ap, const float* coded_ap, int nb, const int* frames, const float* randn_table,
                                     long long table_len, const float* tw256, const float* tw512, const float* dc_remover, float* y,
                                     int* n_pulses, void* workspace, int B, int T, int fs, double frame_period_ms, int fft_size,
                                     int max_pulses, void* stream) {
    if (!f0 || !sp || (!ap && !coded_ap) || !randn_table || !y || !n_pulses || !workspace) return V100_ERR_NULL;
    if (fft_size == NF && (!tw256 || !tw512 || !dc_remover)) return V100_ERR_NULL;       // the tables of the 512-point fp32 kernel
    if (coded_ap && (nb < 1 || nb > 5)) return V100_ERR_SHAPE;
    if (B <= 0 || T < 2 || fs <= 0 || fft_size < 64 || fft_size > 2048 || (fft_size & (fft_size - 1)) || max_pulses <= 0 || frame_period_ms <= 0) return V100_ERR_SHAPE;
    const int Ymax = world_ymax(T, fs, frame_period_ms);
    if (Ymax < 2 || table_len < Ymax || B > 65535) return V100_ERR_SHAPE;
    // a chunk's contour window must fit the kernel that runs: the chained kernel's (TBC_CH samples in TBC_NFR frames) or, failing
    // that, the serial kernel's (TBS_CH in TBS_NFR); only when neither does is the shape refused
    const double spf = (double)fs * frame_period_ms / 1000.0;
    const bool chain_fits = (double)TBC_CH / spf + 4.0 <= (double)TBC_NFR, serial_fits = (double)TBS_CH / spf + 4.0 <= (double)TBS_NFR;
    if (!chain_fits && !serial_fits) return V100_ERR_SHAPE;
    const long long Y = (Ymax + 63) & ~63LL, Pc = (max_pulses + 63) & ~63LL;
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    WorldParams p{};
    p.f0 = f0; p.sp = sp; p.ap = ap ? ap : sp; p.coded = coded_ap; p.nb = nb; p.frames = frames; p.randn = randn_table; p.table_len = table_len; p.y = y; p.n_pulses = n_pulses;
    p.idx = (int*)w;                    w += (size_t)B * Pc * 4;
    p.xshift = (float*)w;               w += (size_t)B * Pc * 4;
    p.resp = (float*)w;                 w += (size_t)B * Pc * fft_size * 4;
    p.vuv = (unsigned char*)w;          w += (size_t)B * Y;
    w = (char*)(((uintptr_t)w + 255) & ~(uintptr_t)255);
    const int nch = (Ymax + TBC_CH - 1) / TBC_CH + 1;
    char* chain0 = w;
    TbChain ch{};
    ch.nchunks = nch;
    ch.cval = (double*)w;               w += (size_t)B * nch * 8;
    ch.cflag = (int*)w;                 w += (size_t)B * nch * 4;
    ch.nflag = (int*)w;                 w += (size_t)B * nch * 4;
    ch.nval = (int*)w;                  w += (size_t)B * nch * 4;
    ch.ticket = (int*)w;                w += (size_t)B * 4;
    const size_t chain_bytes = (size_t)(w - chain0);
    p.B = B; p.T = T; p.fs = fs; p.Ymax = Ymax; p.Pcap = (int)Pc;
    p.frame_period = frame_period_ms / 1000.0; p.frame_period_ms = frame_period_ms;
    p.tw256 = tw256; p.tw512 = tw512; p.dcr = dc_remover; p.nf = fft_size;
    // (Ymax as the row pitch of total / vuv would leave rows of `total` 8-byte aligned only when Ymax is even: use the padded Y)
    p.Ymax = Ymax;
    hipStream_t st = (hipStream_t)stream;
    WorldParams pt = p;
    pt.Ymax = (int)Y;                   // workspace rows are Y long; y rows are Ymax long (kernels 1 and 2 only touch the workspace)
    static const bool tb_serial_env = []() { const char* e = getenv("V100_WORLD_TB_SERIAL"); return e && e[0] == '1'; }();     // A/B: one workgroup per utterance
    const bool tb_serial = (tb_serial_env && serial_fits) || !chain_fits;
    if (tb_serial) V100_GGL(world_timebase_kernel, dim3((unsigned)B), dim3(256), 0, st, pt);
    else {
        if (hipMemsetAsync(chain0, 0, chain_bytes, st) != hipSuccess) return V100_ERR_LAUNCH;
        V100_GGL(world_timebase_chain_kernel, dim3((unsigned)((Ymax + TBC_CH - 1) / TBC_CH), (unsigned)B), dim3(TBC_NT), 0, st, pt, ch);
    }
    if (fft_size == NF) {
        int gx = (max_pulses + 3) / 4;
        if (gx > 2048) gx = 2048;
        V100_GGL(world_pulse_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, st, pt);
    } else {
        int logN = 0;
        while ((1 << logN) < fft_size) ++logN;
        const int H = fft_size / 2;
        const size_t lds = sizeof(double) * (2 * (size_t)fft_size + 2 * H + 2 * 2 * (H + 2) + 3 * (H + 2) + fft_size);
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void*)world_pulse_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
                return V100_ERR_LAUNCH;
            attr = true;
        }
        int gx = max_pulses < 4096 ? max_pulses : 4096;
        V100_GGL(world_pulse_f64_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), lds, st, pt, logN);
    }
    // overlap-add reads idx / resp (pitch Pcap) and writes y (pitch Ymax)
    V100_GGL(world_overlap_add_kernel, dim3((unsigned)((Ymax + 255) / 256), (unsigned)B), dim3(256), 0, st, p);
    return v100_launch_status();
}
