// WORLD analysis on the device -- pyworld.dio / cheaptrick / d4c / code_aperiodicity as WORLDVocoder.encode calls them
// (voice100/vocoder.py:61-74) -- in fp64, the arithmetic type of the reference (it converts the waveform to double first).
//
// PARITY UNPINNED: pyworld (a Cython wrapper over M. Morise's C++ WORLD) is neither in the reference tree nor in the build image.
// The kernels follow the published algorithms (DIO: Morise et al., AES 35th Conf. 2009; CheapTrick: Morise, Speech Communication 67,
// 2015; D4C: Morise, Speech Communication 84, 2016) in the structure restated by oracle/world_analysis.py, which is their checker.
//
// Work decomposition (this is launch- and latency-bound integer / fp64 work, nowhere near a roofline -- the point is that encode()
// no longer leaves the device or needs pyworld; a 10 s utterance is ~1000 frames):
//   DIO         the two filters (50 Hz low cut, one Nuttall low pass per band) are FIRs of 641 and <= 284 taps: applied in the time
//               domain (the reference multiplies spectra of a 2^18-point FFT: the same linear convolution, its buffer is sized so
//               that nothing wraps); zero-crossing events are compacted in order by one workgroup per (utterance, band, kind);
//               candidates and scores per frame; the contour fix-up is sequential by nature and runs on one lane per utterance.
//   CheapTrick  one workgroup per frame: window, 3 FFTs of fft_size in LDS, smoothing by differences of a cumulative sum, liftering.
//   D4C         one workgroup per voiced frame: "love train" voicing check (1 FFT of 2048), then 5 + bands FFTs of 2048, four
//               smoothings, a bitonic sort of the band's power spectrum.
// WORLD's safeguard noise (randn() * 1e-12 on window samples, |randn()| * eps on CheapTrick's power bins) is drawn from the fixed
// sequence at the offsets at which the sequential C++ would consume it (an exclusive scan over the frames).
#include "common.h"
#include "../../include/voice100_hip.h"
#include <math.h>
#include "world_f64.h"

namespace {
#pragma clang fp contract(off)

constexpr double kPi = 3.1415926535897932384;
constexpr double kSafeMin = 1e-12;
constexpr double kEps = 2.220446049250313e-16;
constexpr double kDioDither = 1e-13;
constexpr double kMaxValue = 100000.0;
constexpr double kDefaultF0 = 500.0;
constexpr double kFloorF0D4C = 47.0;
constexpr double kLoveTrainF0 = 40.0;
constexpr double kFreqInterval = 3000.0;
constexpr int NT = 256;            // threads per workgroup of every kernel here
constexpr int MAXB = 16;           // DIO bands

__host__ __device__ inline int mround(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }
__host__ __device__ inline int dio_frames(int fs, int len, double fp) { return (int)(1000.0 * len / fs / fp) + 1; }

// ---------------------------------------------------------------------------------------------------------------------------------
// DIO
struct DioParams {
    const float* x; const int* lengths; int B, pitch, fs;
    double f0_floor, f0_ceil, frame_period, allowed_range;
    int nbands, hc, zp, npitch, Tmax;
    double boundary[MAXB]; int half[MAXB];
    const double* lowcut; const double* nuttall;
    double* mean; double* z; long long zpitch; double* filt; long long fpitch; double* fine; long long epitch; int* counts;
    double* cand; double* f0; int* negi; int* posi;
};

__global__ __launch_bounds__(1024) void dio_mean_kernel(DioParams p) {
    __shared__ double red[16], redm[16];
    const int b = blockIdx.x, len = p.lengths[b];
    const float* x = p.x + (long long)b * p.pitch;
    double s = 0.0, mx = 0.0;
    for (int i = threadIdx.x; i < len; i += 1024) {
        const double v = (double)x[i];
        s += v;
        mx = fmax(mx, fabs(v));
    }
    s = wave_sum_f64(s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; redm[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, m = 0.0;
        for (int i = 0; i < 16; ++i) { t += red[i]; m = fmax(m, redm[i]); }
        p.mean[b] = t / (double)(len + 1);                      // y has x_length + 1 samples, the last one zero
        p.mean[p.B + b] = m * kDioDither;                       // amplitude of the explicit noise floor (see dio_dither)
    }
}

// The noise floor WORLD gets by accident from its FFT convolution, made explicit and deterministic (oracle/world_analysis.py, DIGITAL
// SILENCE): uniform in [-1, 1) from an integer hash of the sample index (lowbias32)
__device__ inline double dio_dither(unsigned m) {
    unsigned h = m;
    h = (h ^ (h >> 16)) * 0x7FEB352Du;
    h = (h ^ (h >> 15)) * 0x846CA68Bu;
    h = h ^ (h >> 16);
    return (double)h / 2147483648.0 - 1.0;
}

// z = y' (*) low cut, y' = y - mean on [0, y_length), for n in [-zp, y_length + zp)
__global__ __launch_bounds__(NT) void dio_lowcut_kernel(DioParams p) {
    extern __shared__ double sh[];
    const int b = blockIdx.y, len = p.lengths[b], ylen = len + 1, hc = p.hc;
    const int n0 = (int)blockIdx.x * NT - p.zp;
    if (n0 >= ylen + p.zp) return;
    const float* x = p.x + (long long)b * p.pitch;
    const double mean = p.mean[b], amp = p.mean[p.B + b];
    for (int i = threadIdx.x; i < NT + 2 * hc; i += NT) {
        const int m = n0 - hc + i;
        double v = 0.0;
        if (m >= 0 && m < ylen) {
            v = (m < len ? (double)x[m] : 0.0) - mean;
            v = v + dio_dither((unsigned)m) * amp;
        }
        sh[i] = v;
    }
    __syncthreads();
    const int n = n0 + (int)threadIdx.x;
    if (n >= ylen + p.zp) return;
    double acc = 0.0;
    for (int j = -hc; j <= hc; ++j) acc += p.lowcut[j + hc] * sh[(int)threadIdx.x + hc - j];
    p.z[(long long)b * p.zpitch + n + p.zp] = acc;
}

// filtered[n] = sum_k nuttall[k] z[n + 2h - k], n in [0, y_length): the low-passed signal with the window's delay taken out
__global__ __launch_bounds__(NT) void dio_band_kernel(DioParams p) {
    extern __shared__ double sh[];
    const int b = blockIdx.z, band = blockIdx.y, ylen = p.lengths[b] + 1, h = p.half[band], taps = 4 * h;
    const int n0 = (int)blockIdx.x * NT;
    if (n0 >= ylen) return;
    const double* z = p.z + (long long)b * p.zpitch + p.zp;
    // z index range of this tile: [n0 + 2h - (taps - 1), n0 + NT - 1 + 2h]
    const int lo = n0 + 2 * h - (taps - 1), cnt = NT + taps - 1;
    for (int i = threadIdx.x; i < cnt; i += NT) {
        const int m = lo + i;
        sh[i] = (m >= -p.zp && m < ylen + p.zp) ? z[m] : 0.0;
    }
    __syncthreads();
    const int n = n0 + (int)threadIdx.x;
    if (n >= ylen) return;
    const double* w = p.nuttall + (long long)band * p.npitch;
    double acc = 0.0;
    for (int k = 0; k < taps; ++k) acc += w[k] * sh[(int)threadIdx.x + taps - 1 - k];
    p.filt[((long long)b * p.nbands + band) * p.fpitch + n] = acc;
}

// the four kinds of events of one band: negative-going zero crossings of f, -f, d/dt(-f) ... (ZeroCrossingEngine), compacted in order
__global__ __launch_bounds__(NT) void dio_events_kernel(DioParams p) {
    __shared__ int red[16];
    const int kind = blockIdx.x, band = blockIdx.y, b = blockIdx.z, ylen = p.lengths[b] + 1;
    const double* f = p.filt + ((long long)b * p.nbands + band) * p.fpitch;
    double* out = p.fine + (((long long)b * p.nbands + band) * 4 + kind) * p.epitch;
    const int slen = kind < 2 ? ylen : ylen - 1;
    auto s = [&](int i) -> double {
        switch (kind) {
            case 0: return f[i];
            case 1: return -f[i];
            case 2: return f[i + 1] - f[i];                    // (-f[i]) - (-f[i+1]): differentiated after the negation
            default: return f[i] - f[i + 1];
        }
    };
    int written = 0;
    constexpr int PER = 8;
    for (int c0 = 0; c0 < slen - 1; c0 += NT * PER) {
        const int i0 = c0 + (int)threadIdx.x * PER;
        double ev[PER / 2 + 1];
        int n = 0;
        if (i0 < slen - 1) {
            double cur = s(i0);
            for (int j = 0; j < PER; ++j) {
                const int i = i0 + j;
                if (i >= slen - 1) break;
                const double nxt = s(i + 1);
                if (0.0 < cur && nxt <= 0.0) {
                    if (n < PER / 2 + 1) ev[n] = (double)(i + 1) - cur / (nxt - cur);
                    ++n;
                }
                cur = nxt;
            }
        }
        int tot;
        const int off = block_excl_scan<int>(n, red, &tot);
        for (int j = 0; j < n; ++j) out[written + off + j] = ev[j];
        written += tot;
    }
    if (threadIdx.x == 0) p.counts[((long long)b * p.nbands + band) * 4 + kind] = written;
}

// interp1 of the (location, interval) pairs made of consecutive fine edges, WORLD semantics (end segments extrapolate)
__device__ inline double dio_interp(const double* fine, int nint, double xi, double fs) {
    int lo = 0, hi = nint;                                     // first k with loc[k] > xi
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double loc = (fine[mid] + fine[mid + 1]) / 2.0 / fs;
        if (loc <= xi) lo = mid + 1; else hi = mid;
    }
    int k = lo < 1 ? 1 : (lo > nint - 1 ? nint - 1 : lo);
    const double x0 = (fine[k - 1] + fine[k]) / 2.0 / fs, x1 = (fine[k] + fine[k + 1]) / 2.0 / fs;
    const double y0 = fs / (fine[k] - fine[k - 1]), y1 = fs / (fine[k + 1] - fine[k]);
    const double sft = (xi - x0) / (x1 - x0);
    return y0 + sft * (y1 - y0);
}

__global__ __launch_bounds__(NT) void dio_candidates_kernel(DioParams p) {
    const int b = blockIdx.y, t = blockIdx.x * NT + threadIdx.x;
    const int T = dio_frames(p.fs, p.lengths[b], p.frame_period);
    if (t >= T) return;
    const double tpos = (double)t * p.frame_period / 1000.0, fs = (double)p.fs;
    double best_score = 0.0, best = 0.0;
    for (int band = 0; band < p.nbands; ++band) {
        const int* cnt = p.counts + ((long long)b * p.nbands + band) * 4;
        double c = 0.0, sc = kMaxValue;
        const bool ok = cnt[0] - 1 - 2 > 0 && cnt[1] - 1 - 2 > 0 && cnt[2] - 1 - 2 > 0 && cnt[3] - 1 - 2 > 0;
        if (ok) {
            double v[4];
            for (int k = 0; k < 4; ++k)
                v[k] = dio_interp(p.fine + (((long long)b * p.nbands + band) * 4 + k) * p.epitch, cnt[k] - 1, tpos, fs);
            c = (((v[0] + v[1]) + v[2]) + v[3]) / 4.0;
            double q = 0.0;
            for (int k = 0; k < 4; ++k) q += (v[k] - c) * (v[k] - c);
            sc = sqrt(q / 3.0);
            const double bd = p.boundary[band];
            if (c > bd || c < bd / 2.0 || c > p.f0_ceil || c < p.f0_floor) { c = 0.0; sc = kMaxValue; }
        }
        p.cand[((long long)b * p.nbands + band) * p.Tmax + t] = c;
        if (band == 0 || best_score > sc) { best_score = sc; best = c; }
    }
    p.f0[(long long)b * p.Tmax + t] = best;                    // the best contour; dio_fix_kernel rewrites it in place
}

__device__ inline double dio_select_best(double cur, double past, const double* cand, int nb, int Tmax, int target, double allowed) {
    const double ref = (cur * 3.0 - past) / 2.0;
    double best = cand[target], err = fabs(ref - best);
    for (int i = 1; i < nb; ++i) {
        const double c = cand[(long long)i * Tmax + target], e = fabs(ref - c);
        if (e < err) { err = e; best = c; }
    }
    if (fabs(1.0 - best / ref) > allowed) return 0.0;
    return best;
}

// FixF0Contour: jumps out, short voiced sections out, then every section grown forward and backward along the closest candidates
__global__ __launch_bounds__(NT) void dio_fix_kernel(DioParams p, double* s1) {
    const int b = blockIdx.x, T = dio_frames(p.fs, p.lengths[b], p.frame_period);
    double* f0 = p.f0 + (long long)b * p.Tmax;                 // holds the best contour on entry
    double* a = s1 + (long long)b * p.Tmax;
    const double* cand = p.cand + (long long)b * p.nbands * p.Tmax;
    const int vrm = (int)(0.5 + 1000.0 / p.frame_period / p.f0_floor) * 2 + 1;
    if (T <= vrm) {
        for (int i = threadIdx.x; i < T; i += NT) f0[i] = 0.0;
        return;
    }
    auto base = [&](int i) -> double { return (i < vrm || i >= T - vrm) ? 0.0 : f0[i]; };
    for (int i = threadIdx.x; i < T; i += NT) {
        double v = 0.0;
        if (i >= vrm) {
            const double bi = base(i), bp = base(i - 1);
            v = fabs((bi - bp) / (kSafeMin + bi)) < p.allowed_range ? bi : 0.0;
        }
        a[i] = v;
    }
    __syncthreads();
    const int center = (vrm - 1) / 2;
    // steps 2-4 on an LDS copy of the contour when it fits (<= 4096 frames = 41 s at 10 ms): the section growing below is one lane walking
    // the contour, and dependent global loads cost it ~0.3 us a frame
    extern __shared__ double fix_lds[];
    const bool in_lds = p.Tmax <= 4096;             // (the launcher sizes the LDS by Tmax)
    double* w = in_lds ? fix_lds : f0;
    int* neg = in_lds ? (int*)(fix_lds + T) : p.negi + (long long)b * p.Tmax;
    int* pos = in_lds ? neg + T / 2 + 2 : p.posi + (long long)b * p.Tmax;
    // ... and the candidates of every band beside it when they fit too (each frame a section grows by reads one candidate per band)
    const bool cand_lds = in_lds && (long long)p.nbands * p.Tmax * 8 <= 96 * 1024;
    if (cand_lds) {
        double* sc = (double*)(((uintptr_t)(pos + T / 2 + 2) + 7) & ~(uintptr_t)7);
        for (int i = threadIdx.x; i < p.nbands * p.Tmax; i += NT) sc[i] = cand[i];
        cand = sc;
    }
    for (int i = threadIdx.x; i < T; i += NT) {
        double v = a[i];
        if (i >= center && i < T - center)
            for (int j = -center; j <= center; ++j)
                if (a[i + j] == 0.0) { v = 0.0; break; }
        w[i] = v;                                              // step 2 (the best contour is not needed any more)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int nneg = 0, npos = 0;
        for (int i = 1; i < T; ++i) {
            if (w[i] == 0.0 && w[i - 1] != 0.0) neg[nneg++] = i - 1;
            else if (w[i - 1] == 0.0 && w[i] != 0.0) pos[npos++] = i;
        }
        for (int i = 0; i < nneg; ++i) {
            const int limit = i == nneg - 1 ? T - 1 : neg[i + 1];
            for (int j = neg[i]; j < limit; ++j) {
                w[j + 1] = dio_select_best(w[j], w[j - 1], cand, p.nbands, p.Tmax, j + 1, p.allowed_range);
                if (w[j + 1] == 0.0) break;
            }
        }
        for (int i = npos - 1; i >= 0; --i) {
            const int limit = i == 0 ? 1 : pos[i - 1];
            for (int j = pos[i]; j > limit; --j) {
                w[j - 1] = dio_select_best(w[j], w[j + 1], cand, p.nbands, p.Tmax, j - 1, p.allowed_range);
                if (w[j - 1] == 0.0) break;
            }
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += NT) f0[i] = w[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// pieces shared by CheapTrick and D4C (all on LDS arrays, one workgroup per frame)

// real input r[0 .. n) (zero beyond) -> spectrum in a (natural order)
__device__ void fft_real_lds(cd* a, const double* r, int n, double scale_index, int logN, const cd* tw, int logNT) {
    const int N = 1 << logN;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        double v = i < n ? r[i] : 0.0;
        if (scale_index != 0.0) v *= (double)i + 1.0;
        a[brev(i, logN)] = {v, 0.0};
    }
    __syncthreads();
    fft_lds(a, logN, tw, logNT, false);
}

__device__ inline double interp1q(double x0, double dx, const double* y, int n, double xi) {
    const double pos = (xi - x0) / dx;
    int base = (int)pos;
    base = base < 0 ? 0 : (base > n - 2 ? n - 2 : base);
    const double frac = pos - (double)base;
    return y[base] + (y[base + 1] - y[base]) * frac;
}

// DCCorrection (common.cpp): the spectrum below F0 gets its mirror image about F0 added.  In place; scratch: >= 2 + f0 F / fs doubles
__device__ void dc_correction(double* p, double f0, int fs, int F, double* scratch) {
    const int upper = 2 + (int)(f0 * F / fs), rn = upper - 1;
    __syncthreads();
    for (int i = threadIdx.x; i < rn; i += blockDim.x) {
        const double axis = (double)i * fs / F;
        const double xi = (axis - f0) / (-(double)fs / F);
        int base = (int)xi;
        if (base > upper - 1) base = upper - 1;
        const double frac = xi - (double)base;
        scratch[i] = p[base] + (p[base + 1] - p[base]) * frac;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rn; i += blockDim.x) p[i] += scratch[i];
    __syncthreads();
}

// LinearSmoothing (common.cpp): moving average of `width` Hz as differences of the cumulative sum of the mirrored spectrum.
// src[0 .. F/2], dst may be src; seg: >= F/2 + 2 boundary + 1 doubles
__device__ void linear_smoothing(const double* src, double* dst, double* seg, double width, int fs, int F, double* red) {
    const int half = F / 2, boundary = (int)(width * F / fs) + 1, n = half + 2 * boundary + 1;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double v;
        if (i < boundary) v = src[boundary - i];
        else if (i < half + boundary) v = src[i - boundary];
        else v = src[half - (i - (half + boundary))];
        seg[i] = v * fs / F;
    }
    cumsum_lds(seg, n, red);
    const double origin = -((double)boundary - 0.5) * fs / F, dx = (double)fs / F;
    for (int k = threadIdx.x; k <= half; k += blockDim.x) {
        const double freq = (double)k / F * fs;
        const double lo = interp1q(origin, dx, seg, n, freq - width / 2.0);
        const double hi = interp1q(origin, dx, seg, n, freq + width / 2.0);
        dst[k] = (hi - lo) / width;
    }
    __syncthreads();
}

// F0-adaptive windowing (GetWindowedWaveform of cheaptrick.cpp / d4c.cpp): wav[0 .. 2 half] = x w + noise, its weighted mean removed.
// type 0: CheapTrick's Hanning of 3 periods, unit energy; 1: Hanning of `ratio` periods; 2: Blackman of `ratio` periods.
__device__ int windowed_waveform(const float* x, int len, int fs, double f0, double position, int type, double ratio,
                                 const double* rnd, double* wav, double* wbuf, double* red) {
    const int half = type == 0 ? mround(1.5 * fs / f0) : mround(ratio * fs / f0 / 2.0);
    const int nwin = 2 * half + 1, origin = mround(position * fs + 0.001);
    __syncthreads();
    double sq = 0.0;
    for (int i = threadIdx.x; i < nwin; i += blockDim.x) {
        const int base = i - half;
        double w;
        if (type == 0) {
            const double pos = (double)base / 1.5 / fs;
            w = 0.5 * cos(kPi * pos * f0) + 0.5;
        } else {
            const double pos = (2.0 * base / ratio) / fs;
            w = type == 1 ? 0.5 * cos(kPi * pos * f0) + 0.5 : 0.42 + 0.5 * cos(kPi * pos * f0) + 0.08 * cos(kPi * pos * f0 * 2.0);
        }
        wbuf[i] = w;
        sq += w * w;
    }
    if (type == 0) {
        const double nrm = sqrt(block_sum(sq, red));
        for (int i = threadIdx.x; i < nwin; i += blockDim.x) wbuf[i] = wbuf[i] / nrm;
    }
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nwin; i += blockDim.x) {
        int si = origin + i - half;
        si = si < 0 ? 0 : (si > len - 1 ? len - 1 : si);
        const double w = wbuf[i], v = (double)x[si] * w + rnd[i] * kSafeMin;
        wav[i] = v;
        s1 += v;
        s2 += w;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const double coef = s1 / s2;
    for (int i = threadIdx.x; i < nwin; i += blockDim.x) wav[i] -= wbuf[i] * coef;
    __syncthreads();
    return nwin;
}

// offsets into WORLD's randn() sequence at which the frames of one call start consuming it: an exclusive scan over the frames of an
// utterance (one workgroup each).  mode 0: CheapTrick (window + power bins); 1: D4C love train; 2: D4C general body (after ALL the
// love train's draws)
__global__ __launch_bounds__(NT) void world_offsets_kernel(const double* f0, const int* lengths, const double* ap0, long long* off, int Tmax,
                                                           int fs, double frame_period, int mode, double arg, int fft_size) {
    __shared__ long long red[16];
    const int b = blockIdx.x, T = dio_frames(fs, lengths[b], frame_period);
    const double* f = f0 + (long long)b * Tmax;
    long long* o = off + (long long)b * Tmax;
    auto love = [&](double v) -> long long { return v != 0.0 ? 2 * mround(3.0 * fs / (v > kLoveTrainF0 ? v : kLoveTrainF0) / 2.0) + 1 : 0; };
    long long start = 0, tot;
    if (mode == 2) {
        long long s = 0;
        for (int t = threadIdx.x; t < T; t += NT) s += love(f[t]);
        block_excl_scan<long long>(s, red, &start);
    }
    for (int t0 = 0; t0 < T; t0 += NT) {
        const int t = t0 + (int)threadIdx.x;
        long long n = 0;
        if (t < T) {
            const double v = f[t];
            if (mode == 0) {
                const double c = v <= arg ? kDefaultF0 : v;     // arg = CheapTrick's F0 floor
                n = 2 * mround(1.5 * fs / c) + 1 + fft_size / 2 + 1;
            } else if (mode == 1) {
                n = love(v);
            } else if (v != 0.0 && ap0[(long long)b * Tmax + t] > arg) {   // arg = the voicing threshold
                n = 3 * (2 * mround(4.0 * fs / (v > kFloorF0D4C ? v : kFloorF0D4C) / 2.0) + 1);
            }
        }
        const long long e = block_excl_scan<long long>(n, red, &tot);
        if (t < T) o[t] = start + e;
        start += tot;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// CheapTrick
struct CtParams {
    const float* x; const int* lengths; const double* f0; int B, pitch, Tmax, fs, F, logF;
    double frame_period, q1, f0_floor, log_offset;
    const double* rnd; long long rnd_len; const long long* off; const cd* tw;
    double* sp; float* logsp;
};

__global__ __launch_bounds__(NT) void cheaptrick_kernel(CtParams p) {
    extern __shared__ double sh[];
    const int b = blockIdx.y, t = blockIdx.x, F = p.F, half = F / 2, fs = p.fs;
    const int len = p.lengths[b], T = dio_frames(fs, len, p.frame_period);
    if (t >= T) {                            // beyond this utterance's frames (ragged batch): zero rows
        for (int k = threadIdx.x; k <= half; k += blockDim.x) {
            if (p.sp) p.sp[((long long)b * p.Tmax + t) * (half + 1) + k] = 0.0;
            if (p.logsp) p.logsp[((long long)b * p.Tmax + t) * (half + 1) + k] = 0.f;
        }
        return;
    }
    cd* buf = (cd*)sh;                       // F complex
    double* pw = sh + 2 * F;                 // half + 1 (+ pad)
    double* wbuf = pw + half + 8;            // F
    double* wav = wbuf + F;                  // F
    double* seg = wav + F;                   // half + 2 * boundary + 1
    __shared__ double red[16];
    const float* x = p.x + (long long)b * p.pitch;
    const double f = p.f0[(long long)b * p.Tmax + t], cf0 = f <= p.f0_floor ? kDefaultF0 : f;
    const long long off = p.off[(long long)b * p.Tmax + t];
    double* spo = p.sp ? p.sp + ((long long)b * p.Tmax + t) * (half + 1) : nullptr;
    float* lso = p.logsp ? p.logsp + ((long long)b * p.Tmax + t) * (half + 1) : nullptr;
    const int nwin_need = 2 * mround(1.5 * fs / cf0) + 1;
    const int bnd = (int)(cf0 * 2.0 / 3.0 * F / fs) + 1;                 // the smoothing's mirror margin must fit the spectrum and `seg`
    if (off + nwin_need + half + 1 > p.rnd_len || nwin_need > F || 2 * bnd > F || 2 + (int)(cf0 * F / fs) > half || !(cf0 > 0.0)) {       // fail loudly: NaN rows
        for (int k = threadIdx.x; k <= half; k += blockDim.x) {
            if (spo) spo[k] = __builtin_nan("");
            if (lso) lso[k] = __builtin_nanf("");
        }
        return;
    }
    const double position = (double)t * p.frame_period / 1000.0;
    const int nwin = windowed_waveform(x, len, fs, cf0, position, 0, 1.5, p.rnd + off, wav, wbuf, red);
    fft_real_lds(buf, wav, nwin, 0.0, p.logF, p.tw, p.logF);
    for (int k = threadIdx.x; k <= half; k += blockDim.x) pw[k] = buf[k].x * buf[k].x + buf[k].y * buf[k].y;
    dc_correction(pw, cf0, fs, F, seg);
    linear_smoothing(pw, pw, seg, cf0 * 2.0 / 3.0, fs, F, red);
    const double* noise = p.rnd + off + nwin;
    for (int k = threadIdx.x; k <= half; k += blockDim.x) pw[k] = log(pw[k] + fabs(noise[k]) * kEps);
    __syncthreads();
    // cepstrum of the (even) log spectrum, liftered, back: SmoothingWithRecovery
    for (int i = threadIdx.x; i < F; i += blockDim.x) buf[brev(i, p.logF)] = {pw[i <= half ? i : F - i], 0.0};
    __syncthreads();
    fft_lds(buf, p.logF, p.tw, p.logF, false);
    for (int k = threadIdx.x; k <= half; k += blockDim.x) {
        const double quef = (double)k / (double)fs;
        const double sm = k == 0 ? 1.0 : sin(kPi * cf0 * quef) / (kPi * cf0 * quef);
        const double comp = (1.0 - 2.0 * p.q1) + 2.0 * p.q1 * cos(2.0 * kPi * quef * cf0);
        wbuf[k] = buf[k].x * sm * comp;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F; i += blockDim.x) buf[brev(i, p.logF)] = {wbuf[i <= half ? i : F - i], 0.0};
    __syncthreads();
    fft_lds(buf, p.logF, p.tw, p.logF, true);
    for (int k = threadIdx.x; k <= half; k += blockDim.x) {
        const double v = exp(buf[k].x / (double)F);
        if (spo) spo[k] = v;
        if (lso) lso[k] = (float)log(v + p.log_offset);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// D4C
struct D4cParams {
    const float* x; const int* lengths; const double* f0; int B, pitch, Tmax, fs, F, F2, logF2, nb, wlen;
    double frame_period, threshold;
    const double* rnd; long long rnd_len; const long long* off_lt; const long long* off_gb; const cd* tw; const double* nuttall;
    double* ap0; double* ap; double* coded; float* coded32;
};

constexpr int D4C_SEG = 1880;        // doubles: the longest 4-period window (1877 samples at 47 Hz, 22.05 kHz) and a smoothing's mirrored cumulative sum share it
constexpr int DG = 512;              // threads of the D4C kernels: with <= 80 KB of LDS two of their workgroups share a CU

// D4CLoveTrain: the share of the power below 4 kHz in the power below 7.9 kHz (both above 100 Hz) of a 3-period Blackman window
__global__ __launch_bounds__(DG) void d4c_lovetrain_kernel(D4cParams p) {
    extern __shared__ double sh[];
    __shared__ double red[16];
    const int b = blockIdx.y, t = blockIdx.x, fs = p.fs, F2 = p.F2;
    const int len = p.lengths[b], T = dio_frames(fs, len, p.frame_period);
    if (t >= T) return;
    const double f = p.f0[(long long)b * p.Tmax + t];
    double* out = p.ap0 + (long long)b * p.Tmax + t;
    if (f == 0.0) { if (threadIdx.x == 0) *out = 0.0; return; }
    cd* buf = (cd*)sh;
    double* wav = sh + 2 * F2;
    double* wbuf = wav + F2;
    double* pw = wbuf + F2;
    const double cf0 = f > kLoveTrainF0 ? f : kLoveTrainF0;
    const long long off = p.off_lt[(long long)b * p.Tmax + t];
    const int need = 2 * mround(3.0 * fs / cf0 / 2.0) + 1;
    if (off + need > p.rnd_len || need > F2) { if (threadIdx.x == 0) *out = __builtin_nan(""); return; }
    const float* x = p.x + (long long)b * p.pitch;
    const double position = (double)t * p.frame_period / 1000.0;
    const int nwin = windowed_waveform(x, len, fs, cf0, position, 2, 3.0, p.rnd + off, wav, wbuf, red);
    fft_real_lds(buf, wav, nwin, 0.0, p.logF2, p.tw, p.logF2);
    const int b0 = (int)ceil(100.0 * F2 / fs), b1 = (int)ceil(4000.0 * F2 / fs), b2 = (int)ceil(7900.0 * F2 / fs);
    for (int k = threadIdx.x; k <= F2 / 2; k += blockDim.x) pw[k] = k <= b0 ? 0.0 : buf[k].x * buf[k].x + buf[k].y * buf[k].y;
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k <= b2; k += blockDim.x) {
        s2 += pw[k];
        if (k <= b1) s1 += pw[k];
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) *out = s1 / s2;
}

// ascending bitonic sort of a[0 .. n) in LDS, n = E * blockDim.x: a thread keeps elements tid + e * blockDim.x in registers; partner distances
// below 64 are lane exchanges inside a wave (45 of the 55 stages at n = 1024), distances of blockDim.x and more stay inside the thread, the
// others go through LDS
template <int E>
__device__ void bitonic_sort(double* a, int logn) {
    const int n = 1 << logn, nt = blockDim.x, tid = threadIdx.x;
    __syncthreads();
    double v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = a[tid + e * nt];
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            double o[E];
            if (j >= nt) {                                     // partner in this thread's own registers
                const int de = j / nt;
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = v[e ^ de];
            } else if (j >= 64) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) a[tid + e * nt] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = a[(tid ^ j) + e * nt];
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = __shfl_xor(v[e], j, 64);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = tid + e * nt;
                const bool lower = (i & j) == 0, up = (i & k) == 0;
                const double lo = v[e] < o[e] ? v[e] : o[e], hi = v[e] < o[e] ? o[e] : v[e];
                v[e] = (lower == up) ? lo : hi;
            }
        }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) a[tid + e * nt] = v[e];
    __syncthreads();
}

__global__ __launch_bounds__(DG) void d4c_general_kernel(D4cParams p) {
    extern __shared__ double sh[];
    __shared__ double red[16];
    __shared__ double coarse[8];
    const int b = blockIdx.y, t = blockIdx.x, fs = p.fs, F2 = p.F2, h2 = F2 / 2, half = p.F / 2, nb = p.nb;
    const int len = p.lengths[b], T = dio_frames(fs, len, p.frame_period);
    const long long row = (long long)b * p.Tmax + t;
    double* apo = p.ap ? p.ap + row * (half + 1) : nullptr;
    if (t >= T) {                             // beyond this utterance's frames (ragged batch): zero rows
        for (int k = threadIdx.x; k <= half; k += blockDim.x) if (apo) apo[k] = 0.0;
        if (threadIdx.x < nb) {
            if (p.coded) p.coded[row * nb + threadIdx.x] = 0.0;
            if (p.coded32) p.coded32[row * nb + threadIdx.x] = 0.f;
        }
        return;
    }
    const double f = p.f0[row];
    const bool voiced = f != 0.0 && p.ap0[row] > p.threshold;       // a NaN love-train value (table too short) stays unvoiced-coded as NaN below
    const bool bad = f != 0.0 && !(p.ap0[row] == p.ap0[row]);
    const double unv = 1.0 - kSafeMin;
    if (!voiced || bad) {
        const double v = bad ? __builtin_nan("") : unv;
        for (int k = threadIdx.x; k <= half; k += blockDim.x) if (apo) apo[k] = v;
        if (threadIdx.x < nb) {
            const double c = 20.0 * log10(v);
            if (p.coded) p.coded[row * nb + threadIdx.x] = c;
            if (p.coded32) p.coded32[row * nb + threadIdx.x] = (float)c;
        }
        return;
    }
    // 80.6 KB, so that TWO workgroups share a CU (every phase here is a short latency-bound step): buffers whose lives do not overlap share
    // storage -- the windowed waveform and the smoothings' cumulative sums (a waveform is dead once its last FFT has read it), the window
    // values and the first spectrum of a centroid pair, the imaginary half of that spectrum and the group delay
    cd* buf = (cd*)sh;                        // F2 complex
    double* seg = sh + 2 * F2;                // D4C_SEG
    double* wav = seg;
    double* tmpr = seg + D4C_SEG;             // h2 + 1   } the two together hold the window (<= D4C_SEG values) while a waveform is built
    double* tmpi = tmpr + h2 + 1;             // h2 + 1   }
    double* cen = tmpi + h2 + 1;              // h2 + 1
    double* pw = cen + h2 + 1;                // h2 + 1
    double* gd = tmpi;
    double* wbuf = tmpr;
    const double cf0 = f > kFloorF0D4C ? f : kFloorF0D4C;
    const int nw4 = 2 * mround(4.0 * fs / cf0 / 2.0) + 1;
    const long long off = p.off_gb[row];
    const int bmax = (int)(cf0 * F2 / fs) + 1;
    if (off + 3LL * nw4 > p.rnd_len || nw4 > D4C_SEG || nw4 > 2 * (h2 + 1) || h2 + 2 * bmax + 1 > D4C_SEG) {
        for (int k = threadIdx.x; k <= half; k += blockDim.x) if (apo) apo[k] = __builtin_nan("");
        if (threadIdx.x < nb) {
            if (p.coded) p.coded[row * nb + threadIdx.x] = __builtin_nan("");
            if (p.coded32) p.coded32[row * nb + threadIdx.x] = __builtin_nanf("");
        }
        return;
    }
    const float* x = p.x + (long long)b * p.pitch;
    const double position = (double)t * p.frame_period / 1000.0;
    // static centroid: two Blackman windows of four periods, a quarter period either side
    for (int k = threadIdx.x; k <= h2; k += blockDim.x) cen[k] = 0.0;
    for (int side = 0; side < 2; ++side) {
        const double pos = side == 0 ? position - 0.25 / cf0 : position + 0.25 / cf0;
        const int nwin = windowed_waveform(x, len, fs, cf0, pos, 2, 4.0, p.rnd + off + (long long)side * nw4, wav, wbuf, red);
        double pwr = 0.0;
        for (int i = threadIdx.x; i < nwin; i += blockDim.x) pwr += wav[i] * wav[i];
        pwr = sqrt(block_sum(pwr, red));
        for (int i = threadIdx.x; i < nwin; i += blockDim.x) wav[i] = wav[i] / pwr;
        fft_real_lds(buf, wav, nwin, 0.0, p.logF2, p.tw, p.logF2);
        for (int k = threadIdx.x; k <= h2; k += blockDim.x) { tmpr[k] = buf[k].x; tmpi[k] = buf[k].y; }
        fft_real_lds(buf, wav, nwin, 1.0, p.logF2, p.tw, p.logF2);
        for (int k = threadIdx.x; k <= h2; k += blockDim.x) cen[k] += buf[k].x * tmpr[k] + tmpi[k] * buf[k].y;
    }
    dc_correction(cen, cf0, fs, F2, seg);
    // smoothed power spectrum: a Hanning window of four periods
    {
        const int nwin = windowed_waveform(x, len, fs, cf0, position, 1, 4.0, p.rnd + off + 2LL * nw4, wav, wbuf, red);
        fft_real_lds(buf, wav, nwin, 0.0, p.logF2, p.tw, p.logF2);
        for (int k = threadIdx.x; k <= h2; k += blockDim.x) pw[k] = buf[k].x * buf[k].x + buf[k].y * buf[k].y;
        dc_correction(pw, cf0, fs, F2, seg);
        linear_smoothing(pw, pw, seg, cf0, fs, F2, red);
    }
    // static group delay, its smooth part taken out
    for (int k = threadIdx.x; k <= h2; k += blockDim.x) gd[k] = cen[k] / pw[k];
    linear_smoothing(gd, gd, seg, cf0 / 2.0, fs, F2, red);
    linear_smoothing(gd, tmpr, seg, cf0, fs, F2, red);
    for (int k = threadIdx.x; k <= h2; k += blockDim.x) gd[k] -= tmpr[k];
    __syncthreads();
    // band aperiodicity: the tail of the sorted power spectrum of the windowed group delay
    const int wlen = p.wlen, hw = wlen / 2, boundary = mround(F2 * 8.0 / wlen);
    for (int i = 0; i < nb; ++i) {
        const int center = (int)(kFreqInterval * (i + 1) * F2 / fs);
        __syncthreads();
        for (int j = threadIdx.x; j < F2; j += blockDim.x) {
            const double v = j < wlen ? gd[center - hw + j] * p.nuttall[j] : 0.0;
            buf[brev(j, p.logF2)] = {v, 0.0};
        }
        __syncthreads();
        fft_lds(buf, p.logF2, p.tw, p.logF2, false);
        // power spectrum: bins 0 .. h2-1 are sorted (a power of two), the Nyquist bin is ranked into them
        const double extra = buf[h2].x * buf[h2].x + buf[h2].y * buf[h2].y;
        for (int k = threadIdx.x; k < h2; k += blockDim.x) tmpr[k] = buf[k].x * buf[k].x + buf[k].y * buf[k].y;
        bitonic_sort<2>(tmpr, p.logF2 - 1);
        // sorted cumulative sum at index h2 - boundary - 1 = the m = h2 - boundary smallest of the h2 + 1 values
        const int m = h2 - boundary;
        double below = 0.0;
        for (int k = threadIdx.x; k < h2; k += blockDim.x) below += tmpr[k] < extra ? 1.0 : 0.0;
        const int rank = (int)block_sum(below, red);
        const int take = rank < m ? m - 1 : m;
        double at = 0.0, acc = 0.0;
        for (int k = threadIdx.x; k < h2; k += blockDim.x) {
            acc += tmpr[k];
            if (k < take) at += tmpr[k];
        }
        at = block_sum(at, red);
        acc = block_sum(acc, red);
        if (threadIdx.x == 0) {
            if (rank < m) at += extra;
            acc += extra;
            double c = 10.0 * log10(at / acc);
            c = c + (cf0 - 100.0) / 50.0;
            coarse[i] = c < 0.0 ? c : 0.0;
        }
        __syncthreads();
    }
    // GetAperiodicity: the coarse values between -60 dB at 0 Hz and -1e-12 dB at fs / 2, linear in dB over frequency
    auto axis = [&](int j) -> double { return j <= nb ? (double)j * kFreqInterval : fs / 2.0; };
    auto val = [&](int j) -> double { return j == 0 ? -60.0 : (j <= nb ? coarse[j - 1] : -kSafeMin); };
    for (int k = threadIdx.x; k <= half; k += blockDim.x) {
        const double xi = (double)k * fs / p.F;
        int j = 0;
        while (j < nb + 2 && axis(j) <= xi) ++j;                 // first grid point beyond xi
        j = j < 1 ? 1 : (j > nb + 1 ? nb + 1 : j);
        const double s = (xi - axis(j - 1)) / (axis(j) - axis(j - 1));
        const double v = pow(10.0, (val(j - 1) + s * (val(j) - val(j - 1))) / 20.0);
        pw[k] = v;
        if (apo) apo[k] = v;
    }
    __syncthreads();
    // CodeAperiodicity: 20 log10 of it sampled every 3 kHz
    if (threadIdx.x < nb) {
        const double xi = kFreqInterval * ((double)threadIdx.x + 1.0);
        int k = 0;
        while (k <= half && (double)k * fs / p.F <= xi) ++k;
        k = k < 1 ? 1 : (k > half ? half : k);
        const double x0 = (double)(k - 1) * fs / p.F, x1 = (double)k * fs / p.F;
        const double y0 = 20.0 * log10(pw[k - 1]), y1 = 20.0 * log10(pw[k]);
        const double c = y0 + (xi - x0) / (x1 - x0) * (y1 - y0);
        if (p.coded) p.coded[row * nb + threadIdx.x] = c;
        if (p.coded32) p.coded32[row * nb + threadIdx.x] = (float)c;
    }
}

inline int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
inline long long al(long long v) { return (v + 255) & ~255LL; }

struct DioPlan {
    int nbands, hc, zp, npitch, Tmax, ylen;
    double boundary[MAXB]; int half[MAXB];
    long long o_mean, o_z, zpitch, o_filt, fpitch, o_fine, epitch, o_counts, o_cand, o_s1, o_neg, o_pos, total;
};
bool dio_plan(int B, int max_len, int fs, double f0_floor, double f0_ceil, double channels, double frame_period, DioPlan* pl) {
    if (B <= 0 || max_len <= 0 || fs <= 0 || !(f0_floor > 0) || !(f0_ceil > f0_floor) || !(channels > 0) || !(frame_period > 0)) return false;
    const int nb = 1 + (int)(log(f0_ceil / f0_floor) / 0.69314718055994529 * channels);
    if (nb < 1 || nb > MAXB) return false;
    pl->nbands = nb;
    int hmax = 0;
    for (int i = 0; i < nb; ++i) {
        pl->boundary[i] = f0_floor * pow(2.0, (i + 1) / channels);
        pl->half[i] = mround(fs / pl->boundary[i] / 2.0);
        if (pl->half[i] < 1) return false;
        if (pl->half[i] > hmax) hmax = pl->half[i];
    }
    pl->hc = mround(fs / 50.0);
    pl->zp = 2 * hmax;
    pl->npitch = 4 * hmax;
    pl->ylen = max_len + 1;
    pl->Tmax = dio_frames(fs, max_len, frame_period);
    long long o = 0;
    pl->o_mean = o; o += al(8LL * 2 * B);              // means, then dither amplitudes
    pl->zpitch = (pl->ylen + 2LL * pl->zp + 7) & ~7LL;
    pl->o_z = o; o += al(8LL * B * pl->zpitch);
    pl->fpitch = (pl->ylen + 7LL) & ~7LL;
    pl->o_filt = o; o += al(8LL * B * nb * pl->fpitch);
    pl->epitch = (pl->ylen / 2 + 8LL) & ~7LL;
    pl->o_fine = o; o += al(8LL * B * nb * 4 * pl->epitch);
    pl->o_counts = o; o += al(4LL * B * nb * 4);
    pl->o_cand = o; o += al(8LL * B * nb * pl->Tmax);
    pl->o_s1 = o; o += al(8LL * B * pl->Tmax);
    pl->o_neg = o; o += al(4LL * B * pl->Tmax);
    pl->o_pos = o; o += al(4LL * B * pl->Tmax);
    pl->total = o;
    return true;
}
}   // namespace

// WORLD's randn() after randn_reseed(), in double (the analysis adds it scaled by 1e-12 / eps): HOST function on a HOST buffer
extern "C" int v100_world_randn_host_f64(double* host_out, long long n) {
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
        host_out[i] = tmp / 268435456.0 - 6.0;
    }
    return V100_OK;
}

extern "C" int v100_world_frames(int fs, int length, double frame_period_ms) {
    if (fs <= 0 || length < 0 || !(frame_period_ms > 0)) return -1;
    return dio_frames(fs, length, frame_period_ms);
}

extern "C" int v100_world_dio_bands(int fs, double f0_floor, double f0_ceil, double channels_in_octave, int* half_lengths, int max_bands) {
    DioPlan pl;
    if (!dio_plan(1, 1, fs, f0_floor, f0_ceil, channels_in_octave, 1.0, &pl) || pl.nbands > max_bands) return -1;
    if (half_lengths) for (int i = 0; i < pl.nbands; ++i) half_lengths[i] = pl.half[i];
    return pl.nbands;
}

extern "C" long long v100_world_dio_workspace_bytes(int B, int max_len, int fs, double f0_floor, double f0_ceil, double channels_in_octave,
                                                    double frame_period_ms) {
    DioPlan pl;
    if (!dio_plan(B, max_len, fs, f0_floor, f0_ceil, channels_in_octave, frame_period_ms, &pl)) return -1;
    return pl.total;
}

extern "C" int v100_world_dio(const float* x, const int* lengths, int B, int max_len, int pitch, int fs, double f0_floor, double f0_ceil,
                              double channels_in_octave, double frame_period_ms, double allowed_range, const double* lowcut,
                              const double* nuttall, double* f0, void* workspace, void* stream) {
    if (!x || !lengths || !lowcut || !nuttall || !f0 || !workspace) return V100_ERR_NULL;
    DioPlan pl;
    if (pitch < max_len || B > 65535 || !dio_plan(B, max_len, fs, f0_floor, f0_ceil, channels_in_octave, frame_period_ms, &pl)) return V100_ERR_SHAPE;
    char* ws = (char*)workspace;
    DioParams p{};
    p.x = x; p.lengths = lengths; p.B = B; p.pitch = pitch; p.fs = fs;
    p.f0_floor = f0_floor; p.f0_ceil = f0_ceil; p.frame_period = frame_period_ms; p.allowed_range = allowed_range;
    p.nbands = pl.nbands; p.hc = pl.hc; p.zp = pl.zp; p.npitch = pl.npitch; p.Tmax = pl.Tmax;
    for (int i = 0; i < pl.nbands; ++i) { p.boundary[i] = pl.boundary[i]; p.half[i] = pl.half[i]; }
    p.lowcut = lowcut; p.nuttall = nuttall;
    p.mean = (double*)(ws + pl.o_mean); p.z = (double*)(ws + pl.o_z); p.zpitch = pl.zpitch;
    p.filt = (double*)(ws + pl.o_filt); p.fpitch = pl.fpitch; p.fine = (double*)(ws + pl.o_fine); p.epitch = pl.epitch;
    p.counts = (int*)(ws + pl.o_counts); p.cand = (double*)(ws + pl.o_cand); p.f0 = f0;
    p.negi = (int*)(ws + pl.o_neg); p.posi = (int*)(ws + pl.o_pos);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(f0, 0, sizeof(double) * (size_t)B * pl.Tmax, st) != hipSuccess) return V100_ERR_LAUNCH;
    V100_GGL(dio_mean_kernel, dim3(B), dim3(1024), 0, st, p);
    V100_GGL(dio_lowcut_kernel, dim3((pl.ylen + 2 * pl.zp + NT - 1) / NT, B), dim3(NT), sizeof(double) * (NT + 2 * pl.hc), st, p);
    V100_GGL(dio_band_kernel, dim3((pl.ylen + NT - 1) / NT, pl.nbands, B), dim3(NT), sizeof(double) * (NT + pl.npitch), st, p);
    V100_GGL(dio_events_kernel, dim3(4, pl.nbands, B), dim3(NT), 0, st, p);
    V100_GGL(dio_candidates_kernel, dim3((pl.Tmax + NT - 1) / NT, B), dim3(NT), 0, st, p);
    // LDS: the contour + the two section lists when an utterance has <= 4096 frames (else the kernel works in global memory)
    const int tl = pl.Tmax <= 4096 ? pl.Tmax : 0;
    size_t fix_lds = tl ? sizeof(double) * tl + sizeof(int) * (2 * (tl / 2 + 2)) + 32 : 0;
    if (tl && (long long)pl.nbands * pl.Tmax * 8 <= 96 * 1024) fix_lds += sizeof(double) * (size_t)pl.nbands * pl.Tmax;
    static bool fix_attr = false;
    if (!fix_attr) {
        if (hipFuncSetAttribute((const void*)dio_fix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) return V100_ERR_LAUNCH;
        fix_attr = true;
    }
    V100_GGL(dio_fix_kernel, dim3(B), dim3(NT), fix_lds, st, p, (double*)(ws + pl.o_s1));
    return v100_launch_status();
}

extern "C" long long v100_world_randn_bound(int kind, int T, int fs, int fft_size) {
    if (T < 0 || fs <= 0) return -1;
    if (kind == 0) return (long long)T * (fft_size + fft_size / 2 + 1);
    return (long long)T * ((2LL * mround(1.5 * fs / kLoveTrainF0) + 1) + 3 * (2LL * mround(2.0 * fs / kFloorF0D4C) + 1));
}

extern "C" int v100_world_cheaptrick(const float* x, const int* lengths, const double* f0, int B, int max_len, int pitch, int fs,
                                     double frame_period_ms, double q1, int fft_size, const double* randn_table, long long table_len,
                                     const double* twiddle, double* sp, float* logsp, double log_offset, long long* offsets, void* stream) {
    if (!x || !lengths || !f0 || !randn_table || !twiddle || !offsets || (!sp && !logsp)) return V100_ERR_NULL;
    if (B <= 0 || B > 65535 || max_len <= 0 || pitch < max_len || fs <= 0 || !(frame_period_ms > 0) || (fft_size != 512 && fft_size != 1024 && fft_size != 2048))
        return V100_ERR_SHAPE;
    const int Tmax = dio_frames(fs, max_len, frame_period_ms);
    CtParams p{};
    p.x = x; p.lengths = lengths; p.f0 = f0; p.B = B; p.pitch = pitch; p.Tmax = Tmax; p.fs = fs; p.F = fft_size; p.logF = ilog2(fft_size);
    p.frame_period = frame_period_ms; p.q1 = q1; p.f0_floor = 3.0 * fs / (fft_size - 3.0); p.log_offset = log_offset;
    p.rnd = randn_table; p.rnd_len = table_len; p.off = offsets; p.tw = (const cd*)twiddle; p.sp = sp; p.logsp = logsp;
    hipStream_t st = (hipStream_t)stream;
    V100_GGL(world_offsets_kernel, dim3(B), dim3(NT), 0, st, f0, lengths, (const double*)nullptr, offsets, Tmax, fs, frame_period_ms, 0, p.f0_floor,
             fft_size);
    // LDS: F complex + (F/2 + 8) + F + F + (F/2 + 2 * boundary + 1), boundary <= F/2
    const size_t lds = sizeof(double) * (2 * fft_size + fft_size / 2 + 8 + 2 * fft_size + fft_size / 2 + fft_size + 8);
    static bool attr[3] = {false, false, false};
    const int ai = fft_size == 512 ? 0 : (fft_size == 1024 ? 1 : 2);
    if (!attr[ai]) {
        if (hipFuncSetAttribute((const void*)cheaptrick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
            return V100_ERR_LAUNCH;
        attr[ai] = true;
    }
    V100_GGL(cheaptrick_kernel, dim3(Tmax, B), dim3(NT), lds, st, p);
    return v100_launch_status();
}

extern "C" long long v100_world_d4c_workspace_bytes(int B, int max_len, int fs, double frame_period_ms) {
    if (B <= 0 || max_len <= 0 || fs <= 0 || !(frame_period_ms > 0)) return -1;
    const long long Tmax = dio_frames(fs, max_len, frame_period_ms);
    return 3 * al(8LL * B * Tmax);
}

extern "C" int v100_world_d4c(const float* x, const int* lengths, const double* f0, int B, int max_len, int pitch, int fs, double frame_period_ms,
                              double threshold, int fft_size, const double* randn_table, long long table_len, const double* twiddle,
                              const double* nuttall, int window_length, double* ap, double* coded, float* coded32, void* workspace, void* stream) {
    if (!x || !lengths || !f0 || !randn_table || !twiddle || !nuttall || !workspace || (!ap && !coded && !coded32)) return V100_ERR_NULL;
    if (B <= 0 || B > 65535 || max_len <= 0 || pitch < max_len || fs <= 0 || !(frame_period_ms > 0) || fft_size < 4 || (fft_size & (fft_size - 1))) return V100_ERR_SHAPE;
    const int F2 = (int)pow(2.0, 1.0 + (int)(log(4.0 * fs / kFloorF0D4C + 1) / 0.69314718055994529));
    const int FL = (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / kLoveTrainF0 + 1) / 0.69314718055994529));
    const double lim = (fs / 2.0 - kFreqInterval) < 15000.0 ? (fs / 2.0 - kFreqInterval) : 15000.0;
    const int nb = (int)(lim / kFreqInterval);
    const int wlen = (int)(kFreqInterval * F2 / fs) * 2 + 1;
    if (F2 != 2048 || FL != 2048 || nb < 1 || nb > 8 || wlen != window_length || wlen > F2) return V100_ERR_SHAPE;   // 16 kHz / 22.05 kHz
    if ((int)(kFreqInterval * nb * F2 / fs) + wlen / 2 > F2 / 2 || (int)ceil(7900.0 * F2 / fs) > F2 / 2) return V100_ERR_SHAPE;
    const int Tmax = dio_frames(fs, max_len, frame_period_ms);
    char* ws = (char*)workspace;
    const long long seg = al(8LL * B * Tmax);
    D4cParams p{};
    p.x = x; p.lengths = lengths; p.f0 = f0; p.B = B; p.pitch = pitch; p.Tmax = Tmax; p.fs = fs; p.F = fft_size; p.F2 = F2; p.logF2 = ilog2(F2);
    p.nb = nb; p.wlen = wlen; p.frame_period = frame_period_ms; p.threshold = threshold;
    p.rnd = randn_table; p.rnd_len = table_len; p.tw = (const cd*)twiddle; p.nuttall = nuttall;
    p.off_lt = (const long long*)ws; p.off_gb = (const long long*)(ws + seg); p.ap0 = (double*)(ws + 2 * seg);
    p.ap = ap; p.coded = coded; p.coded32 = coded32;
    hipStream_t st = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)d4c_general_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess ||
            hipFuncSetAttribute((const void*)d4c_lovetrain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
            return V100_ERR_LAUNCH;
        attr = true;
    }
    V100_GGL(world_offsets_kernel, dim3(B), dim3(NT), 0, st, f0, lengths, (const double*)nullptr, (long long*)p.off_lt, Tmax, fs, frame_period_ms, 1,
             0.0, fft_size);
    V100_GGL(d4c_lovetrain_kernel, dim3(Tmax, B), dim3(DG), sizeof(double) * (2 * F2 + F2 + F2 + F2 / 2 + 8), st, p);
    V100_GGL(world_offsets_kernel, dim3(B), dim3(NT), 0, st, f0, lengths, (const double*)p.ap0, (long long*)p.off_gb, Tmax, fs, frame_period_ms, 2,
             threshold, fft_size);
    V100_GGL(d4c_general_kernel, dim3(Tmax, B), dim3(DG), sizeof(double) * (2 * F2 + D4C_SEG + 4 * (F2 / 2 + 1)), st, p);
    return v100_launch_status();
}
