// K7 as ONE kernel: waveform -> log-mel, the arithmetic of MelSpectrogramAudioTransform (voice100/data_modules.py:276-291:
// torchaudio MelSpectrogram(16 kHz, n_fft 512, win 400, hop 160, 64 mels, center / reflect, periodic Hann, power 2, HTK, no norm)
// then log(mel.T + 1e-6)).
//
// Round 2 ran it as five launches (framing copy, real-DFT GEMM on the exact-fp32 MFMA, power, filterbank GEMM, log + transpose)
// with a 2.5x framed copy and two intermediate tensors in HBM: 0.33 ms for 256 one-second chunks, ~27 % of the fp32-MFMA floor of
// the dense DFT (13.6 GFLOP for what an FFT does in 0.6).  Here one WAVE owns one frame:
//   load the 400 windowed samples straight from the waveform (reflect padding by index arithmetic; every sample is re-read by
//   3.2 overlapping frames out of L1 / L2) as 256 complex points z[n] = x[2n] + i x[2n+1]  ->  256-point radix-4 DIF FFT, four
//   stages of one butterfly per lane, exchanged through 2 KB of LDS (wave-local: no barrier)  ->  the real-input split
//   X[k] = (Z[k] + Z*[256-k]) / 2 - i W512^k (Z[k] - Z*[256-k]) / 2  ->  |X[k]|^2 into LDS  ->  lane m sums its triangular
//   filter's bins (a filter is a contiguous run of <= 32 bins)  ->  log(mel + offset)  ->  64 lanes store one 256-byte row of
//   out[b][t][:].
// The waveform is read once from HBM, the output written once; no frames / spec / power tensors.  fp32 throughout, twiddles and
// window from tables computed in double by the caller; error against a float64 FFT ~1e-6 of the frame's largest bin.
#include "common.h"
#include "../../include/voice100_hip.h"

namespace {
struct MelParams {
    const float* x; float* out;
    const float* window;       // [512] the periodic Hann window zero-padded (centred) to n_fft
    const float* tw256;        // [256][2] cos, -sin of 2 pi k / 256   (W256^k)
    const float* tw512;        // [257][2] cos, -sin of 2 pi k / 512   (W512^k)
    const int* mel_start;      // [64] first bin of filter m
    const int* mel_count;      // [64] bins of filter m (<= 32)
    const float* mel_w;        // [64][32] weights of filter m's bins (zero padded)
    int B, N, T, hop, n_mels;
    float log_offset;
    long frames;
};
struct cplx { float re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx mul_mi(cplx a) { return {a.im, -a.re}; }          // a * (-i)
// forward radix-4 butterfly: y[q'] = sum_q a[q] * (-i)^(q q')
__device__ __forceinline__ void radix4(cplx (&a)[4]) {
    const cplx s0 = cadd(a[0], a[2]), s1 = csub(a[0], a[2]), s2 = cadd(a[1], a[3]), s3 = mul_mi(csub(a[1], a[3]));
    a[0] = cadd(s0, s2); a[2] = csub(s0, s2); a[1] = cadd(s1, s3); a[3] = csub(s1, s3);
}
__device__ __forceinline__ int rev4x4(int k) {          // reverse the four base-4 digits of k (0 .. 255)
    return ((k & 3) << 6) | (((k >> 2) & 3) << 4) | (((k >> 4) & 3) << 2) | ((k >> 6) & 3);
}

__global__ __launch_bounds__(256) void log_mel_fused_kernel(MelParams p) {
    __shared__ float2 zbuf[4][256];
    __shared__ float pbuf[4][260];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2* z = zbuf[wave];
    float* pw = pbuf[wave];
    // frame-invariant per-lane constants: window at this lane's 8 samples, twiddles of its butterflies
    float wn[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) { wn[q][0] = p.window[2 * (lane + 64 * q)]; wn[q][1] = p.window[2 * (lane + 64 * q) + 1]; }
    cplx tw[3][3];                        // [stage 0..2][q' - 1]
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int quarter = 64 >> (2 * s);                 // n / 4 for n = 256, 64, 16
        const int j = lane & (quarter - 1);
        const int mult = 1 << (2 * s);                     // 256 / n
#pragma unroll
        for (int qq = 1; qq < 4; ++qq) {
            const int k = (mult * j * qq) & 255;
            tw[s][qq - 1] = {p.tw256[2 * k], p.tw256[2 * k + 1]};
        }
    }
    cplx w512[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int k = lane + 64 * q; w512[q] = {p.tw512[2 * k], p.tw512[2 * k + 1]}; }
    const int mstart = lane < p.n_mels ? p.mel_start[lane] : 0, mcount = lane < p.n_mels ? p.mel_count[lane] : 0;
    int maxcount = mcount;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxcount = max(maxcount, __shfl_xor(maxcount, o, 64));

    const long stride = (long)gridDim.x * 4;
    for (long f = (long)blockIdx.x * 4 + wave; f < p.frames; f += stride) {
        const int b = (int)(f / p.T), t = (int)(f - (long)b * p.T);
        const float* xb = p.x + (size_t)b * p.N;
        const int base = t * p.hop - 256;                  // waveform index of frame sample 0 (center = True: n_fft / 2 of padding)
        cplx a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int i = base + 2 * (lane + 64 * q) + e;
                i = i < 0 ? -i : i;                            // reflect (no edge repeat), torch.stft pad_mode="reflect"
                i = i >= p.N ? 2 * (p.N - 1) - i : i;
                const float w = wn[q][e];
                v[e] = w != 0.f ? xb[i] * w : 0.f;             // the window is zero outside its 400 samples: those are not loaded
            }
            a[q] = {v[0], v[1]};
        }
        // ---- 256-point radix-4 DIF FFT: stage s has sub-transform length n = 256 >> 2s; lane = butterfly (g, j) ----
        radix4(a);
#pragma unroll
        for (int qq = 1; qq < 4; ++qq) a[qq] = cmul(a[qq], tw[0][qq - 1]);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) z[lane + 64 * qq] = make_float2(a[qq].re, a[qq].im);
        asm volatile("" ::: "memory");                        // wave-local exchange through LDS: program order is enough
#pragma unroll
        for (int s = 1; s < 4; ++s) {
            const int quarter = 64 >> (2 * s);                 // 16, 4, 1
            const int g = (lane / quarter) * (4 * quarter), j = lane & (quarter - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float2 v = z[g + j + q * quarter]; a[q] = {v.x, v.y}; }
            asm volatile("" ::: "memory");
            radix4(a);
            if (s < 3) {
#pragma unroll
                for (int qq = 1; qq < 4; ++qq) a[qq] = cmul(a[qq], tw[s][qq - 1]);
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) z[g + j + qq * quarter] = make_float2(a[qq].re, a[qq].im);
            asm volatile("" ::: "memory");
        }
        // ---- real-input split + power: Z[k] sits at z[rev(k)] ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = lane + 64 * q;
            const float2 zk = z[rev4x4(k)], zm = z[rev4x4((256 - k) & 255)];
            const cplx e = {0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};           // (Z[k] + conj Z[N-k]) / 2
            const cplx d = {0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y)};           // (Z[k] - conj Z[N-k]) / 2
            const cplx o = mul_mi(d);                                              // odd-sample spectrum
            const cplx xk = cadd(e, cmul(w512[q], o));
            pw[k] = xk.re * xk.re + xk.im * xk.im;
            if (k == 0) {                                                          // X[256] = Xe[0] - Xo[0]
                const cplx xn = csub(e, o);
                pw[256] = xn.re * xn.re + xn.im * xn.im;
            }
        }
        asm volatile("" ::: "memory");
        // ---- triangular mel filters: lane m sums its run of bins ----
        float mel = 0.f;
        const float* wrow = p.mel_w + lane * 32;
        for (int i = 0; i < maxcount; ++i)
            if (i < mcount) mel = fmaf(wrow[i], pw[mstart + i], mel);
        if (lane < p.n_mels) p.out[((size_t)b * p.T + t) * p.n_mels + lane] = logf(mel + p.log_offset);
        asm volatile("" ::: "memory");                        // the next frame's stage-0 stores stay behind these reads
    }
}
}   // namespace

extern "C" int v100_log_mel_fused(const float* x, float* out, const float* window, const float* tw256, const float* tw512,
                                  const int* mel_start, const int* mel_count, const float* mel_w, int B, int N, int T, int hop,
                                  int n_fft, int n_mels, float log_offset, void* stream) {
    if (!x || !out || !window || !tw256 || !tw512 || !mel_start || !mel_count || !mel_w) return V100_ERR_NULL;
    if (B <= 0 || T <= 0 || hop <= 0 || n_fft != 512 || n_mels <= 0 || n_mels > 64 || N <= n_fft / 2) return V100_ERR_SHAPE;
    if ((long)(T - 1) * hop > N) return V100_ERR_SHAPE;
    MelParams p{x, out, window, tw256, tw512, mel_start, mel_count, mel_w, B, N, T, hop, n_mels, log_offset, (long)B * T};
    long blocks = (p.frames + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    V100_GGL(log_mel_fused_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return v100_launch_status();
}
