// K8: BatchSpectrogramAugumentation (voice100/audio.py:17-108) as ONE fused pass over the log-mel
// batch [B, T, F] (F = mel bins, contiguous).  The reference applies up to seven tensor ops in
// sequence (index_select along T, index_select along F, subtract, two slice fills, exp/log noise
// mix, exp/mask/clamp/log); every one is a pure function of the output coordinate, so the chain is
// evaluated per output element: one gather read (two for mixaudio) and one write, HBM-bound.
// The random decisions are drawn on the host exactly as the reference draws them and arrive here
// as plain numbers.
#include "common.h"
#include <cstdlib>

struct AugParams {
    const float* x;          // [B, Tin, F]
    const int* len;          // [B] lengths AFTER timestretch, or (raw_len) BEFORE it: the kernel derives len*rate/100 itself
    int raw_len;
    int* len_out;            // raw_len form: [B] stretched lengths and [B] (stretched + 1) / 2 (the encoder's output lengths,
    int* half_out;           //   asr.py:80-81), written by block 0 so the step needs no integer torch ops for them
    const float* uniform;    // [B, Tout, F] torch.rand draw for mixnoise (or null)
    float* y;                // [B, Tout, F]
    float* yt;               // optional [B, F, Tout]: the same values transposed (the encoder's input layout, asr.py:111)
    int B, Tin, Tout, F;
    int stretch_rate;        // 0 = off, else 50..149 : t_src = t*100/rate
    float pitch_rate;        // 0 = off, else f_src = clamp(int(f*rate))
    float amp;               // subtracted (0 = off)
    int n_tmask; int tm_s[3]; int tm_e[3]; float tm_a[3];   // normalised [s,e) spans along T
    int fm_on; int fm_s; int fm_e; float fm_a;
    int noise_on; float noise_low, noise_high, noise_std;
    int mix;                 // 1 = mixaudio (0.9 x + 0.1 next utterance), 0 = maskaudio
    float log_offset;
};

__device__ __forceinline__ float aug_chain(const AugParams& p, int b, int t, int f) {
    const int ts = p.stretch_rate ? (t * 100) / p.stretch_rate : t;
    int fs = f;
    if (p.pitch_rate != 0.f) {
        fs = (int)((float)f * p.pitch_rate);
        fs = fs < 0 ? 0 : (fs > p.F - 1 ? p.F - 1 : fs);
    }
    float v = p.x[((size_t)b * p.Tin + ts) * p.F + fs];
    v -= p.amp;
    for (int i = 0; i < p.n_tmask; ++i)
        if (t >= p.tm_s[i] && t < p.tm_e[i]) v = p.tm_a[i];
    if (p.fm_on && f >= p.fm_s && f < p.fm_e) v = p.fm_a;
    if (p.noise_on) {
        // torch.linspace(low, high, 64): symmetric evaluation from both ends (step = (high-low)/63)
        const float step = (p.noise_high - p.noise_low) / 63.0f;
        const float scale = (f < 32) ? p.noise_low + step * (float)f : p.noise_high - step * (float)(63 - f);
        const float nz = p.uniform[((size_t)b * p.Tout + t) * p.F + f] * p.noise_std + scale;
        v = logf(fmaxf(expf(v) + expf(nz), p.log_offset));
    }
    return v;
}

__device__ __forceinline__ int aug_len(const AugParams& p, int b) {
    const int l = p.len[b];
    return (p.raw_len && p.stretch_rate) ? (l * p.stretch_rate) / 100 : l;      // audio.py:58 (trunc == floor: lengths are >= 0)
}

// the value of output element (b, t, f): audio.py:52-108 evaluated per output coordinate
__device__ __forceinline__ float aug_element(const AugParams& p, int b, int t, int f) {
    const float m = t < aug_len(p, b) ? 1.f : 0.f;
    const float xv = expf(aug_chain(p, b, t, f)) * m;
    float out;
    if (p.mix) {
        const int b2 = (b + 1 == p.B) ? 0 : b + 1;
        const float m2 = t < aug_len(p, b2) ? 1.f : 0.f;
        const float yv = expf(aug_chain(p, b2, t, f)) * m2;
        out = (0.9f * xv + 0.1f * yv) * m;
    } else {
        out = xv;
    }
    return logf(fmaxf(out, p.log_offset));
}

__device__ __forceinline__ void aug_write_lengths(const AugParams& p) {
    for (int b = threadIdx.x; b < p.B; b += 256) {
        const int l = aug_len(p, b);
        if (p.len_out) p.len_out[b] = l;
        if (p.half_out) p.half_out[b] = (l + 1) / 2;
    }
}

__global__ __launch_bounds__(256) void augment_fused_kernel(AugParams p) {
    if (p.raw_len && blockIdx.x == 0) aug_write_lengths(p);
    const long total = (long)p.B * p.Tout * p.F;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int f = (int)(i % p.F);
        const int t = (int)((i / p.F) % p.Tout);
        const int b = (int)(i / ((long)p.F * p.Tout));
        p.y[i] = aug_element(p, b, t, f);
    }
}

// F == 64 (audio.py:96 fixes 64 mel bins): a wave is one frame, lane = mel bin -- every index but f is wave-uniform (the element form
// above spends three 64-bit divisions per element and runs at 1 TB/s).  A workgroup owns 64 consecutive frames of one utterance and,
// when the transposed twin is asked for, passes them through LDS so that both layouts are written in whole 256-byte rows: the
// encoder's [B, 64, T] input (asr.py:111) then costs one extra store instead of a transpose launch.  Same arithmetic per element.
__global__ __launch_bounds__(1024) void augment_rows64_kernel(AugParams p) {
    __shared__ float tile[64][65];
    if (p.raw_len && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 256) aug_write_lengths(p);
    const int b = blockIdx.y, t0 = blockIdx.x * 64, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // 16 waves x 4 frames, the four frames' chains side by side (a wave that walked 16 frames one after the other ran at the latency
    // of its gathers: 21 us against the element form's 15)
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + wave + 16 * i;
        v[i] = t < p.Tout ? aug_element(p, b, t, lane) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave + 16 * i, t = t0 + r;
        if (t < p.Tout) p.y[((size_t)b * p.Tout + t) * 64 + lane] = v[i];
        if (p.yt) tile[lane][r] = v[i];
    }
    if (!p.yt) return;
    __syncthreads();
    const int t = t0 + lane;
    if (t < p.Tout)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int f = wave + 16 * i; p.yt[((size_t)b * 64 + f) * p.Tout + t] = tile[f][lane]; }
}

static int augment_launch(const float* x, const int* len, int raw_len, int* len_out, int* half_out, const float* uniform, float* y,
                          float* yt,
                          int B, int Tin, int Tout, int F, int stretch_rate, float pitch_rate, float amp, int n_tmask,
                          const int* tm_s, const int* tm_e, const float* tm_a, int fm_on, int fm_s, int fm_e, float fm_a,
                          int noise_on, float noise_low, float noise_high, float noise_std, int mix, float log_offset,
                          void* stream) {
    if (!x || !len || !y) return V100_ERR_NULL;
    if (B <= 0 || Tin <= 0 || Tout <= 0 || F <= 0 || n_tmask < 0 || n_tmask > 3) return V100_ERR_SHAPE;
    if (noise_on && (!uniform || F != 64)) return V100_ERR_SHAPE;     // audio.py:96 hard-codes 64 mel bins
    if (stretch_rate && ((long)(Tout - 1) * 100 / stretch_rate >= Tin)) return V100_ERR_SHAPE;
    AugParams p;
    p.x = x; p.len = len; p.uniform = uniform; p.y = y; p.yt = yt;
    p.raw_len = raw_len; p.len_out = len_out; p.half_out = half_out;
    p.B = B; p.Tin = Tin; p.Tout = Tout; p.F = F;
    p.stretch_rate = stretch_rate; p.pitch_rate = pitch_rate; p.amp = amp;
    p.n_tmask = n_tmask;
    for (int i = 0; i < 3; ++i) {       // tm_* are HOST arrays (three small numbers drawn on the host)
        p.tm_s[i] = i < n_tmask ? tm_s[i] : 0; p.tm_e[i] = i < n_tmask ? tm_e[i] : 0; p.tm_a[i] = i < n_tmask ? tm_a[i] : 0.f;
    }
    p.fm_on = fm_on; p.fm_s = fm_s; p.fm_e = fm_e; p.fm_a = fm_a;
    p.noise_on = noise_on; p.noise_low = noise_low; p.noise_high = noise_high; p.noise_std = noise_std;
    p.mix = mix; p.log_offset = log_offset;
    const char* rows = getenv("V100_AUG_ROWS");      // 0: the element form for every F (A/B and the equality test; read per call)
    if (F == 64 && B <= 65535 && !(rows && rows[0] == '0')) {
        V100_GGL(augment_rows64_kernel, dim3((unsigned)ceil_div(Tout, 64), (unsigned)B), dim3(1024), 0, (hipStream_t)stream, p);
        return v100_launch_status();
    }
    if (yt) return V100_ERR_SHAPE;        // the transposed twin exists in the 64-bin form only
    const long total = (long)B * Tout * F;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    V100_GGL(augment_fused_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return v100_launch_status();
}

extern "C" int v100_augment_fused(const float* x, const int* len, const float* uniform, float* y, int B, int Tin, int Tout,
                                  int F, int stretch_rate, float pitch_rate, float amp, int n_tmask, const int* tm_s,
                                  const int* tm_e, const float* tm_a, int fm_on, int fm_s, int fm_e, float fm_a, int noise_on,
                                  float noise_low, float noise_high, float noise_std, int mix, float log_offset, void* stream) {
    return augment_launch(x, len, 0, nullptr, nullptr, uniform, y, nullptr, B, Tin, Tout, F, stretch_rate, pitch_rate, amp, n_tmask, tm_s,
                          tm_e, tm_a, fm_on, fm_s, fm_e, fm_a, noise_on, noise_low, noise_high, noise_std, mix, log_offset, stream);
}

// The same pass fed the lengths BEFORE timestretch: it derives the stretched lengths itself (len * rate / 100, audio.py:58) and
// also writes them (len_out) and the encoder's output lengths (half_out = (len_out + 1) / 2, asr.py:80-81) -- the three integer
// tensor ops a training step otherwise spends five tiny launches on.  Either output may be null.
extern "C" int v100_augment_fused_len(const float* x, const int* len_raw, int* len_out, int* half_out, const float* uniform,
                                      float* y, int B, int Tin, int Tout, int F, int stretch_rate, float pitch_rate, float amp,
                                      int n_tmask, const int* tm_s, const int* tm_e, const float* tm_a, int fm_on, int fm_s,
                                      int fm_e, float fm_a, int noise_on, float noise_low, float noise_high, float noise_std,
                                      int mix, float log_offset, void* stream) {
    return augment_launch(x, len_raw, 1, len_out, half_out, uniform, y, nullptr, B, Tin, Tout, F, stretch_rate, pitch_rate, amp, n_tmask,
                          tm_s, tm_e, tm_a, fm_on, fm_s, fm_e, fm_a, noise_on, noise_low, noise_high, noise_std, mix, log_offset,
                          stream);
}

// ... and with the transposed twin yt [B, F, Tout] written by the same pass (F == 64 only; yt may be null)
extern "C" int v100_augment_fused_len_t(const float* x, const int* len_raw, int* len_out, int* half_out, const float* uniform,
                                        float* y, float* yt, int B, int Tin, int Tout, int F, int stretch_rate, float pitch_rate,
                                        float amp, int n_tmask, const int* tm_s, const int* tm_e, const float* tm_a, int fm_on,
                                        int fm_s, int fm_e, float fm_a, int noise_on, float noise_low, float noise_high,
                                        float noise_std, int mix, float log_offset, void* stream) {
    return augment_launch(x, len_raw, 1, len_out, half_out, uniform, y, yt, B, Tin, Tout, F, stretch_rate, pitch_rate, amp, n_tmask,
                          tm_s, tm_e, tm_a, fm_on, fm_s, fm_e, fm_a, noise_on, noise_low, noise_high, noise_std, mix, log_offset,
                          stream);
}
