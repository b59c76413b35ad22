// K8: BatchSpectrogramAugumentation (voice100/audio.py:17-108) as ONE fused pass over the log-mel
// batch [B, T, F] (F = mel bins, contiguous).  The reference applies up to seven tensor ops in
// sequence (index_select along T, index_select along F, subtract, two slice fills, exp/log noise
// mix, exp/mask/clamp/log); every one is a pure function of the output coordinate, so the chain is
// evaluated per output element: one gather read (two for mixaudio) and one write, HBM-bound.
// The random decisions are drawn on the host exactly as the reference draws them and arrive here
// as plain numbers.
#include "common.h"

struct AugParams {
    const float* x;          // [B, Tin, F]
    const int* len;          // [B] lengths AFTER timestretch, or (raw_len) BEFORE it: the kernel derives len*rate/100 itself
    int raw_len;
    int* len_out;            // raw_len form: [B] stretched lengths and [B] (stretched + 1) / 2 (the encoder's output lengths,
    int* half_out;           //   asr.py:80-81), written by block 0 so the step needs no integer torch ops for them
    const float* uniform;    // [B, Tout, F] torch.rand draw for mixnoise (or null)
    float* y;                // [B, Tout, F]
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

__global__ __launch_bounds__(256) void augment_fused_kernel(AugParams p) {
    if (p.raw_len && blockIdx.x == 0)
        for (int b = threadIdx.x; b < p.B; b += 256) {
            const int l = aug_len(p, b);
            if (p.len_out) p.len_out[b] = l;
            if (p.half_out) p.half_out[b] = (l + 1) / 2;
        }
    const long total = (long)p.B * p.Tout * p.F;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int f = (int)(i % p.F);
        const int t = (int)((i / p.F) % p.Tout);
        const int b = (int)(i / ((long)p.F * p.Tout));
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
        p.y[i] = logf(fmaxf(out, p.log_offset));
    }
}

static int augment_launch(const float* x, const int* len, int raw_len, int* len_out, int* half_out, const float* uniform, float* y,
                          int B, int Tin, int Tout, int F, int stretch_rate, float pitch_rate, float amp, int n_tmask,
                          const int* tm_s, const int* tm_e, const float* tm_a, int fm_on, int fm_s, int fm_e, float fm_a,
                          int noise_on, float noise_low, float noise_high, float noise_std, int mix, float log_offset,
                          void* stream) {
    if (!x || !len || !y) return V100_ERR_NULL;
    if (B <= 0 || Tin <= 0 || Tout <= 0 || F <= 0 || n_tmask < 0 || n_tmask > 3) return V100_ERR_SHAPE;
    if (noise_on && (!uniform || F != 64)) return V100_ERR_SHAPE;     // audio.py:96 hard-codes 64 mel bins
    if (stretch_rate && ((long)(Tout - 1) * 100 / stretch_rate >= Tin)) return V100_ERR_SHAPE;
    AugParams p;
    p.x = x; p.len = len; p.uniform = uniform; p.y = y;
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
    return augment_launch(x, len, 0, nullptr, nullptr, uniform, y, B, Tin, Tout, F, stretch_rate, pitch_rate, amp, n_tmask, tm_s,
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
    return augment_launch(x, len_raw, 1, len_out, half_out, uniform, y, B, Tin, Tout, F, stretch_rate, pitch_rate, amp, n_tmask,
                          tm_s, tm_e, tm_a, fm_on, fm_s, fm_e, fm_a, noise_on, noise_low, noise_high, noise_std, mix, log_offset,
                          stream);
}
