// K4 glue, K7, K9: the small HBM-bound kernels around the conv stacks.
//
//   shift_copy        strided / shifted channel-block copy: builds the tap-stacked operand of
//                     ConvTranspose1d(k5,s2,p2) (voice100/models/tts.py:22) for the pointwise GEMM,
//                     interleaves its even/odd output phases (+bias), and the inverse for backward
//   stft_frames       center/reflect framing of a waveform into [B][win][T] columns (hop 160),
//                     voice100/data_modules.py:276-281 (torchaudio Spectrogram semantics)
//   power_spectrum    |re|^2 + |im|^2 of the real-DFT GEMM output
//   log_transpose     log(mel + 1e-6) with the [B][n_mels][T] -> [B][T][n_mels] layout change
//                     (data_modules.py:290-291)
//   world_unnormalize x*std + mean per feature and the F0 gate (tts.py:197-200, _layers_v1.py:132-138)
//   exp_clip          max(exp(x) - offset, 0)   (voice100/vocoder.py:99)
#include "common.h"

__global__ __launch_bounds__(256) void shift_copy_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         const float* __restrict__ bias, int C, int Tin, int Tout, int in_ctot,
                                                         int in_coff, int out_ctot, int out_coff, int in_mul, int in_add, int out_mul,
                                                         int out_add, int n, int accumulate, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int u = (int)(i % n);
        const int c = (int)((i / n) % C);
        const int b = (int)(i / ((long)n * C));
        const int to = u * out_mul + out_add;
        if (to < 0 || to >= Tout) continue;
        const int ti = u * in_mul + in_add;
        float v = 0.f;
        if (ti >= 0 && ti < Tin) v = in[((size_t)b * in_ctot + in_coff + c) * Tin + ti];
        if (bias) v += bias[c];
        float* dst = out + ((size_t)b * out_ctot + out_coff + c) * Tout + to;
        *dst = accumulate ? *dst + v : v;
    }
}

// frames[b][n][t] = x[b][reflect(t*hop + n + left - n_fft/2)], n in [0, win), left = (n_fft - win)/2
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ x, float* __restrict__ frames, int N, int T,
                                                          int hop, int win, int n_fft, long total) {
    const int left = (n_fft - win) / 2, half = n_fft / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const int n = (int)((i / T) % win);
        const int b = (int)(i / ((long)T * win));
        int idx = t * hop + n + left - half;
        if (idx < 0) idx = -idx;
        if (idx >= N) idx = 2 * (N - 1) - idx;
        frames[i] = x[(size_t)b * N + idx];
    }
}

__global__ __launch_bounds__(256) void power_spectrum_kernel(const float* __restrict__ spec, float* __restrict__ pw, int F, int T, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const int f = (int)((i / T) % F);
        const int b = (int)(i / ((long)T * F));
        const float re = spec[((size_t)b * 2 * F + f) * T + t];
        const float im = spec[((size_t)b * 2 * F + F + f) * T + t];
        pw[i] = fmaf(re, re, im * im);
    }
}

// out[b][t][c] = log(in[b][c][t] + offset)
__global__ __launch_bounds__(256) void log_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T, float offset) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, t = t0 + tx;
        if (c < C && t < T) tile[ty + 8 * i][tx] = logf(in[((size_t)b * C + c) * T + t] + offset);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, c = c0 + tx;
        if (c < C && t < T) out[((size_t)b * T + t) * C + c] = tile[tx][ty + 8 * i];
    }
}

// x [B][T][A] with A = 1 (hasf0 logit) + 1 (f0) + S (logspc) + Cap (codeap)  ->  f0 [B][T], logspc [B][T][S], codeap [B][T][Cap]
__global__ __launch_bounds__(256) void world_unnormalize_kernel(const float* __restrict__ x, float* __restrict__ f0, float* __restrict__ logspc,
                                                                float* __restrict__ codeap, const float* __restrict__ f0_mean,
                                                                const float* __restrict__ f0_std, const float* __restrict__ ls_mean,
                                                                const float* __restrict__ ls_std, const float* __restrict__ ca_mean,
                                                                const float* __restrict__ ca_std, int S, int Cap, long rows) {
    const int A = 2 + S + Cap;
    const long total = rows * A;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int a = (int)(i % A);
        const long r = i / A;
        const float v = x[i];
        if (a == 0) continue;
        if (a == 1) {
            const float gate = x[r * A];
            const float y = fmaf(f0_std[0], v, f0_mean[0]);
            f0[r] = gate < 0.f ? 0.f : y;
        } else if (a < 2 + S) {
            const int s = a - 2;
            logspc[r * S + s] = fmaf(ls_std[s], v, ls_mean[s]);
        } else {
            const int s = a - 2 - S;
            codeap[r * Cap + s] = fmaf(ca_std[s], v, ca_mean[s]);
        }
    }
}

__global__ void exp_clip_kernel(const float* __restrict__ x, float* __restrict__ y, float offset, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = fmaxf(expf(x[i]) - offset, 0.f);
}

static inline unsigned grid_for(long total) {
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

extern "C" int v100_shift_copy(const float* in, float* out, const float* bias, int B, int C, int Tin, int Tout, int in_ctot,
                               int in_coff, int out_ctot, int out_coff, int in_mul, int in_add, int out_mul, int out_add, int n,
                               int accumulate, void* stream) {
    if (!in || !out) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || Tin <= 0 || Tout <= 0 || n <= 0 || in_coff < 0 || out_coff < 0 || in_coff + C > in_ctot ||
        out_coff + C > out_ctot || out_mul <= 0 || in_mul <= 0) return V100_ERR_SHAPE;
    const long total = (long)B * C * n;
    V100_GGL(shift_copy_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, bias, C, Tin, Tout, in_ctot,
                       in_coff, out_ctot, out_coff, in_mul, in_add, out_mul, out_add, n, accumulate, total);
    return v100_launch_status();
}

extern "C" int v100_stft_frames(const float* x, float* frames, int B, int N, int T, int hop, int win, int n_fft, void* stream) {
    if (!x || !frames) return V100_ERR_NULL;
    if (B <= 0 || N <= n_fft / 2 || T <= 0 || hop <= 0 || win <= 0 || win > n_fft) return V100_ERR_SHAPE;
    if ((long)(T - 1) * hop > N) return V100_ERR_SHAPE;
    const long total = (long)B * win * T;
    V100_GGL(stft_frames_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, frames, N, T, hop, win, n_fft, total);
    return v100_launch_status();
}

extern "C" int v100_power_spectrum(const float* spec, float* pw, int B, int F, int T, void* stream) {
    if (!spec || !pw) return V100_ERR_NULL;
    if (B <= 0 || F <= 0 || T <= 0) return V100_ERR_SHAPE;
    const long total = (long)B * F * T;
    V100_GGL(power_spectrum_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, spec, pw, F, T, total);
    return v100_launch_status();
}

extern "C" int v100_log_transpose(const float* in, float* out, int B, int C, int T, float offset, void* stream) {
    if (!in || !out) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || T <= 0) return V100_ERR_SHAPE;
    V100_GGL(log_transpose_kernel, dim3(ceil_div(T, 32), ceil_div(C, 32), B), dim3(256), 0, (hipStream_t)stream, in, out, C, T, offset);
    return v100_launch_status();
}

extern "C" int v100_world_unnormalize(const float* x, float* f0, float* logspc, float* codeap, const float* f0_mean, const float* f0_std,
                                      const float* ls_mean, const float* ls_std, const float* ca_mean, const float* ca_std,
                                      int B, int T, int S, int Cap, void* stream) {
    if (!x || !f0 || !logspc || !codeap || !f0_mean || !f0_std || !ls_mean || !ls_std || !ca_mean || !ca_std) return V100_ERR_NULL;
    if (B <= 0 || T <= 0 || S <= 0 || Cap <= 0) return V100_ERR_SHAPE;
    const long rows = (long)B * T;
    V100_GGL(world_unnormalize_kernel, dim3(grid_for(rows * (2 + S + Cap))), dim3(256), 0, (hipStream_t)stream, x, f0, logspc, codeap,
                       f0_mean, f0_std, ls_mean, ls_std, ca_mean, ca_std, S, Cap, rows);
    return v100_launch_status();
}

extern "C" int v100_exp_clip(const float* x, float* y, float offset, long long n, void* stream) {
    if (!x || !y) return V100_ERR_NULL;
    if (n <= 0) return V100_ERR_SHAPE;
    V100_GGL(exp_clip_kernel, dim3(grid_for((long)n)), dim3(256), 0, (hipStream_t)stream, x, y, offset, (long)n);
    return v100_launch_status();
}
