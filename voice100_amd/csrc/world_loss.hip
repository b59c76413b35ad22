// K12: WORLDLoss (voice100/models/_layers_v1.py:37-93) with the target preparation of
// AlignTextToAudioModel._calc_batch_loss (voice100/models/tts.py:203-206) in one pass, value and gradient.
//
// The reference runs ~25 elementwise / reduction kernels over [B, T, 2+S+Cap] tensors per step: hasf0 = f0 >= 30,
// WORLDNorm.normalize of the three targets, adjust_size, the padding mask, BCE-with-logits, three squared (or absolute)
// errors with the mel-slope weights, four masked sums, the mask sum, four divisions -- and as many again in backward.
// Here one wave owns one frame (b, t) of the decoder output pred [B][Tp][A], A = 1 + 1 + S + Cap, reads the frame and its
// targets once, forms the four loss terms, and writes the frame's gradient for unit upstream gradients straight away;
// per-block partial sums go to a slab that one small kernel reduces in a fixed order (deterministic, no atomics).
#include "common.h"
#include <math.h>

struct WorldLossParams {
    const float* pred;      // [B][Tp][A]
    const float* f0;        // [B][Tt]       raw (norm given) or normalised
    const float* hasf0;     // [B][Tt] or null (then hasf0 = raw f0 >= 30, needs norm)
    const float* logspc;    // [B][Tt][S]
    const float* codeap;    // [B][Tt][Cap]
    const int* length;      // [B]
    const float* f0_mean; const float* f0_std; const float* ls_mean; const float* ls_std; const float* ca_mean; const float* ca_std;
    const float* w;         // [S] or null
    float* partial;         // [nblocks][4]
    float* unit;            // [B][Tp][A]
    int B, Tp, Tt, S, Cap, l1;
};

__device__ __forceinline__ void wl_el(bool l1, float d, float& v, float& g) {
    if (l1) { v = fabsf(d); g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    else { v = d * d; g = 2.f * d; }
}

// sum over b of clamp(length[b], 0, n): the number of unmasked frames (torch.sum(mask), _layers_v1.py:80,88-92)
__device__ __forceinline__ float wl_mask_sum(const int* __restrict__ length, int B, int n, int lane) {
    float s = 0.f;
    for (int b = lane; b < B; b += 64) s += (float)min(max(length[b], 0), n);
    return wave_sum(s);
}

__global__ __launch_bounds__(256) void world_loss_kernel(WorldLossParams p) {
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A = 2 + p.S + p.Cap;
    const int n = min(p.Tp, p.Tt);
    const long fi = (long)blockIdx.x * 4 + wave;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (fi < (long)p.B * p.Tp) {
        const int b = (int)(fi / p.Tp), t = (int)(fi % p.Tp);
        const float* pr = p.pred + fi * A;
        float* un = p.unit + fi * A;
        const bool valid = t < n && t < p.length[b];                 // wave-uniform
        if (!valid) {
            for (int a = lane; a < A; a += 64) un[a] = 0.f;
        } else {
            const float inv_ms = 1.f / wl_mask_sum(p.length, p.B, n, lane);
            const bool l1 = p.l1 != 0;
            const bool norm = p.f0_mean != nullptr;
            const size_t ti = (size_t)b * p.Tt + t;
            const float f0raw = p.f0[ti];
            const float z = p.hasf0 ? p.hasf0[ti] : (f0raw >= 30.0f ? 1.f : 0.f);
            const float f0t = norm ? (f0raw - p.f0_mean[0]) / p.f0_std[0] : f0raw;
            if (lane == 0) {
                const float x = pr[0];
                // binary_cross_entropy_with_logits: max(x, 0) - x*z + log(1 + exp(-|x|))
                t0 = fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
                un[0] = (1.f / (1.f + expf(-x)) - z) * inv_ms;
                float v, g;
                wl_el(l1, pr[1] - f0t, v, g);
                t1 = v * z;
                un[1] = g * z * inv_ms;
            }
            const float wmean = 1.f / (float)p.S;
            for (int s = lane; s < p.S; s += 64) {
                float tg = p.logspc[ti * p.S + s];
                if (norm) tg = (tg - p.ls_mean[s]) / p.ls_std[s];
                const float ws = p.w ? p.w[s] : wmean;
                float v, g;
                wl_el(l1, pr[2 + s] - tg, v, g);
                t2 = fmaf(v, ws, t2);
                un[2 + s] = g * ws * inv_ms;
            }
            const float cmean = 1.f / (float)p.Cap;
            for (int s = lane; s < p.Cap; s += 64) {
                float tg = p.codeap[ti * p.Cap + s];
                if (norm) tg = (tg - p.ca_mean[s]) / p.ca_std[s];
                float v, g;
                wl_el(l1, pr[2 + p.S + s] - tg, v, g);
                t3 = fmaf(v, cmean, t3);
                un[2 + p.S + s] = g * cmean * inv_ms;
            }
        }
    }
    t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2); t3 = wave_sum(t3);
    if (lane == 0) { red[wave][0] = t0; red[wave][1] = t1; red[wave][2] = t2; red[wave][3] = t3; }
    __syncthreads();
    if (threadIdx.x < 4)
        p.partial[(size_t)blockIdx.x * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// loss[k] = (sum of the partials in block order, pairwise inside the block of 256 threads) / sum(mask)
__global__ __launch_bounds__(256) void world_loss_finalize_kernel(const float* __restrict__ partial, int nparts, const int* __restrict__ length,
                                                                 int B, int n, float* __restrict__ loss) {
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nparts; i += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += partial[(size_t)i * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = wave_sum(s[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave][k] = s[k];
    const float ms = wl_mask_sum(length, B, n, lane);
    __syncthreads();
    if (threadIdx.x < 4) loss[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / ms;
}

// dpred[b][t][a] = unit[b][t][a] * gout[term(a)]
__global__ void world_loss_bwd_kernel(const float* __restrict__ unit, const float* __restrict__ gout, float* __restrict__ dpred, int S, int Cap,
                                      long total) {
    const int A = 2 + S + Cap;
    const float g0 = gout[0], g1 = gout[1], g2 = gout[2], g3 = gout[3];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int a = (int)(i % A);
        const float g = a == 0 ? g0 : (a == 1 ? g1 : (a < 2 + S ? g2 : g3));
        dpred[i] = unit[i] * g;
    }
}

extern "C" int v100_world_loss_parts(int B, int Tp) { return (int)(((long)B * Tp + 3) / 4); }

extern "C" int v100_world_loss(const float* pred, const float* f0, const float* hasf0, const float* logspc, const float* codeap,
                               const int* length, const float* f0_mean, const float* f0_std, const float* ls_mean, const float* ls_std,
                               const float* ca_mean, const float* ca_std, const float* w, float* partial, float* loss, float* unit,
                               int B, int Tp, int Tt, int S, int Cap, int l1, void* stream) {
    if (!pred || !f0 || !logspc || !codeap || !length || !partial || !loss || !unit) return V100_ERR_NULL;
    if (B <= 0 || Tp <= 0 || Tt <= 0 || S <= 0 || Cap <= 0) return V100_ERR_SHAPE;
    const bool norm = f0_mean != nullptr;
    if (norm && (!f0_std || !ls_mean || !ls_std || !ca_mean || !ca_std)) return V100_ERR_NULL;
    if (!norm && (f0_std || ls_mean || ls_std || ca_mean || ca_std)) return V100_ERR_SHAPE;     // all six or none
    if (!norm && !hasf0) return V100_ERR_NULL;       // hasf0 can only be derived from the RAW f0
    hipStream_t st = (hipStream_t)stream;
    WorldLossParams p{pred, f0, hasf0, logspc, codeap, length, f0_mean, f0_std, ls_mean, ls_std, ca_mean, ca_std, w, partial, unit,
                      B, Tp, Tt, S, Cap, l1};
    const int nparts = v100_world_loss_parts(B, Tp);
    V100_GGL(world_loss_kernel, dim3(nparts), dim3(256), 0, st, p);
    V100_GGL(world_loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nparts, length, B, Tp < Tt ? Tp : Tt, loss);
    return v100_launch_status();
}

extern "C" int v100_world_loss_bwd(const float* unit, const float* gout, float* dpred, int B, int Tp, int S, int Cap, void* stream) {
    if (!unit || !gout || !dpred) return V100_ERR_NULL;
    if (B <= 0 || Tp <= 0 || S <= 0 || Cap <= 0) return V100_ERR_SHAPE;
    const long total = (long)B * Tp * (2 + S + Cap);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    V100_GGL(world_loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, unit, gout, dpred, S, Cap, total);
    return v100_launch_status();
}
