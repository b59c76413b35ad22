// Adam step of the whole model in ONE launch (torch.optim.Adam as the reference configures it: asr.py:169-176, tts.py:132-135,
// 239-241 -- L2-style weight_decay added to the gradient, bias correction, no amsgrad).  PyTorch's fused multi-tensor Adam
// takes 3 launches and ~145 us for the 11.6 M parameters of asr_en_base (its tensor lists travel in 4 KB kernel-argument
// chunks); here the (parameter, moment) pointers and the chunk list live in device tables that are built once, and only
// the gradient pointers -- autograd hands out fresh gradient tensors every step -- are uploaded per step (8 bytes per tensor).
#include "common.h"
#include <math.h>

struct AdamChunk { int tensor; int count; long long offset; };     // `count` elements of tensor `tensor` starting at `offset`

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamChunk* __restrict__ chunks, float* const* __restrict__ params,
                                                        const float* const* __restrict__ grads, float* const* __restrict__ exp_avg,
                                                        float* const* __restrict__ exp_avg_sq, float lr_over_bc1, float omb1,
                                                        float beta2, float omb2, float eps, float weight_decay, float inv_sqrt_bc2) {
    const AdamChunk ch = chunks[blockIdx.x];
    float* __restrict__ p = params[ch.tensor] + ch.offset;
    const float* __restrict__ g = grads[ch.tensor] + ch.offset;
    float* __restrict__ m = exp_avg[ch.tensor] + ch.offset;
    float* __restrict__ v = exp_avg_sq[ch.tensor] + ch.offset;
    auto upd = [&](float& pv, float gv, float& mv, float& vv) {
        gv = fmaf(weight_decay, pv, gv);
        // omb1 / omb2 = 1 - beta formed in double on the host, as torch does (1.f - 0.999f is off by 1.3e-5 relative)
        mv = fmaf(omb1, gv - mv, mv);                              // exp_avg.lerp_(grad, 1 - beta1)
        vv = fmaf(omb2, gv * gv, beta2 * vv);                      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        pv -= lr_over_bc1 * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    };
    const bool vec = ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0);
    if (vec) {
        const int n4 = ch.count >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
            const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = pv[e], b = mv[e], c = vv[e];
                upd(a, gv[e], b, c);
                pv[e] = a; mv[e] = b; vv[e] = c;
            }
            reinterpret_cast<f32x4*>(p)[i] = pv; reinterpret_cast<f32x4*>(m)[i] = mv; reinterpret_cast<f32x4*>(v)[i] = vv;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < ch.count; i += 256) upd(p[i], g[i], m[i], v[i]);
    } else {
        for (int i = threadIdx.x; i < ch.count; i += 256) upd(p[i], g[i], m[i], v[i]);
    }
}

extern "C" int v100_adam_chunk_elems() { return 16384; }

// chunks: device array of nchunks {int tensor, int count, long long offset}; params / grads / exp_avg / exp_avg_sq: device arrays of
// float pointers indexed by tensor.  step >= 1 is the number of this update (bias corrections 1 - beta^step).
extern "C" int v100_adam_step(const void* chunks, int nchunks, const void* params, const void* grads, const void* exp_avg,
                              const void* exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                              void* stream) {
    if (!chunks || !params || !grads || !exp_avg || !exp_avg_sq) return V100_ERR_NULL;
    if (nchunks <= 0 || step < 1) return V100_ERR_SHAPE;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    V100_GGL(adam_step_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)chunks,
                       (float* const*)params, (const float* const*)grads, (float* const*)exp_avg, (float* const*)exp_avg_sq,
                       (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                       (float)(1.0 / sqrt(bc2)));
    return v100_launch_status();
}
