// Weight-gradient partial sums: S partial [M x K] fp32 tiles written as slabs and summed by a second launch (what the library does:
// deterministic, slab order fixed) against S workgroups adding their tile into ONE [M x K] buffer with fp32 atomics (no slabs, no
// second launch, order not fixed).  M x K = 2048 x 512 (the 512-wide blocks' dW), 256 x 128 tiles, S = 16: 512 workgroups, as the
// step's pw_wgrad launch.  Only the epilogue traffic is modelled (each workgroup stores a tile of constants).
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics tools/micro/slab_vs_atomic.hip -o /tmp/sva && /tmp/sva
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int M = 2048, K = 512, TM = 256, TK = 128, S = 16, NSET = 6;
__global__ __launch_bounds__(256) void slab_store(float* __restrict__ slabs, float v) {
    const int tile = blockIdx.x % 32, s = blockIdx.x / 32, tm = tile / 4, tk = tile % 4;
    float* base = slabs + (size_t)s * M * K + (size_t)tm * TM * K + tk * TK;
    for (int i = threadIdx.x; i < TM * TK / 4; i += 256) {
        const int r = i / (TK / 4), c = (i % (TK / 4)) * 4;
        *reinterpret_cast<f32x4*>(base + (size_t)r * K + c) = f32x4{v, v, v, v};
    }
}
__global__ __launch_bounds__(256) void slab_reduce(const f32x4* __restrict__ slabs, f32x4* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = slabs[i];
#pragma unroll 8
    for (int s = 1; s < S; ++s) a += slabs[(size_t)s * n4 + i];
    out[i] = a;
}
__global__ __launch_bounds__(256) void atomic_store(float* __restrict__ out, float v) {
    const int tile = blockIdx.x % 32, tm = tile / 4, tk = tile % 4;
    float* base = out + (size_t)tm * TM * K + tk * TK;
    for (int i = threadIdx.x; i < TM * TK; i += 256) {
        const int r = i / TK, c = i % TK;
        unsafeAtomicAdd(base + (size_t)r * K + c, v);
    }
}
__global__ void zero(f32x4* p, long n4) { const long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n4) p[i] = f32x4{0, 0, 0, 0}; }
int main() {
    float *slabs, *out; const size_t n = (size_t)M * K;
    hipMalloc(&slabs, NSET * S * n * 4); hipMalloc(&out, NSET * n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    const int it = 60;
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(a);
        for (int i = 0; i < it; ++i) {
            float* sl = slabs + (size_t)(i % NSET) * S * n; float* o = out + (size_t)(i % NSET) * n;
            slab_store<<<32 * S, 256>>>(sl, 1.f);
            slab_reduce<<<(n / 4 + 255) / 256, 256>>>((const f32x4*)sl, (f32x4*)o, n / 4);
        }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (pass) printf("slabs: store + reduce      %7.1f us per weight gradient (2 launches)\n", ms * 1000 / it);
        hipEventRecord(a);
        for (int i = 0; i < it; ++i) slab_store<<<32 * S, 256>>>(slabs + (size_t)(i % NSET) * S * n, 1.f);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (pass) printf("slabs: store only          %7.1f us\n", ms * 1000 / it);
        hipEventRecord(a);
        for (int i = 0; i < it; ++i) {
            float* o = out + (size_t)(i % NSET) * n;
            zero<<<(n / 4 + 255) / 256, 256>>>((f32x4*)o, n / 4);
            atomic_store<<<32 * S, 256>>>(o, 1.f);
        }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (pass) printf("atomics: zero + 16-way add %7.1f us per weight gradient (2 launches)\n", ms * 1000 / it);
        hipEventRecord(a);
        for (int i = 0; i < it; ++i) atomic_store<<<32 * S, 256>>>(out + (size_t)(i % NSET) * n, 1.f);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (pass) printf("atomics: 16-way add only   %7.1f us\n", ms * 1000 / it);
    }
    float h[4]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("check: out[0] = %.1f (sum of adds since the last zero)\n", h[0]);
    return 0;
}
