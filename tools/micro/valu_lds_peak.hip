// Ceiling probe for the depthwise inner loop on gfx950: fp32 v_fmac issue rate with 0 / 1 / 2 ds_read_b128 per 32 FMAs,
// at a given occupancy.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_lds_peak.hip -o build/valu_lds_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NLDS, int PAD>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    __shared__ float pad[PAD > 0 ? PAD : 1];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1e-9f * i;
    if (PAD > 1 && threadIdx.x == 0) pad[PAD - 1] = 0.f;
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = threadIdx.x * 1e-3f + r;
    f32x4 a = {1.0001f, 0.9999f, 1.0002f, 0.9998f};
    f32x4 w = {1e-6f, 2e-6f, 3e-6f, 4e-6f};
    const float* pw = lds + (threadIdx.x & 63) * 8;
    const float* pb = lds + 2048;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (NLDS >= 1) a = *reinterpret_cast<const f32x4*>(pw + 4 * ((it + u) & 255));
            if (NLDS >= 2) w = *reinterpret_cast<const f32x4*>(pb + 4 * ((it + u) & 255));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = fmaf(w[(e + r) & 3], a[e], acc[r]);
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(acc[r]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += acc[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + (PAD > 1 ? pad[PAD - 1] : 0.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

// Software-pipelined like the depthwise kernel: chunk u+1's window (and tap) reads are issued before chunk u's FMAs.
// R = outputs per lane: 4*R FMAs per chunk; TAPS = 1 streams a tap chunk next to every window chunk.
template <int R, int TAPS, int LSTRIDE = 8>
__global__ __launch_bounds__(256) void probe_pipe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1e-9f * i;
    __syncthreads();
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = threadIdx.x * 1e-3f + r;
    // lane stride of the window reads in floats: 8 = the depthwise kernel's R (2-way bank conflict inside the hardware's
    // 16-lane ds_read_b128 groups), 12 = padded (conflict-free)
    const float* pw = lds + (threadIdx.x & 63) * LSTRIDE;
    const float* pb = lds + 2048;
    f32x4 a = *reinterpret_cast<const f32x4*>(pw), w = *reinterpret_cast<const f32x4*>(pb);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 an = *reinterpret_cast<const f32x4*>(pw + 4 * ((it + u + 1) & 127));
            f32x4 wn = w;
            if (TAPS) wn = *reinterpret_cast<const f32x4*>(pb + 4 * ((it + u + 1) & 255));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fmaf(w[(e + r) & 3], a[e], acc[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[r]));
            __builtin_amdgcn_sched_barrier(0);
            a = an; w = wn;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int R, int TAPS, int LSTRIDE = 8>
static void run_pipe(const char* name, int wgs_per_cu, float* out) {
    const int iters = 4000 * 8 / R, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_pipe<R, TAPS, LSTRIDE>), dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe_pipe<R, TAPS, LSTRIDE>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 256 * 8 * 4 * R * (double)iters * grid;
    printf("pipelined %-34s wgs/cu=%d  %8.1f us  %7.1f TFLOP/s\n", name, wgs_per_cu, ms * 1e3, flop / (ms * 1e-3) / 1e12);
}

// bf16 dot2 variant of the same pipelined loop: one v_dot2c_f32_bf16 = two MACs per lane (bf16 pairs, fp32 accumulate).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <int R, int TAPS>
__global__ __launch_bounds__(256) void probe_dot2(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1e-9f * i;
    __syncthreads();
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = threadIdx.x * 1e-3f + r;
    const float* pw = lds + (threadIdx.x & 63) * 8;
    const float* pb = lds + 2048;
    f32x4 a = *reinterpret_cast<const f32x4*>(pw), w = *reinterpret_cast<const f32x4*>(pb);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 an = *reinterpret_cast<const f32x4*>(pw + 4 * ((it + u + 1) & 255));
            f32x4 wn = w;
            if (TAPS) wn = *reinterpret_cast<const f32x4*>(pb + 4 * ((it + u + 1) & 255));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    acc[r] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w[(e + r) & 3]), __builtin_bit_cast(bf16x2_t, a[e]), acc[r], false);
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[r]));
            __builtin_amdgcn_sched_barrier(0);
            a = an; w = wn;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int R, int TAPS>
static void run_dot2(const char* name, int wgs_per_cu, float* out) {
    const int iters = 4000 * 8 / R, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_dot2<R, TAPS>), dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe_dot2<R, TAPS>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double macs = 2.0 * 256 * 8 * 4 * R * (double)iters * grid;     // 2 MACs per dot2
    printf("dot2 bf16 %-34s wgs/cu=%d  %8.1f us  %7.1f TMAC-FLOP/s (2 flop per MAC)\n", name, wgs_per_cu, ms * 1e3, 2.0 * macs / (ms * 1e-3) / 1e12);
}

template <int NLDS, int PAD>
static void run(const char* name, int wgs_per_cu, float* out, long long* cyc) {
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NLDS, PAD>), dim3(grid), dim3(256), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NLDS, PAD>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = 2.0 * 256 * 8 * 32 * (double)iters * grid;
    printf("%-28s wgs/cu=%d  %8.1f us  %7.1f TFLOP/s   s_memtime ticks/iter=%.1f  (ticks/us=%.0f)\n", name, wgs_per_cu, ms * 1e3,
           flop / (ms * 1e-3) / 1e12, (double)c / iters, (double)c / (ms * 1e3));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 8);
    for (int w : {1, 2, 3, 4, 8}) {
        if (w == 1) { run<0, 1>("fma only", 1, out, cyc); run<1, 1>("fma + 1 ds_read/32", 1, out, cyc); run<2, 1>("fma + 2 ds_read/32", 1, out, cyc); }
        if (w == 2) { run<0, 1>("fma only", 2, out, cyc); run<1, 1>("fma + 1 ds_read/32", 2, out, cyc); run<2, 1>("fma + 2 ds_read/32", 2, out, cyc); }
        if (w == 3) { run<0, 1>("fma only", 3, out, cyc); run<1, 1>("fma + 1 ds_read/32", 3, out, cyc); run<2, 1>("fma + 2 ds_read/32", 3, out, cyc); }
        if (w == 4) { run<0, 1>("fma only", 4, out, cyc); run<1, 1>("fma + 1 ds_read/32", 4, out, cyc); run<2, 1>("fma + 2 ds_read/32", 4, out, cyc); }
        if (w == 8) { run<0, 1>("fma only", 8, out, cyc); run<1, 1>("fma + 1 ds_read/32", 8, out, cyc); run<2, 1>("fma + 2 ds_read/32", 8, out, cyc); }
    }
    for (int w : {3, 4, 6, 8}) {
        run_pipe<8, 1>("R=8  window+taps (2 reads/32 FMA)", w, out);
        run_pipe<8, 1, 12>("R=8  window+taps, lane stride 12", w, out);
        run_pipe<8, 0, 12>("R=8  window only, lane stride 12", w, out);
        run_pipe<8, 0>("R=8  window only (1 read/32 FMA)", w, out);
        run_pipe<16, 1>("R=16 window+taps (2 reads/64 FMA)", w, out);
        run_pipe<16, 0>("R=16 window only (1 read/64 FMA)", w, out);
        run_pipe<32, 1>("R=32 window+taps (2 reads/128 FMA)", w, out);
    }
    for (int w : {3, 4, 8}) {
        run_dot2<8, 1>("R=8  window+taps (2 reads/32 dot2)", w, out);
        run_dot2<8, 0>("R=8  window only", w, out);
        run_dot2<16, 1>("R=16 window+taps", w, out);
    }
    return 0;
}
