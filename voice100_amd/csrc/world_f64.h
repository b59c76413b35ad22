// fp64 workgroup helpers shared by the WORLD analysis kernels (world_analysis.hip) and the general-size WORLD synthesis kernel
// (world.hip): complex type, workgroup sum / exclusive scan / cumulative sum, and an in-place radix-2 FFT on an LDS array.
#pragma once
#include <hip/hip_runtime.h>

namespace {
#pragma clang fp contract(off)
struct cd { double x, y; };

__device__ inline double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// sum over the workgroup, returned to every thread; red: >= 16 doubles of LDS
__device__ inline double block_sum(double v, double* red) {
    const int nw = blockDim.x >> 6;
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
// exclusive scan of one value per thread over the workgroup; *total = sum.  red: >= 16 values of LDS
template <typename V>
__device__ inline V block_excl_scan(V v, V* red, V* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    V inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const V o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) red[w] = inc;
    __syncthreads();
    V base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) {
        if (i < w) base += red[i];
        tot += red[i];
    }
    *total = tot;
    return base + inc - v;
}
// in-place cumulative sum of a[0 .. n) in LDS: every thread sums a contiguous chunk, a workgroup scan joins the chunks
__device__ inline void cumsum_lds(double* a, int n, double* red) {
    const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x, i0 = (int)threadIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    __syncthreads();
    double s = 0.0;
    for (int i = i0; i < i1; ++i) s += a[i];
    double tot;
    double acc = block_excl_scan<double>(s, red, &tot);
    for (int i = i0; i < i1; ++i) { acc += a[i]; a[i] = acc; }
    __syncthreads();
}

// in-place radix-2 FFT of a[0 .. 2^logN) whose elements were STORED bit-reversed; tw[k] = exp(-2 pi i k / 2^logNT); the caller syncs before
// Stages with butterfly spans up to 128 elements stay inside the 128-element block a wave's 64 consecutive butterflies cover (for every
// blockDim that is a multiple of 64), so they need no workgroup barrier: a wave's LDS operations execute in order, only the compiler
// must not move them (WAVE_LDS_ORDER).  The later stages exchange across waves.
#define WAVE_LDS_ORDER() asm volatile("" ::: "memory")
__device__ void fft_lds(cd* a, int logN, const cd* __restrict__ tw, int logNT, bool inverse) {
    const int n2 = 1 << (logN - 1);
    for (int s = 1; s <= logN; ++s) {
        const int h = 1 << (s - 1);
        for (int idx = threadIdx.x; idx < n2; idx += blockDim.x) {
            const int k = idx & (h - 1), i = ((idx >> (s - 1)) << s) + k;
            cd w = tw[k << (logNT - s)];
            if (inverse) w.y = -w.y;
            const cd u = a[i], v = a[i + h];
            const double tr = v.x * w.x - v.y * w.y, ti = v.x * w.y + v.y * w.x;
            a[i] = {u.x + tr, u.y + ti};
            a[i + h] = {u.x - tr, u.y - ti};
        }
        if (s < 7 && s < logN) WAVE_LDS_ORDER();
        else __syncthreads();
    }
}
__device__ inline unsigned brev(unsigned i, int logN) { return __brev(i) >> (32 - logN); }


#pragma clang fp contract(fast)
}   // namespace
