// What access pattern does HBM like?  A "depthwise-shaped" copy (no arithmetic): y[b][c][:] = x[b][c][:] for 1 KB rows of a
// [B][C][512] bf16 tensor, one 256-thread workgroup per channel, wave w takes rows b = w, w + 4, ... with D rows of loads in flight --
// the memory behaviour of dwconv_fwd16_stream_kernel -- against the same bytes laid out channel-major ([C][B][512]: a channel's 32 rows
// are 32 KB contiguous), at 1 / 2 / 4 workgroups per CU resident, plain or nontemporal.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/spp tools/micro/stream_pattern_probe.hip && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int D, bool NT, bool CMAJOR, int MINB>
__global__ __launch_bounds__(256, MINB) void rows_copy(const u32x4* __restrict__ x, u32x4* __restrict__ y, int B, int C, int nper) {
    // persistent over channels: workgroup g handles channels g, g + gridDim.x, ...
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        u32x4 v[D];
        const int nrows = (B - wave + 3) >> 2;
        auto idx = [&](int r) -> size_t {
            const int b = wave + 4 * r;
            return (CMAJOR ? ((size_t)c * B + b) : ((size_t)b * C + c)) * 64 + lane;
        };
#pragma unroll
        for (int d = 0; d < D; ++d) if (d < nrows) v[d] = NT ? __builtin_nontemporal_load(x + idx(d)) : x[idx(d)];
        for (int r0 = 0; r0 < nrows; r0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int r = r0 + d;
                if (r < nrows) {
                    const u32x4 t = v[d];
                    if (r + D < nrows) v[d] = NT ? __builtin_nontemporal_load(x + idx(r + D)) : x[idx(r + D)];
                    if (NT) __builtin_nontemporal_store(t, y + idx(r)); else y[idx(r)] = t;
                }
            }
        }
    }
}

// GEMM-epilogue-shaped stores / X-operand-shaped loads: a 512-thread workgroup owns a [256 rows x 128 columns] bf16 tile of a
// [B][M][512] (batch-major) or [M][B][512] (channel-major) tensor and writes (or reads) it as 16 passes of 16 rows x 256 B (one row
// per half-wave, 8 B per lane) -- what PwEpilogueFull does after its LDS transposition.  Tiles are dealt (b, tt, mt)-major like the
// GEMM's grid.  Answers: would channel-major hidden tensors (which the depthwise kernels like) cost the GEMMs their store rate?
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool CMAJOR, bool LOAD>
__global__ __launch_bounds__(512) void tile_rows(u32x2* __restrict__ y, int B, int M, unsigned* sink) {
    const int ntt = 4, nmt = M / 256;
    const int v = blockIdx.x;
    const int mt = v % nmt, tt = (v / nmt) % ntt, b = v / (nmt * ntt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    unsigned acc = 0;
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
        const int m = mt * 256 + pass * 16 + wave * 2 + half;
        const size_t row = CMAJOR ? ((size_t)m * B + b) : ((size_t)b * M + m);
        u32x2* q = y + (row * 512 + tt * 128) / 4 + col;          // 4 bf16 per u32x2
        if (LOAD) { const u32x2 t = *q; acc += t[0] ^ t[1]; }
        else *q = u32x2{(unsigned)v, (unsigned)pass};
    }
    if (LOAD && acc == 0x12345678u) *sink = acc;
}

int main() {
    const int B = 32, C = 2048, SETS = 10;
    const size_t n16 = (size_t)B * C * 64;          // 16-byte pieces per tensor (64 MB)
    std::vector<u32x4*> xs(SETS), ys(SETS);
    for (int i = 0; i < SETS; ++i) { hipMalloc(&xs[i], n16 * 16); hipMalloc(&ys[i], n16 * 16); hipMemset(xs[i], i + 1, n16 * 16); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int grid) {
        for (int i = 0; i < SETS; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, xs[i], ys[i], B, C, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int iters = 40;
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, xs[i % SETS], ys[i % SETS], B, C, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s grid %5d: %6.1f us  %5.0f GB/s\n", name, grid, ms / iters * 1e3, 2.0 * n16 * 16 / (ms / iters * 1e-3) / 1e9);
    };
    for (int grid : {256, 512, 1024, 2048}) {
        run("[B][C][T] rows, D=4 plain", rows_copy<4, false, false, 1>, grid);
        run("[B][C][T] rows, D=4 nt", rows_copy<4, true, false, 1>, grid);
        run("[C][B][T] rows, D=4 plain", rows_copy<4, false, true, 1>, grid);
        run("[C][B][T] rows, D=4 nt", rows_copy<4, true, true, 1>, grid);
    }
    run("[B][C][T] rows, D=8 nt", rows_copy<8, true, false, 1>, 256);
    run("[C][B][T] rows, D=8 nt", rows_copy<8, true, true, 1>, 256);
    run("[B][C][T] rows, D=1 plain (general kernel)", rows_copy<1, false, false, 1>, 2048);
    run("[B][C][T] rows, D=2 plain", rows_copy<2, false, false, 1>, 2048);
    {
        const int M = 2048;
        unsigned* sink; hipMalloc(&sink, 4);
        auto runt = [&](const char* name, auto kern) {
            const int grid = (M / 256) * 4 * B;
            for (int i = 0; i < SETS; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, (u32x2*)ys[i], B, M, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int iters = 40;
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, (u32x2*)ys[i % SETS], B, M, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-60s: %6.1f us  %5.0f GB/s\n", name, ms / iters * 1e3, 1.0 * n16 * 16 / (ms / iters * 1e-3) / 1e9);
        };
        runt("GEMM-epilogue stores, [B][M][T] (batch-major) 67 MB", tile_rows<false, false>);
        runt("GEMM-epilogue stores, [M][B][T] (channel-major)", tile_rows<true, false>);
        runt("tile-row loads,       [B][M][T]", tile_rows<false, true>);
        runt("tile-row loads,       [M][B][T]", tile_rows<true, true>);
    }
    return 0;
}
