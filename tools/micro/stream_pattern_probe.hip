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
    return 0;
}
