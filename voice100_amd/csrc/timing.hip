// Opt-in HIP-event timing of the hot kernels, for bench.py's roofline line: events are recorded on the
// stream the kernel is launched on, inside the timed region.  An event pair costs ~1 us of host time and ~3 us
// of queue time per launch (with every hot kernel tagged: ~0.8 ms of an 8 ms step), so the caller picks the tags it needs (torch.cuda.Event
// pairs from Python cost ~10 us each and distorted the step).  Off by default; no effect on results.
#include "common.h"
#include "timing.h"
#include "depthwise_common.h"     // v100_chan_of_block
#include <vector>
#include <mutex>
#include <atomic>
#include <stdlib.h>

namespace {
struct Pair { hipEvent_t a, b; int tag; double bytes; };
std::mutex g_mu;
unsigned g_mask = 0;          // bit t set: launches tagged t are timed
std::vector<Pair> g_pairs;
size_t g_used = 0;
const size_t kMaxPairs = 16384;
thread_local int g_region_tag = -1;
thread_local double g_region_bytes = 0.0;
thread_local int g_in_region = 0;   // > 0: this thread is inside a V100TimedRegion that is being timed (its launches belong to that tag, not to "other")
}

// A timed REGION (the 1x1-GEMM entry points: one or two launches each): since round 5 its launches are timed like every other one, by
// the dispatch packets' own timestamps -- V100_GGL asks v100_timing_other() for an event pair, which is booked under the region's tag --
// instead of two hipEventRecord markers around the region (~3 us per region on top of the kernels, and a queue drain per marker: the
// family figures read 0.2 ms per step above the kernel trace's).  The region's algorithmic bytes go to its first launch.
void v100_timing_begin(int tag, hipStream_t st, int* slot, double bytes) {
    (void)st;
    *slot = -1;
    if (!((g_mask >> tag) & 1u)) return;
    g_region_tag = tag;
    g_region_bytes = bytes;
    *slot = 0;
    ++g_in_region;
}

V100TimedLaunch::V100TimedLaunch(int tag, double bytes) {
    if (!((g_mask >> tag) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used >= kMaxPairs) return;
    if (g_used >= g_pairs.size()) {
        Pair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
        g_pairs.push_back(p);
    }
    g_pairs[g_used].tag = tag;
    g_pairs[g_used].bytes = bytes;
    a = g_pairs[g_used].a;
    b = g_pairs[g_used].b;
    ++g_used;
}

// every launch that goes through plain V100_GGL (common.h): timed under V100_T_OTHER when that bit is on
std::atomic<unsigned> g_other_on{0};
extern "C" int v100_timing_other(void** a, void** b) {
    const bool in_region = g_in_region > 0;
    if (!in_region && !g_other_on.load(std::memory_order_relaxed)) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    const int tag = in_region ? g_region_tag : V100_T_OTHER;
    if (!((g_mask >> tag) & 1u) || g_used >= kMaxPairs) return 0;
    if (g_used >= g_pairs.size()) {
        Pair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return 0;
        g_pairs.push_back(p);
    }
    g_pairs[g_used].tag = tag;
    g_pairs[g_used].bytes = in_region ? g_region_bytes : 0.0;
    if (in_region) g_region_bytes = 0.0;
    *a = g_pairs[g_used].a;
    *b = g_pairs[g_used].b;
    ++g_used;
    return 1;
}

void v100_timing_end(int slot, hipStream_t st) {
    (void)st;
    if (slot < 0) return;
    --g_in_region;
}

extern "C" int v100_timing_enable(int tag_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = (unsigned)tag_mask;
    g_other_on.store(((unsigned)tag_mask >> V100_T_OTHER) & 1u, std::memory_order_relaxed);
    if (tag_mask) g_used = 0;
    return V100_OK;
}

// Sum of elapsed ms, number of launches and algorithmic bytes recorded under `tag` since the last enable(mask != 0).
// Synchronises the device.
extern "C" int v100_timing_read(int tag, double* ms, long long* count, double* bytes) {
    if (!ms || !count || !bytes) return V100_ERR_NULL;
    if (hipDeviceSynchronize() != hipSuccess) return V100_ERR_LAUNCH;
    std::lock_guard<std::mutex> lk(g_mu);
    double total = 0.0, nbytes = 0.0;
    long long n = 0;
    for (size_t i = 0; i < g_used; ++i) {
        if (g_pairs[i].tag != tag) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_pairs[i].a, g_pairs[i].b) == hipSuccess) { total += t; ++n; nbytes += g_pairs[i].bytes; }
    }
    *ms = total;
    *count = n;
    *bytes = nbytes;
    return V100_OK;
}

// kernel launches issued by this library in this process so far (every launch site goes through V100_GGL / V100_EXT_GGL)
extern "C" long long v100_launch_count(void) { return g_v100_launches.load(std::memory_order_relaxed); }

// Device-copy yardstick for the HBM-bound kernels (bench.py `copy_probe_gbs`): dst[i] = src[i] in 16-byte pieces, U independent
// loads in flight per thread before the first store, workgroups of 256 threads walking the buffer grid-stride.
// n16 = number of 16-byte pieces.  Same ABI conventions as every other entry point (device pointers, no sync).
// V100_COPY_VARIANT (tuning only): digit 1 = unroll (1: 4, 2: 8, 3: 2), digit 2 = nontemporal (0 plain, 1 nt), digit 3 = workgroups per CU.
template <int U, bool NT_>
__global__ __launch_bounds__(256) void copy_probe_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n16) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT_ ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT_) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
extern "C" int v100_copy_probe(const void* src, void* dst, long long nbytes, void* stream) {
    if (!src || !dst) return V100_ERR_NULL;
    if (nbytes <= 0 || (nbytes & 15)) return V100_ERR_SHAPE;
    // default 111 = one workgroup per CU, nontemporal, 4 pieces in flight per lane: the fastest of profiles/r03_copy_probe_variants.txt
    static const int variant = [] { const char* e = getenv("V100_COPY_VARIANT"); return e ? atoi(e) : 111; }();
    const int un = variant % 10, nt = (variant / 10) % 10, wpc = variant / 100 > 0 ? variant / 100 : 8;
    const dim3 grid(256 * wpc), block(256);
    const f32x4* s = (const f32x4*)src;
    f32x4* d = (f32x4*)dst;
    const long long n16 = nbytes / 16;
    hipStream_t st = (hipStream_t)stream;
#define CP(U_, N_) V100_GGL((copy_probe_kernel<U_, N_>), grid, block, 0, st, s, d, n16)
    if (un == 2) { if (nt) CP(8, true); else CP(8, false); }
    else if (un == 3) { if (nt) CP(2, true); else CP(2, false); }
    else { if (nt) CP(4, true); else CP(4, false); }
#undef CP
    return v100_launch_status();
}

// The depthwise streaming kernels' ACCESS PATTERN as a pure copy (no arithmetic): y[b][c][:] = x[b][c][:] for the P-sample bf16 rows of
// a [B][C][P] tensor, one 256-thread workgroup per channel, wave w takes rows b = w, w + 4, ..., one row of loads in flight ahead of
// the stores, nontemporal -- what dwconv_fwd16_stream_kernel does to memory (tools/micro/stream_pattern_probe.hip, D = 1 nt).  bench.py
// times it on a rotating working set to report the [B][C][T] one-row-per-wave pattern's own ceiling next to the 8 TB/s nominal rate
// (round-4 review, item 3: the kernel is graded against what its layout can reach).  P * 2 bytes must be a multiple of 1024 here.
__global__ __launch_bounds__(256) void rows_copy_probe_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int B, int C, int P16B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = v100_chan_of_block<16>(blockIdx.x, C);      // the streaming kernels' own blockIdx -> channel order (round 6)
    const int per = P16B / 1024;                        // 1 KB pieces per row (64 lanes x 16 B)
    const int nrows = (B - wave + 3) >> 2;
    for (int q = 0; q < per; ++q) {
        auto idx = [&](int r) -> size_t { return ((size_t)(wave + 4 * r) * C + c) * (size_t)(P16B / 16) + q * 64 + lane; };
        f32x4 v = nrows > 0 ? __builtin_nontemporal_load(x + idx(0)) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < nrows; ++r) {
            const f32x4 t = v;
            if (r + 1 < nrows) v = __builtin_nontemporal_load(x + idx(r + 1));
            __builtin_nontemporal_store(t, y + idx(r));
        }
    }
}
extern "C" int v100_rows_copy_probe(const void* src, void* dst, int B, int C, int row_bytes, void* stream) {
    if (!src || !dst) return V100_ERR_NULL;
    if (B <= 0 || C <= 0 || row_bytes <= 0 || (row_bytes & 1023)) return V100_ERR_SHAPE;
    V100_GGL(rows_copy_probe_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, B, C, row_bytes);
    return v100_launch_status();
}
