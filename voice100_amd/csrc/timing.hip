// Opt-in HIP-event timing of the hot kernels, for bench.py's roofline line: events are recorded on the
// stream the kernel is launched on, inside the timed region.  An event pair costs ~1 us of host time and ~3 us
// of queue time per launch (with every hot kernel tagged: ~0.8 ms of an 8 ms step), so the caller picks the tags it needs (torch.cuda.Event
// pairs from Python cost ~10 us each and distorted the step).  Off by default; no effect on results.
#include "common.h"
#include "timing.h"
#include <vector>
#include <mutex>
#include <stdlib.h>

namespace {
struct Pair { hipEvent_t a, b; int tag; double bytes; };
std::mutex g_mu;
unsigned g_mask = 0;          // bit t set: launches tagged t are timed
std::vector<Pair> g_pairs;
size_t g_used = 0;
const size_t kMaxPairs = 16384;
}

void v100_timing_begin(int tag, hipStream_t st, int* slot, double bytes) {
    *slot = -1;
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
    (void)hipEventRecord(g_pairs[g_used].a, st);
    *slot = (int)g_used++;
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

void v100_timing_end(int slot, hipStream_t st) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_pairs[slot].b, st);
}

extern "C" int v100_timing_enable(int tag_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = (unsigned)tag_mask;
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
