#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "common.h"
enum { V100_T_DW_FWD = 0, V100_T_DW_BWD_DATA = 1, V100_T_DW_WGRAD = 2, V100_T_PW_GEMM = 3, V100_T_PW_WGRAD = 4, V100_T_OTHER = 5 };
void v100_timing_begin(int tag, hipStream_t st, int* slot, double bytes);
void v100_timing_end(int slot, hipStream_t st);
// A launch timed by the dispatch packet's own start/stop timestamps (hipExtLaunchKernelGGL): no marker packets in the queue,
// so the event pair reads the kernel's duration as rocprofv3 does (a region bracketed by two hipEventRecord reads ~3 us more).
// a == nullptr: timing of this tag is off, launch normally.
struct V100TimedLaunch {
    hipEvent_t a = nullptr, b = nullptr;
    V100TimedLaunch(int tag, double bytes = 0.0);
};
#define V100_LAUNCH(TL, kernel, grid, block, shmem, st, ...)                                                   \
    do {                                                                                                       \
        if ((TL).a) V100_EXT_GGL(kernel, grid, block, shmem, st, (TL).a, (TL).b, 0, __VA_ARGS__);     \
        else V100_GGL(kernel, grid, block, shmem, st, __VA_ARGS__);                                  \
    } while (0)

struct V100TimedRegion {
    int slot; hipStream_t st;
    V100TimedRegion(int tag, hipStream_t s, double bytes = 0.0) : st(s) { v100_timing_begin(tag, s, &slot, bytes); }
    ~V100TimedRegion() { v100_timing_end(slot, st); }
};
