#pragma once
#include <hip/hip_runtime.h>
enum { V100_T_DW_FWD = 0, V100_T_DW_BWD_DATA = 1, V100_T_DW_WGRAD = 2, V100_T_PW_GEMM = 3, V100_T_PW_WGRAD = 4 };
void v100_timing_begin(int tag, hipStream_t st, int* slot, double bytes);
void v100_timing_end(int slot, hipStream_t st);
struct V100TimedRegion {
    int slot; hipStream_t st;
    V100TimedRegion(int tag, hipStream_t s, double bytes = 0.0) : st(s) { v100_timing_begin(tag, s, &slot, bytes); }
    ~V100TimedRegion() { v100_timing_end(slot, st); }
};
