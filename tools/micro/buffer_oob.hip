// Probe: how gfx950 range-checks a 16-byte raw buffer load that is only partly inside the descriptor.
// hipcc --offload-arch=gfx950 -O2 -o buffer_oob tools/micro/buffer_oob.hip && ./buffer_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* x, int n, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    // lane 0: straddles the end; lane 1: negative voffset; lane 2: negative via soffset; lane 3: fully inside, unaligned
    int vo = 0, so = 0;
    if (lane == 0) vo = (n - 2) * 4;
    if (lane == 1) vo = -8;
    if (lane == 2) { vo = 8; so = 0; }
    if (lane == 3) vo = 4;
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, lane == 2 ? -16 : 0, 0));
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = v[i];
}
int main() {
    const int n = 64;
    float h[n], *d, *o, ho[16];
    for (int i = 0; i < n; ++i) h[i] = 100.f + i;
    hipMalloc(&d, n * 4 + 4096); hipMalloc(&o, 64);
    hipMemset(d, 0x7f, n * 4 + 4096);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(4), 0, 0, d, n, o);
    hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost);
    const char* name[4] = {"straddle end (n-2)", "voffset -8", "voffset 8 + soffset -16", "inside, unaligned +4"};
    for (int l = 0; l < 4; ++l) printf("%-26s: %g %g %g %g\n", name[l], ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3]);
    return 0;
}
