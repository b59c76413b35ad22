// Cost of a kernel launch on this stack, and whether the size of the argument block is on the GPU's critical path:
//  (1) empty kernels back to back (20000 launches): time per launch with 16-, 256- and 408-byte argument blocks (host- or
//      dispatch-bound, whichever is slower);
//  (2) kernels that each keep the GPU busy ~10 us (the host runs far ahead): time per launch minus the kernel's own duration is what
//      the dispatch of a DEPENDENT kernel adds -- compared across the argument sizes.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Args { float v[N]; };
template <class P> __global__ void k(P p, float* out, int spin) {
    long long t0 = wall_clock64();
    while (spin && wall_clock64() - t0 < spin) {}
    if (p.v[0] == 12345.f) out[0] = p.v[1];
}
template <class P> static void run(const char* name, int spin, int n) {
    float* out; hipMalloc(&out, 4);
    P p{};
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, p, out, spin);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, p, out, spin);
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("%s spin %5d ticks: host %.2f us per launch, with drain %.2f us per launch\n", name, spin,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    hipFree(out);
}
int main() {
    run<Args<4>>("16-byte args ", 0, 20000); run<Args<64>>("256-byte args", 0, 20000); run<Args<102>>("408-byte args", 0, 20000);
    // wall_clock64 ticks at 100 MHz: 1000 ticks = 10 us
    run<Args<4>>("16-byte args ", 1000, 3000); run<Args<64>>("256-byte args", 1000, 3000); run<Args<102>>("408-byte args", 1000, 3000);
    return 0;
}
