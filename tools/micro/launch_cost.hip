// Host cost of a kernel launch on this stack: hipLaunchKernelGGL of an empty kernel with a 16-byte and a 256-byte argument block,
// 20000 launches back to back (the GPU drains them as fast as it can), wall time per launch on the host side.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float v[64]; };
struct Small { float v[4]; };
template <class P> __global__ void k(P p, float* out) { if (p.v[0] == 12345.f) out[0] = p.v[1]; }
template <class P> static void run(const char* name) {
    float* out; hipMalloc(&out, 4);
    P p{}; 
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, p, out);
    hipDeviceSynchronize();
    const int n = 20000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, p, out);
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("%s: host %.2f us per launch, with drain %.2f us per launch\n", name,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    hipFree(out);
}
int main() { run<Small>("16-byte args "); run<Big>("256-byte args"); return 0; }
