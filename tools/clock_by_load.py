#!/usr/bin/env python3
"""Clock and power the chip settles at under different sustained loads (is the training step's 2.1-2.3 GHz a property of the chip under
ANY load, or of what the load does?): each load runs back to back for ~8 s while pp_dpm_sclk / hwmon power are sampled (bench.gpu_clock_power).
  python3 tools/clock_by_load.py"""
import os, sys, time, threading, torch
sys.path.insert(0, os.getcwd())
import bench
from voice100_amd import functional as F_, _native as N
dev = torch.device("cuda:0"); N.load()

def sample_while(fn, seconds=8.0):
    stop, samples = [False], []
    def watch():
        while not stop[0]:
            s = bench.gpu_clock_power()
            if s.get("sclk_mhz"): samples.append((s["sclk_mhz"], s.get("power_w")))
            time.sleep(0.25)
    th = threading.Thread(target=watch); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    stop[0] = True; th.join()
    tail = samples[len(samples) // 2:]                       # the settled half
    clk = sum(c for c, _ in tail) / max(len(tail), 1)
    pw = [p for _, p in tail if p]
    return clk, (sum(pw) / len(pw) if pw else float("nan")), (time.time() - t0) / n * 1e6

g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(8192, 8192, device=dev, generator=g).bfloat16(); Bm = torch.randn(8192, 8192, device=dev, generator=g).bfloat16()
W = torch.randn(2048, 512, device=dev, generator=g).bfloat16(); X = torch.randn(512, 16384, device=dev, generator=g).bfloat16()
big = torch.empty(1 << 28, device=dev); big2 = torch.empty(1 << 28, device=dev)
loads = [("vendor GEMM 8192^3 bf16 (matrix-pipe bound)", lambda: torch.matmul(A, Bm)),
         ("vendor GEMM 2048 x 512 x 16384 bf16 (the step's short-K shape)", lambda: torch.matmul(W, X)),
         ("1 GiB device copy (HBM bound)", lambda: big2.copy_(big))]
# the library's depthwise forward on the bench shape (HBM bound, MFMA Toeplitz)
B, C, T, K = 32, 2048, 512, 83
P = F_.pitch16(T, B)
a1 = torch.randn(B, C, P, device=dev, generator=g).bfloat16(); a2 = torch.empty_like(a1)
wd = torch.randn(C, K, device=dev, generator=g); s1 = torch.rand(C, device=dev, generator=g) + 0.5; t1 = torch.randn(C, device=dev, generator=g)
st = torch.empty(N.helper("v100_dw_num_groups", B, C), C, 2, device=dev)
loads.append(("library depthwise forward C = 2048, k = 83 (HBM bound, Toeplitz MFMA)",
              lambda: N.call("v100_dwconv_fwd_train_io", a1, wd, s1, t1, a2, st, N.helper("v100_dw_num_groups", B, C), B, C, T, K, 1 | 8)))
print("idle:", bench.gpu_clock_power())
for name, fn in loads:
    fn(); torch.cuda.synchronize()
    clk, pw, us = sample_while(fn)
    print(f"{name:75s} {clk:7.0f} MHz {pw:7.0f} W   {us:9.1f} us per call")
