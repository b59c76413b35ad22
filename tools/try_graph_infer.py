#!/usr/bin/env python3
"""Eval-mode forwards replayed as a HIP graph (voice100_amd.infer.GraphedForward) against eager calls: latency at the small, launch-bound
shapes (configs[0]: B = 2 x 256 frames) and at the larger ones, results bit-identical.   python tools/try_graph_infer.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    from voice100_amd import functional as F_
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel
    from voice100_amd.infer import GraphedForward
    dev = torch.device("cuda")
    torch.manual_seed(0)
    for prec in ("bf16", "fp32"):
        F_.set_matmul_precision(prec)
        m = AudioToTextCTC(64, 512, 29, 512).to(dev).eval()
        for B, T in ((2, 256), (32, 101), (1, 100), (8, 512), (32, 1024)):
            x = torch.rand(B, T, 64, device=dev)
            with torch.no_grad():
                ref = m(x).clone()
                te = timeit(lambda: m(x))
                torch.cuda.synchronize()
                h0 = time.perf_counter()
                for _ in range(50):
                    m(x)
                th = (time.perf_counter() - h0) / 50          # host enqueue alone (the queue is empty at the start, 50 calls deep at the end)
                torch.cuda.synchronize()
            g = GraphedForward(m, x)
            out = g(x)
            same = torch.equal(out, ref)
            x2 = torch.rand(B, T, 64, device=dev)
            with torch.no_grad():
                ref2 = m(x2).clone()
            same2 = torch.equal(g(x2), ref2)
            tg = timeit(lambda: g(x))
            print(f"asr eval {prec} B={B} T={T}: eager {te*1e3:.3f} ms (host enqueue {th*1e3:.3f}), graph {tg*1e3:.3f} ms, identical {same and same2}")
        t = AlignTextToAudioModel(vocab_size=29, hidden_size=512, use_mcep=True).to(dev).eval()
        for B, L in ((1, 64), (16, 512)):
            at = torch.randint(0, 29, (B, L), device=dev)
            with torch.no_grad():
                ref = [r.clone() for r in t.predict(at)]
                te = timeit(lambda: t.predict(at))
            g = GraphedForward(t.predict, at)
            out = g(at)
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            tg = timeit(lambda: g(at))
            print(f"tts predict {prec} B={B} L={L}: eager {te*1e3:.3f} ms, graph {tg*1e3:.3f} ms, identical {same}")
    F_.set_matmul_precision("fp32")


if __name__ == "__main__":
    main()
