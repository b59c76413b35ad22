#!/usr/bin/env python3
"""Inference throughput of the other BASELINE.json configs on one GPU (not the driver's headline bench):
  config 1 (GPU): asr_en_base eval forward, B=2 x 256 frames (the reference's CPU-runnable case)
  config 3: tts_en_base  aligntext [16, 512] -> TextToAlign-free audio model -> WORLD features (predict)
  config 5: streaming ASR, 1-second 16 kHz chunks -> log-mel -> encoder -> logits -> greedy CTC decode
python tools/bench_infer.py [--precision bf16|fp32] [--iters 50]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voice100_amd import functional as F_
from voice100_amd.asr import AudioToTextCTC
from voice100_amd.tts import AlignTextToAudioModel
from voice100_amd.mel import MelSpectrogramAudioTransform
from voice100_amd.decode import ctc_greedy_decode


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    F_.set_matmul_precision(args.precision)
    torch.manual_seed(1234)
    out = {"precision": args.precision}
    with torch.no_grad():
        asr = AudioToTextCTC(64, 512, 29, 512).to(dev).eval()
        x = torch.rand(2, 256, 64, device=dev)
        dt = timeit(lambda: asr(x), args.iters)
        out["config1_asr_eval_B2_T256"] = {"ms": round(dt * 1e3, 3), "frames_per_s": round(2 * 256 / dt, 1)}
        x = torch.randn(32, 1024, 64, device=dev)
        dt = timeit(lambda: asr(x), args.iters)
        out["asr_eval_B32_T1024"] = {"ms": round(dt * 1e3, 3), "frames_per_s": round(32 * 1024 / dt, 1)}

        tts = AlignTextToAudioModel(vocab_size=29, hidden_size=512, use_mcep=False).to(dev).eval()
        at = torch.randint(0, 29, (16, 512), device=dev)
        dt = timeit(lambda: tts.predict(at), args.iters)
        out["config3_tts_predict_B16_L512"] = {"ms": round(dt * 1e3, 3), "aligntext_frames_per_s": round(16 * 512 / dt, 1),
                                               "world_frames_per_s": round(16 * 1023 / dt, 1)}

        mel = MelSpectrogramAudioTransform().to(dev)
        for B in (32, 256):
            wav = torch.rand(B, 16000, device=dev) * 2 - 1

            def stream():
                feats = mel(wav)                       # [B, 101, 64]
                ids, n = ctc_greedy_decode(asr(feats))
                return ids

            dt = timeit(stream, args.iters)
            out[f"config5_stream_1s_chunks_B{B}"] = {"ms": round(dt * 1e3, 3), "chunks_per_s": round(B / dt, 1),
                                                     "frames_per_s": round(B * 101 / dt, 1), "x_realtime": round(B / dt, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
