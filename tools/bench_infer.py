#!/usr/bin/env python3
"""Inference throughput of the other BASELINE.json configs (not the driver's headline bench):
  config 1 (GPU): asr_en_base eval forward, B=2 x 256 frames (the reference's CPU-runnable case)
  config 3: tts_en_base  -- audio model alone (aligntext [16, 512] -> WORLD features, `predict`), and the whole chain
            text [16, 128] -> align model -> align() -> audio model -> mcep -> logspc -> spc on the device (infer.TTSPipeline)
  config 5: streaming ASR, 1-second 16 kHz chunks -> log-mel -> encoder -> logits -> greedy CTC decode

    python tools/bench_infer.py [--precision bf16|fp16|fp32] [--iters 50] [--gpus N]

--gpus N (N > 1): pure data-parallel scatter, one process per GPU (SURVEY.md 8e "Inference", BASELINE configs[4]): this
process starts the N ranks itself (a child `python -m torch.distributed.run`, before anything here touches the GPU), every
rank holds a replica (same seed) and takes its share of each request batch / every N-th chunk (voice100_amd.infer.scatter_run);
no collective on the data path, results gathered on rank 0.  Times are barrier-to-barrier, max over ranks; throughput is the
whole job's.  The run fails if the process group's size differs from --gpus."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=256, help="1-second chunks per request batch (whole job)")
    args = ap.parse_args()

    from voice100_amd.trainer import init_distributed, launch_ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    from voice100_amd import functional as F_
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel, TextToAlignTextModel
    from voice100_amd.mel import MelSpectrogramAudioTransform
    from voice100_amd.vocoder import WORLDVocoder
    from voice100_amd.decode import ctc_greedy_decode
    from voice100_amd.infer import ASRPipeline, TTSPipeline, scatter_run

    rank, local_rank, world = init_distributed()
    if dist.is_initialized():
        world = dist.get_world_size()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world} rank(s)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    F_.set_matmul_precision(args.precision)
    torch.manual_seed(1234)                      # every rank: the same replica

    def timeit(fn, iters):
        for _ in range(5):
            fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = (time.perf_counter() - t0) / iters
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    out = {"precision": args.precision, "n_gpus": world}
    with torch.no_grad():
        asr = AudioToTextCTC(64, 512, 29, 512).to(dev).eval()
        tts = AlignTextToAudioModel(vocab_size=29, hidden_size=512, use_mcep=False).to(dev).eval()
        tts_mcep = AlignTextToAudioModel(vocab_size=29, hidden_size=512, use_mcep=True).to(dev).eval()
        align_model = TextToAlignTextModel(vocab_size=29, hidden_size=512).to(dev).eval()
        # an untrained head predicts ~0 frames per token: bias it to gap ~ 1, length ~ 3 so 128 tokens become ~500 aligned frames
        align_model.layers[4].bias.copy_(torch.tensor([0.6931, 1.3863], device=dev))
        mel = MelSpectrogramAudioTransform().to(dev)
        voc = WORLDVocoder(use_mcep=True).to(dev)
        chain = TTSPipeline(align_model, tts_mcep, voc)
        stream = ASRPipeline(asr, mel)
        g = torch.Generator().manual_seed(7)
        if world == 1:
            x = torch.rand(2, 256, 64, device=dev)
            dt = timeit(lambda: asr(x), args.iters)
            out["config1_asr_eval_B2_T256"] = {"ms": round(dt * 1e3, 3), "frames_per_s": round(2 * 256 / dt, 1)}
            x = torch.randn(32, 1024, 64, device=dev)
            dt = timeit(lambda: asr(x), args.iters)
            out["asr_eval_B32_T1024"] = {"ms": round(dt * 1e3, 3), "frames_per_s": round(32 * 1024 / dt, 1)}

        # config 3, audio model alone and the chain; B = 16 per GPU (weak scaling: the request batch grows with the node)
        B3 = 16 * world
        at = torch.randint(0, 29, (B3, 512), generator=g).to(dev)
        dt = timeit(lambda: scatter_run(lambda a: tts.predict(a), (at,)), args.iters)
        out["config3_tts_predict_B16_L512_per_gpu"] = {"ms": round(dt * 1e3, 3), "aligntext_frames_per_s": round(B3 * 512 / dt, 1),
                                                       "world_frames_per_s": round(B3 * 1023 / dt, 1)}
        text = torch.randint(1, 29, (B3, 128), generator=g).to(dev)
        tlen = torch.randint(64, 129, (B3,), generator=g).to(dev)

        def run_chain(t, n):
            r = chain(t, n)
            return r["f0"], r["spc"], r["codeap"], r["frames"]
        res = scatter_run(run_chain, (text, tlen))
        dt = timeit(lambda: scatter_run(run_chain, (text, tlen)), max(5, args.iters // 2))
        frames = int(res[3].sum()) if res is not None else 0
        out["config3_chain_text128_to_spc_B16_per_gpu"] = {"ms": round(dt * 1e3, 3), "world_frames": frames,
                                                           "world_frames_per_s": round(frames / dt, 1) if frames else None}

        # config 5: chunks round-robined over the ranks
        for B in sorted({32 * world, args.chunks * world}):
            wav = (torch.rand(B, 16000, generator=g) * 2 - 1).to(dev)
            dt = timeit(lambda: scatter_run(lambda w: stream(w), (wav,), mode="round_robin"), args.iters)
            out[f"config5_stream_1s_chunks_B{B // world}_per_gpu"] = {
                "ms": round(dt * 1e3, 3), "chunks_per_s": round(B / dt, 1), "frames_per_s": round(B * 101 / dt, 1),
                "x_realtime": round(B / dt, 1)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
