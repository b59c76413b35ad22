#!/usr/bin/env python3
"""One inference workload, a few iterations, for rocprofv3 --kernel-trace --stats (kernel tables of the inference configs):
    python tools/prof_eval.py asr32 bf16 | stream256 fp16 | stream32 bf16 | asr2 bf16 | predict16 bf16 | chainwave bf16     [--iters 10]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    what, prec = sys.argv[1], sys.argv[2]
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
    from voice100_amd import functional as F_
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel, TextToAlignTextModel
    from voice100_amd.mel import MelSpectrogramAudioTransform
    from voice100_amd.vocoder import WORLDVocoder
    from voice100_amd.decode import ctc_greedy_decode
    from voice100_amd.infer import TTSPipeline
    dev = torch.device("cuda")
    F_.set_matmul_precision(prec)
    torch.manual_seed(1234)
    with torch.no_grad():
        if what == "asr32":
            m = AudioToTextCTC(64, 512, 29, 512).to(dev).eval()
            x = torch.rand(32, 1024, 64, device=dev)
            fn = lambda: m(x)
        elif what == "stream256":
            m = AudioToTextCTC(64, 512, 29, 512).to(dev).eval()
            mel = MelSpectrogramAudioTransform().to(dev)
            wav = torch.rand(256, 16000, device=dev) * 2 - 1
            fn = lambda: ctc_greedy_decode(m(mel(wav)))
        elif what in ("stream32", "stream32nd"):        # configs[4] at B = 32: 1-second chunks, 101 frames -> 51
            m = AudioToTextCTC(64, 512, 29, 512).to(dev).eval()
            mel = MelSpectrogramAudioTransform().to(dev)
            wav = torch.rand(32, 16000, device=dev) * 2 - 1
            fn = (lambda: ctc_greedy_decode(m(mel(wav)))) if what == "stream32" else (lambda: m(mel(wav)))
        elif what == "asr2":                            # configs[0]: B = 2 x 256 frames
            m = AudioToTextCTC(64, 512, 29, 512).to(dev).eval()
            x = torch.rand(2, 256, 64, device=dev)
            fn = lambda: m(x)
        elif what == "predict16":
            t = AlignTextToAudioModel(vocab_size=29, hidden_size=512, use_mcep=False).to(dev).eval()
            at = torch.randint(0, 29, (16, 512), device=dev)
            fn = lambda: t.predict(at)
        elif what == "chainwave":
            al = TextToAlignTextModel(vocab_size=29, hidden_size=512).to(dev).eval()
            al.layers[4].bias.copy_(torch.tensor([0.6931, 1.3863], device=dev))
            au = AlignTextToAudioModel(vocab_size=29, hidden_size=512, use_mcep=True).to(dev).eval()
            chain = TTSPipeline(al, au, WORLDVocoder(use_mcep=True).to(dev))
            text = torch.randint(1, 29, (16, 128), device=dev)
            tlen = torch.randint(64, 129, (16,), device=dev)
            fn = lambda: chain(text, tlen)
        else:
            raise SystemExit("unknown workload")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        print(f"{what} {prec}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per call")


if __name__ == "__main__":
    main()
