#!/usr/bin/env python3
"""configs[2] alone (AlignTextToAudioModel.predict, B = 16 x 512 aligned tokens, bf16) for a kernel trace:
rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o p -- python3 tools/trace_predict.py [--stream]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from voice100_amd import functional as F_, _native as N
dev = torch.device("cuda:0")
N.load()
stream = "--stream" in sys.argv
with torch.no_grad():
    if stream:
        from voice100_amd.asr import AudioToTextCTC
        from voice100_amd.mel import MelSpectrogramAudioTransform
        from voice100_amd.infer import ASRPipeline
        F_.set_matmul_precision("fp16")
        pipe = ASRPipeline(AudioToTextCTC(64, 512, 29, 512).to(dev).eval(), MelSpectrogramAudioTransform().to(dev))
        wav = torch.rand(256, 16000, device=dev) * 2 - 1
        fn = lambda: pipe(wav)
    else:
        from voice100_amd.tts import AlignTextToAudioModel
        F_.set_matmul_precision("bf16")
        tts = AlignTextToAudioModel(vocab_size=29, hidden_size=512, use_mcep=False).to(dev).eval()
        at = torch.randint(0, 29, (16, 512), generator=torch.Generator().manual_seed(7)).to(dev)
        fn = lambda: tts.predict(at)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    for _ in range(20): fn()
    torch.cuda.synchronize()
