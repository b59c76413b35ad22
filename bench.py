#!/usr/bin/env python3
"""Headline benchmark: audio frames/sec, fwd+bwd(+optimizer), "asr_en_base" =
AudioToTextCTC(audio_size=64, embed_size=512, vocab_size=29, hidden_size=512)  (SURVEY.md 0.2),
B = 32 x 1024-frame synthetic log-mel per GPU, CTC targets [32, 100]   (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; batches shard across ranks (weak scaling, 32 utterances per GPU), one RCCL
gradient all-reduce per step overlapped with backward.  Rank 0 prints ONE JSON line.

A "step" = augmentation + forward + log_softmax/CTC + backward + gradient exchange + Adam, with
inputs already resident in HBM.  `roofline` = depthwise forward kernel (the north-star's graded
kernel): algorithmic bytes of its 9 launches per step / their HIP-event time inside the timed
region.  `cpu_baseline` = the CPU oracle (port of the reference's PyTorch-CPU path) timed on this
box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy rate ~6300
PMC_FILE = os.path.join(ROOT, "profiles", "dw_fwd_pmc.json")   # HBM bytes / algorithmic bytes of the depthwise forward kernels (rocprofv3 PMC passes, tools/pmc_table.py --json), tagged with the kernel sources' hash
B_PER_GPU, T_FRAMES, N_MEL, VOCAB, TEXT_LEN = 32, 1024, 64, 29, 100


def dw_source_sha():
    """sha256 (16 hex) over the depthwise kernel sources: a stored PMC ratio is only quoted for the kernels it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "voice100_amd", "csrc", "depthwise*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def csrc_sha():
    """sha256 (16 hex) over every kernel source: profiles/step_pmc.json (HBM bytes of the whole step) is quoted only for the tree it was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "voice100_amd", "csrc", "*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def encoder_dw_bytes(B, T):
    """Algorithmic bytes of the 9 depthwise forward launches: fp32 in + out + taps + BN coeffs."""
    specs = [(256, 11, 2), (1024, 19, 1), (1024, 27, 1), (1024, 35, 1), (1024, 51, 1),
             (2048, 59, 1), (2048, 67, 1), (2048, 75, 1), (2048, 83, 1)]
    total, t = 0, T
    per_layer = []
    for hid, k, s in specs:
        tout = (t - 1) // s + 1
        b = 4 * B * hid * (t + tout) + 4 * hid * k + 8 * hid
        per_layer.append(b)
        total += b
        t = tout
    return total, per_layer


def synth_batch(device, B, seed, frames=T_FRAMES):
    g = torch.Generator().manual_seed(seed)
    audio = (torch.randn(B, frames, N_MEL, generator=g) * 2 - 4).clamp_min(float(torch.log(torch.tensor(1e-6))))
    audio_len = torch.full((B,), frames, dtype=torch.int32)
    text = torch.randint(1, VOCAB, (B, TEXT_LEN), generator=g)
    text_len = torch.full((B,), TEXT_LEN, dtype=torch.int32)
    return ((audio.to(device), audio_len.to(device)), (text.to(device), text_len.to(device)))


def usable_cpus():
    """Logical CPUs this process may run on at once: the scheduler affinity, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def gpu_clock_power(local_rank=0):
    """Shader clock (MHz) and board power (W) as the driver reports them right now, read from sysfs (no GPU call, no child
    process): the `*`-marked level of pp_dpm_sclk and hwmon's freq1_input / power1_average.  Fields that cannot be read are None.
    (MI355X_MICROARCH.md 'DVFS give-back': pp_dpm_sclk is not the in-kernel clock -- it reads up to ~10 % above it under an
    MFMA-dense load -- but it does show whether the device sits at its idle or its loaded level.)"""
    import glob
    out = {"sclk_mhz": None, "hwmon_freq_mhz": None, "power_w": None, "source": None}
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
    if not cards:
        return out
    dev = None
    try:        # the card whose PCI address is this process's HIP device (a box may expose several cards of which one is ours)
        pr = torch.cuda.get_device_properties(local_rank)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        for c in cards:
            if os.path.basename(os.path.realpath(os.path.dirname(c))).startswith(want):
                dev = os.path.dirname(c)
    except Exception:                                            # noqa: BLE001
        pass
    if dev is None:
        dev = os.path.dirname(cards[min(local_rank, len(cards) - 1)])
        out["card_matched_by"] = "index (PCI address not matched)"
    out["source"] = dev
    try:
        for line in open(os.path.join(dev, "pp_dpm_sclk")):
            if line.rstrip().endswith("*"):
                out["sclk_mhz"] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
    except (OSError, ValueError, IndexError):
        pass
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        for name, key, scale in (("freq1_input", "hwmon_freq_mhz", 1e-6), ("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6)):
            try:
                if out[key] is None:
                    out[key] = round(int(open(os.path.join(hw, name)).read()) * scale, 1)
            except (OSError, ValueError):
                pass
    return out


def cpu_baseline():
    """Oracle (CPU port of the reference path, stock PyTorch CPU ops, fp32) on this box's host cores, per BASELINE.md 3:
    (ii) the metric -- train-mode forward + log_softmax + CTC + backward at B = 32 x T = 1024, 1 warm-up + 3 timed iterations per
    thread count, thread counts swept upwards (8, 16, 32, ... up to the usable cores) until throughput stops improving;
    (i) config 1 -- eval forward B = 2 x 256 frames, 3 warm-ups + 20 timed; (iii) config 3 -- AlignTextToAudioModel.predict
    B = 16 x L = 512, 1 warm-up + 3 timed.  `value` is (ii), the bench's own workload."""
    from oracle import cnn
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel
    usable = usable_cpus()
    torch.manual_seed(1234)
    m = AudioToTextCTC(N_MEL, 512, VOCAB, 512)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in state.items() if v.dtype.is_floating_point and "running" not in k}
    state.update(params)
    batch = synth_batch("cpu", B_PER_GPU, 1234)
    counts = sorted({c for c in (8, 16, 32, 64, 128) if c < usable} | {min(usable, 128)})
    best_t, best_n, tried = None, None, []
    for n in counts:
        torch.set_num_threads(n)
        times = []
        for i in range(4):
            for p in params.values():
                p.grad = None
            t0 = time.perf_counter()
            loss = cnn.audio_to_text_ctc_loss(batch, state, training=True)
            loss.backward()
            if i > 0:
                times.append(time.perf_counter() - t0)
        t = sum(times) / len(times)
        tried.append((n, round(B_PER_GPU * T_FRAMES / t, 1)))
        if best_t is not None and t > best_t:
            break                                  # past the knee: more threads only oversubscribe the shared host
        best_t, best_n = t, n
    torch.set_num_threads(best_n)
    with torch.no_grad():
        ev = {k: v.detach() for k, v in state.items()}
        x = torch.rand(2, 256, N_MEL)
        for _ in range(3):
            cnn.audio_to_text_ctc_forward(x, ev, training=False)
        t0 = time.perf_counter()
        for _ in range(20):
            cnn.audio_to_text_ctc_forward(x, ev, training=False)
        c1 = (time.perf_counter() - t0) / 20
        tts = AlignTextToAudioModel(vocab_size=VOCAB, hidden_size=512, use_mcep=False)
        ts = {k: v.detach() for k, v in tts.state_dict().items()}
        at = torch.randint(0, VOCAB, (16, 512))
        cnn.align_text_to_audio_predict(at, ts)
        t0 = time.perf_counter()
        for _ in range(3):
            cnn.align_text_to_audio_predict(at, ts)
        c3 = (time.perf_counter() - t0) / 3
    return {"value": round(B_PER_GPU * T_FRAMES / best_t, 1), "unit": "frames/s", "cores": best_n, "kind": "port",
            "sample": f"oracle.cnn train fwd + log_softmax + CTC + bwd at the bench workload (B={B_PER_GPU} x T={T_FRAMES}, fp32), 1 warm-up + "
                      f"3 timed iterations per thread count; (threads, frames/s) tried = {tried}; {usable} usable logical CPUs "
                      f"({os.cpu_count()} visible)",
            "config1_eval_B2_T256": {"ms": round(c1 * 1e3, 2), "frames_per_s": round(2 * 256 / c1, 1), "cores": best_n},
            "config3_tts_predict_B16_L512": {"ms": round(c3 * 1e3, 1), "aligntext_frames_per_s": round(16 * 512 / c3, 1),
                                             "world_frames_per_s": round(16 * 1023 / c3, 1), "cores": best_n}}


def other_configs(device, N, F_):
    """BASELINE configs 1, 3 and 5 on the GPU, measured after the timed region of the same run (rank 0, N = 1): latency /
    throughput of each, the depthwise forward kernels' HBM fraction inside config 3's VoiceDecoder, and the time of the
    log-mel front-end inside config 5 (its algorithmic bytes: waveform in + log-mel out)."""
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel
    from voice100_amd.mel import MelSpectrogramAudioTransform
    from voice100_amd.decode import ctc_greedy_decode

    def timeit(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    out = {}
    with torch.no_grad():
        asr = AudioToTextCTC(N_MEL, 512, VOCAB, 512).to(device).eval()
        x = torch.rand(2, 256, N_MEL, device=device)
        dt = timeit(lambda: asr(x))
        out["config1_asr_eval_B2_T256"] = {"ms": round(dt * 1e3, 3), "frames_per_s": round(2 * 256 / dt, 1)}
        tts = AlignTextToAudioModel(vocab_size=VOCAB, hidden_size=512, use_mcep=False).to(device).eval()
        at = torch.randint(0, VOCAB, (16, 512), device=device)
        dt = timeit(lambda: tts.predict(at))
        N.timing_enable(["dw_fwd"])
        for _ in range(5):
            tts.predict(at)
        torch.cuda.synchronize()
        n, ms, nbytes = N.timing_read().get("dw_fwd", (0, 0.0, 0.0))
        N.timing_enable(False)
        out["config3_tts_predict_B16_L512"] = {
            "ms": round(dt * 1e3, 3), "aligntext_frames_per_s": round(16 * 512 / dt, 1), "world_frames_per_s": round(16 * 1023 / dt, 1),
            "voice_decoder_dw_gbs": round(nbytes / (ms * 1e-3) / 1e9, 1) if n else None,
            "voice_decoder_dw_frac_of_8tbs": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if n else None,
            "voice_decoder_dw_launches": n}
        # configs[2] is "align + audio model": the align model alone, then the whole chain text -> TextToAlignTextModel -> align()
        # -> AlignTextToAudioModel.predict -> mcep -> log-spectrum GEMM -> exp/clip on the device (voice100_amd.infer.TTSPipeline;
        # tts.py:79-110, 172-201, vocoder.py:89-99), B = 16 sentences of 128 tokens
        from voice100_amd.tts import TextToAlignTextModel
        from voice100_amd.vocoder import WORLDVocoder
        from voice100_amd.infer import TTSPipeline
        al = TextToAlignTextModel(vocab_size=VOCAB, hidden_size=512).to(device).eval()
        # an untrained head predicts ~0 frames per token: bias it to gap ~ 1, length ~ 3 frames, so 128 tokens expand to ~500 aligned
        # frames = ~1000 WORLD frames per sentence, the sizes configs[2] names
        al.layers[4].bias.copy_(torch.tensor([0.6931, 1.3863], device=device))
        text = torch.randint(1, VOCAB, (16, 128), device=device)
        tlen = torch.randint(64, 129, (16,), device=device)
        dt_al = timeit(lambda: al(text))
        au = AlignTextToAudioModel(vocab_size=VOCAB, hidden_size=512, use_mcep=True).to(device).eval()
        voc = WORLDVocoder(use_mcep=True).to(device)
        chain = TTSPipeline(al, au, voc, synthesize=False)
        dt_ch = timeit(lambda: chain(text, tlen))
        wf = int(chain(text, tlen)["frames"].sum())
        out["config3_align_model_B16_L128"] = {"ms": round(dt_al * 1e3, 3), "tokens_per_s": round(16 * 128 / dt_al, 1)}
        out["config3_chain_text_to_spc_B16_L128"] = {"ms": round(dt_ch * 1e3, 3), "world_frames": wf,
                                                    "world_frames_per_s": round(wf / dt_ch, 1)}
        # ... and on to the WAVEFORM (configs[2]: "... -> WORLD features + vocoder"): decode_aperiodicity + synthesize on the device
        # (csrc/world.hip, parity unpinned), 160 samples per WORLD frame at 16 kHz
        chain_w = TTSPipeline(al, au, voc, synthesize=True)
        dt_w = timeit(lambda: chain_w(text, tlen))
        ow = chain_w(text, tlen)
        secs = float(ow["wave_len"].sum()) / 16000.0
        out["config3_chain_text_to_wave_B16_L128"] = {"ms": round(dt_w * 1e3, 3), "world_frames": wf, "audio_seconds": round(secs, 1),
                                                     "pulses": int(ow["n_pulses"].sum()), "x_realtime": round(secs / dt_w, 1),
                                                     "vocoder_ms": round((dt_w - dt_ch) * 1e3, 3)}
        # ... and the way back (the dataset side of the same models, vocoder.py:61-87): WORLD ANALYSIS on the device -- DIO + CheapTrick +
        # D4C + aperiodicity coding + mcep GEMM, fp64 kernels (csrc/world_analysis.hip, parity unpinned) -- of 16 speech-like signals of
        # 10.23 s (harmonics on a wandering F0 gated voiced / unvoiced over a noise floor; the untrained chain's own output is almost
        # entirely unvoiced, which D4C skips)
        from tools.bench_world_analysis import speechlike
        sig = torch.from_numpy(np.stack([speechlike(10.23, 16000, s) for s in range(16)])).to(device)
        dt_an = timeit(lambda: voc.encode_batch(sig))
        an_f0 = voc.encode_batch(sig)[0]
        secs_an = 16 * 10.23
        out["config3_world_analysis_B16"] = {"ms": round(dt_an * 1e3, 3), "audio_seconds": round(secs_an, 1), "x_realtime": round(secs_an / dt_an, 1),
                                             "world_frames_per_s": round(an_f0.numel() / dt_an, 1), "dtype": "f64",
                                             "voiced_frames": int((an_f0 > 0).sum()), "frames": int(an_f0.numel()),
                                             "input": "16 x 10.23 s synthetic speech-like signals (tools/bench_world_analysis.py)"}
        F_.set_matmul_precision("fp16")                 # config 5 names fp16
        mel = MelSpectrogramAudioTransform().to(device)
        B = 256
        wav = torch.rand(B, 16000, device=device) * 2 - 1
        dt_mel = timeit(lambda: mel(wav))
        feats = mel(wav)

        def stream():
            ids, _ = ctc_greedy_decode(asr(mel(wav)))
            return ids
        dt = timeit(stream)
        mel_bytes = 4.0 * B * (16000 + feats.shape[1] * feats.shape[2])
        out["config5_stream_1s_chunks_fp16_B256"] = {
            "ms": round(dt * 1e3, 3), "chunks_per_s": round(B / dt, 1), "mel_frontend_ms": round(dt_mel * 1e3, 3),
            "mel_frontend_algorithmic_gbs": round(mel_bytes / dt_mel / 1e9, 1)}
        F_.set_matmul_precision("bf16")
    return out


def lean_line(out):
    """The printed line: same keys and figures, without the long prose / per-launch tables (those are in the details record)."""
    import copy
    o = copy.deepcopy(out)

    def cut(d, key, n):
        if isinstance(d, dict) and isinstance(d.get(key), str) and len(d[key]) > n:
            d[key] = d[key][:n - 3] + "..."
    cfg = o.get("config") or {}
    if isinstance(cfg.get("workload"), str):
        cfg["workload"] = cfg["workload"].split(" (0 fp32 everywhere")[0]
        cut(cfg, "workload", 180)
    r = o.get("roofline")
    if isinstance(r, dict):
        r.pop("traffic_per_launch", None)
        cut(r, "kernel", 60)
        cut(r, "traffic_source", 60)
        cut(r, "note", 160)
        for k in ("measured_copy_gbs", "frac_of_measured_copy", "copy_probe_gbs", "frac_of_copy_probe", "algorithmic_bytes_nominal_step"):
            r.pop(k, None)                # (the plain-copy yardsticks stay in the details record; the line keeps the pattern copy)
    rs = o.get("roofline_step")
    if isinstance(rs, dict):
        for k in ("note", "families_algorithmic_mb"):
            rs.pop(k, None)
    if isinstance(r, dict):
        r.pop("frac_note", None)
    if isinstance(o.get("launches_per_step"), dict):
        o["launches_per_step"] = o["launches_per_step"].get("value")
    if isinstance(o.get("dp_path_single_rank"), dict):
        o["dp_path_single_rank"].pop("what", None)
    cut(o, "kernel_ms_method", 60)
    if o.get("launch_graph_diagnostic") is None:
        o.pop("launch_graph_diagnostic", None)
    cb = o.get("cpu_baseline")
    if isinstance(cb, dict):
        cut(cb, "sample", 110)
        for k in [k for k in cb if k.startswith("config")]:
            cb.pop(k)
    oc = o.get("other_configs")
    if isinstance(oc, dict):
        keep = ("ms", "chunks_per_s", "x_realtime", "error")
        for k, v in list(oc.items()):
            if isinstance(v, dict):
                oc[k] = {kk: vv for kk, vv in v.items() if kk in keep}
    su = o.get("sustained")
    if isinstance(su, dict):
        for k in ("before", "steps"):
            su.pop(k, None)
        if isinstance(su.get("under_load"), dict):
            su["under_load"] = {k: v for k, v in su["under_load"].items() if k in ("sclk_mhz", "power_w")}
    return o


def main():
    if os.environ.get("V100_BENCH_WATCHDOG"):          # diagnostic: dump every thread's Python stack (and exit) after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["V100_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--windows", type=int, default=5, help="extra timed windows of --steps steps after the reported one (spread only)")
    ap.add_argument("--host-contention", type=int, default=8, help="after the timed region: the same steps beside N-1 busy host processes "
                    "(what the enqueueing thread of one rank sees when 8 ranks share this host); 0/1 = skip")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="after the contract window: back-to-back steps for at least this "
                    "long (steady-state clocks; reported as `sustained`, never as `value`); 0 = skip")
    ap.add_argument("--graph-diag", action="store_true", help="also time a hipGraph replay of the step (timing diagnostic; a failed capture can "
                    "poison the context, so it is not part of the default run)")
    ap.add_argument("--no-extras", action="store_true", help="skip fp32_ms_per_step and dp_path_single_rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--diag-no-timestretch", action="store_true", help="diagnostic only (NOT the benchmark workload): never draw a time-stretch")
    ap.add_argument("--diag-stretch-rate", type=int, default=0, help="diagnostic only: every drawn time-stretch uses this rate (percent)")
    ap.add_argument("--no-presize", action="store_true", help="skip the untimed allocator-sizing step at 149 %% length (PMC passes over this "
                    "script want every dispatch of a kernel to be the same problem)")
    args = ap.parse_args()

    from voice100_amd.trainer import TrainStep, init_distributed, launch_ranks

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, torchrun rendezvous on
        # 127.0.0.1) as a child process, BEFORE this process touches the GPU, and leave with the child's exit code.
        # (no device query here: this process must not hold a GPU handle; a rank that finds no device for its LOCAL_RANK
        # fails with torch's own message and the launcher returns its exit code)
        raise SystemExit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    # The contract is ONE JSON line on stdout.  RCCL prints its version banner with printf at communicator creation (it lands in the C
    # library's buffer and reaches fd 1 at exit, AFTER the JSON line), rocm libraries do similar: keep a private handle on the real
    # stdout for the JSON line and point fd 1 at stderr for everything else, C-level writers included.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from voice100_amd import functional as F_, _native as N
    from voice100_amd.asr import AudioToTextCTC

    rank, local_rank, world = init_distributed()
    if dist.is_initialized():
        world = dist.get_world_size()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world} rank(s): launch with "
                         f"`python bench.py --gpus N` or torchrun --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    N.load()
    F_.set_matmul_precision(args.precision)

    # reference: pl.seed_everything(1234), train_asr.py:13 -- seeds python, numpy and torch; the augmentation draws its
    # decisions from python's `random` (audio.py:35-49), so an unseeded run changes which steps are time-stretched
    # Every rank gets the SAME seed, as seed_everything does under Lightning's DDP: identical initial weights (no
    # broadcast needed) and identical augmentation decisions per step on all ranks (so no rank is the straggler of a
    # step because it alone drew a long time-stretch); only the synthetic data differs per rank.
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    model = AudioToTextCTC(N_MEL, 512, VOCAB, 512, learning_rate=1e-3, weight_decay=4e-5).to(device)
    step = TrainStep(model)
    if args.diag_no_timestretch:
        model.batch_augment.do_timestretch = False
    if args.diag_stretch_rate:
        import voice100_amd.audio as _audio
        _audio.SPECTROGRAM_AUGUMENT_RATE = 1.0          # every op of the augmentation fires: diagnostic workload
        model.batch_augment._diag_rate = args.diag_stretch_rate
    batch = synth_batch(device, B_PER_GPU, 1234 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed: size PyTorch's caching allocator for the longest batch a time-stretch can produce (149 % of T) with one
    # extra step on such a batch.  Otherwise the first stretched step of every new length calls hipMalloc inside the
    # timed region (several ms each), which decides the result of short runs.  RNG streams are restored afterwards.
    rng = (random.getstate(), np.random.get_state(), torch.random.get_rng_state(), torch.cuda.get_rng_state(device))
    aug = getattr(model, "batch_augment", None)
    if aug is not None and not args.no_presize:
        keep, aug.do_timestretch = aug.do_timestretch, False
        (a_long, _), tgt = synth_batch(device, B_PER_GPU, 99, frames=T_FRAMES * 149 // 100)
        step(((a_long, torch.full((B_PER_GPU,), a_long.shape[1], dtype=torch.int32, device=device)), tgt))
        aug.do_timestretch = keep
        del a_long
    random.setstate(rng[0]); np.random.set_state(rng[1]); torch.random.set_rng_state(rng[2]); torch.cuda.set_rng_state(rng[3], device)

    for _ in range(args.warmup):
        step(batch)
    sync()
    if not args.no_kernel_timing:
        # HIP events inside the library, on the launch stream, for the roofline kernel only.  Its launches carry the
        # events in the dispatch packet (hipExtLaunchKernelGGL: the pair reads the kernel's own start/stop timestamps);
        # bracketing all ~130 hot launches of a step with hipEventRecord costs ~0.8 ms of it, hence the separate pass below
        N.timing_enable(["dw_fwd"])
    launches0 = N.launch_count()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(batch)
    host_loop = time.perf_counter() - t0          # the host has ENQUEUED the last step; the GPU may still be running
    sync()
    elapsed = time.perf_counter() - t0
    launches = (N.launch_count() - launches0) / args.steps      # the library's own launches, counted in this run (torch adds ~5 tiny elementwise ones)
    if not args.no_kernel_timing:
        kt_main = N.timing_read()
        N.timing_enable(False)
    # The contract's window above is `value`.  A 20-step window is ~0.07 s, and which of its steps draw a time-stretch moves it
    # by several %: `windows` repeats it (same barrier + synchronise bracket, RNG streams running on) to show the spread.
    windows = []
    for _ in range(args.windows):
        sync()
        tw = time.perf_counter()
        for _ in range(args.steps):
            step(batch)
        sync()
        windows.append((time.perf_counter() - tw) / args.steps * 1e3)
    # Host side under contention: N-1 busy processes stand in for the other ranks' host threads on this box (each rank's Python is one
    # enqueueing thread); same steps, same bracket.  Not part of `value`.
    contended = None
    if world == 1 and args.host_contention > 1:
        import subprocess
        burners = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(args.host_contention - 1)]
        try:
            time.sleep(0.2)
            sync()
            tc = time.perf_counter()
            for _ in range(args.steps):
                step(batch)
            host_c = time.perf_counter() - tc
            sync()
            total_c = time.perf_counter() - tc
        finally:
            for b_ in burners:                      # the exact processes started above, nothing else
                b_.kill()
            for b_ in burners:
                b_.wait()
        contended = {"busy_processes": args.host_contention - 1, "usable_cpus": usable_cpus(),
                     "host_enqueue_ms_per_step": round(host_c / args.steps * 1e3, 3), "ms_per_step": round(total_c / args.steps * 1e3, 3)}
    # Sustained window (review round 3, item 2b): the contract window above is ~0.07 s of GPU time -- a cold burst.  Here the same
    # steps run back to back for >= --sustained-seconds (same RNG streams running on, so ~30 % of them are time-stretched as in
    # the contract window), and the driver-reported shader clock / board power are read at ~80 % of the window, while the queue is
    # full.  Not part of `value`.
    sustained = None
    if world == 1 and args.sustained_seconds > 0:
        n_sus = max(args.steps, int(args.sustained_seconds / max(elapsed / args.steps, 1e-4) * 1.05))
        idle = gpu_clock_power(local_rank)
        sync()
        ts = time.perf_counter()
        mid = None
        for i in range(n_sus):
            step(batch)
            if i == int(n_sus * 0.8):
                mid = gpu_clock_power(local_rank)
        host_s = time.perf_counter() - ts
        sync()
        total_s = time.perf_counter() - ts
        sustained = {"steps": n_sus, "seconds": round(total_s, 2), "ms_per_step": round(total_s / n_sus * 1e3, 3),
                     "frames_per_s": round(B_PER_GPU * T_FRAMES * n_sus / total_s, 1),
                     "host_enqueue_ms_per_step": round(host_s / n_sus * 1e3, 3),
                     "under_load": mid, "before": idle}
    kt, kt_all, kt_nom, kt_all_raw = {}, {}, {}, {}
    if not args.no_kernel_timing:
        kt = kt_main
        # per-kernel-family breakdown: a separate pass after the timed region (it slows the step, see above)
        FAM = ["dw_fwd", "dw_bwd_data", "dw_wgrad", "pw_gemm", "pw_wgrad"]
        N.timing_enable(FAM)                                # (round 5: dispatch-packet timestamps for the GEMM families too; r04 and before bracketed them with event markers, ~+0.2 ms)
        nb = max(2, min(10, args.steps))
        for _ in range(nb):
            step(batch)
        kt_all_raw = dict(N.timing_read())
        kt_all = {k: round(v[1] / nb, 3) for k, v in sorted(kt_all_raw.items())}
        N.timing_enable(False)
        # the same pass on the NOMINAL step only (time-stretch off: every launch is the 1024-frame problem the algorithmic byte model
        # and the PMC passes describe) -> roofline_step.kernel_ms and the depthwise forward's nominal-only fraction
        if aug is not None:
            keep_ts, aug.do_timestretch = aug.do_timestretch, False
        step(batch)
        # one pass with every tag on: since round 5 the GEMM families are timed like everything else, by their dispatch packets' own
        # timestamps (csrc/timing.hip) -- no marker packets, so the tags no longer disturb each other and the sum IS the step's kernel time
        N.timing_enable(FAM + ["other"])
        for _ in range(nb):
            step(batch)
        kt_nom.update({k: (v[0] / nb, v[1] / nb, v[2] / nb) for k, v in sorted(N.timing_read().items())})
        N.timing_enable(False)
        if aug is not None:
            aug.do_timestretch = keep_ts
    # fp32 line (item 2c): the same step in the reference's own default arithmetic (exact-fp32 MFMA GEMMs, fp32 storage everywhere),
    # same model / optimiser state running on.  Extra key only.
    fp32_line = None
    dp_line = None
    if world == 1 and not args.no_extras and args.precision == "bf16":
        try:
            F_.set_matmul_precision("fp32")
            for _ in range(2):
                step(batch)
            sync()
            tf = time.perf_counter()
            for _ in range(args.steps):
                step(batch)
            sync()
            fp32_line = round((time.perf_counter() - tf) / args.steps * 1e3, 3)
        except Exception as e:                                       # noqa: BLE001
            fp32_line = f"error: {type(e).__name__}: {e}"
        finally:
            F_.set_matmul_precision(args.precision)
        # data-parallel path on ONE rank (item 2d): a one-rank RCCL group, the flat gradient buffer + 16 MB buckets + post-accumulate
        # hooks + async all-reduce + mean of voice100_amd/dist.py forced on (force_exchange), the encoder cut into three-block
        # autograd segments as under world > 1 -- i.e. everything a rank of an 8-GPU job does except having peers.
        try:
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            created = False
            if not dist.is_initialized():
                # an explicit store of our own: under torchrun (a one-rank launch leaves the group uninitialised here) both the env:// and
                # the tcp:// rendezvous handlers act as CLIENTS of the launcher's agent store (TORCHELASTIC_USE_AGENT_STORE) and wait for ever
                # for a server on a port that has none
                own_store = dist.TCPStore("127.0.0.1", port, 1, True)
                dist.init_process_group("nccl", store=own_store, rank=0, world_size=1)
                created = True
            try:
                step.buckets.remove_hooks()
                dp_step = TrainStep(model, force_exchange=True)
                F_.set_stack_segment(3)
                for _ in range(3):
                    dp_step(batch)
                sync()
                l0 = N.launch_count()
                td = time.perf_counter()
                for _ in range(args.steps):
                    dp_step(batch)
                host_d = time.perf_counter() - td
                sync()
                tot_d = time.perf_counter() - td
                dp_line = {"ms_per_step": round(tot_d / args.steps * 1e3, 3), "host_enqueue_ms_per_step": round(host_d / args.steps * 1e3, 3),
                           "library_launches_per_step": round((N.launch_count() - l0) / args.steps, 1),
                           "buckets": len(dp_step.buckets.buckets), "flat_mb": round(dp_step.buckets.flat.numel() * 4 / 1e6, 1),
                           "what": "one-rank RCCL group, force_exchange=True (flat buffer with the stack gradients written in place, bucketed async "
                                   "all-reduce from post-accumulate hooks), encoder as three-block autograd segments; same seeded step sequence running on"}
                dp_step.buckets.remove_hooks()
            finally:
                F_.set_stack_segment(None)
                if created:
                    dist.destroy_process_group()
        except Exception as e:                                       # noqa: BLE001
            dp_line = {"error": f"{type(e).__name__}: {e}"}
    # hipGraph replay of the step (review round 3, item 5) -- a TIMING DIAGNOSTIC, not a training mode: the captured step re-uses the
    # augmentation decisions, the dropout seed and the Adam bias-correction scalars of the capture (they are host-drawn kernel arguments),
    # so replays are not valid optimisation steps; what the replay shows is what a launch graph could buy: host time per step and whether
    # the GPU time per step moves at all once the ~150 launches arrive without host gaps.  Nominal length only (time-stretch off).
    graph_diag = None
    if world == 1 and args.graph_diag:
        keep_ts = aug.do_timestretch if aug is not None else None
        try:
            if aug is not None:
                aug.do_timestretch = False
            gstep = TrainStep(model)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    gstep(batch)
            torch.cuda.current_stream().wait_stream(side)
            sync()
            te = time.perf_counter()
            for _ in range(args.steps):
                gstep(batch)
            host_e = time.perf_counter() - te
            sync()
            tot_e = time.perf_counter() - te
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gstep(batch)
            for _ in range(3):
                graph.replay()
            sync()
            tg = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            host_g = time.perf_counter() - tg
            sync()
            tot_g = time.perf_counter() - tg
            graph_diag = {"eager_nominal_T": {"ms_per_step": round(tot_e / args.steps * 1e3, 3), "host_enqueue_ms_per_step": round(host_e / args.steps * 1e3, 3)},
                          "graph_replay": {"ms_per_step": round(tot_g / args.steps * 1e3, 3), "host_enqueue_ms_per_step": round(host_g / args.steps * 1e3, 3)},
                          "note": "timing diagnostic only: replays repeat the captured step's host-drawn arguments (augmentation decisions, dropout seed, "
                                  "Adam scalars); time-stretch off in both arms"}
            del graph
        except Exception as e:                                       # noqa: BLE001
            graph_diag = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        finally:
            if aug is not None and keep_ts is not None:
                aug.do_timestretch = keep_ts
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    if rank == 0:
        # the fraction is reported against the nominal HBM rate AND a device-copy rate measured on this box
        src = torch.empty(1 << 28, device=device, dtype=torch.float32).normal_()      # 1 GiB read + 1 GiB write per copy
        dst = torch.empty_like(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dst.copy_(src); sync()
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); sync()
        copy_gbs = 10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        # the same 1 GiB -> 1 GiB copy by the library's own 16-byte-per-lane kernel (v100_copy_probe): torch's copy_ is not the
        # best a kernel can do on this box, so both yardsticks are reported
        N.call("v100_copy_probe", src, dst, src.numel() * 4); sync()
        e0.record()
        for _ in range(10):
            N.call("v100_copy_probe", src, dst, src.numel() * 4)
        e1.record(); sync()
        probe_gbs = 10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        # ... and a copy with the depthwise kernels' OWN access pattern (v100_rows_copy_probe: 1 KB bf16 rows of a [B][C][T] tensor, one
        # workgroup per channel, one row per wave at a time, nontemporal) at the wide layers' shape, rotating over 4 buffer sets
        # (> 256 MB between two uses of a line: HBM, not the Infinity Cache): what this LAYOUT lets a kernel with no arithmetic reach.
        pat_gbs = None
        try:
            sets = [(torch.empty(B_PER_GPU * 2048 * 512 // 2, device=device, dtype=torch.float32).normal_(),
                     torch.empty(B_PER_GPU * 2048 * 512 // 2, device=device, dtype=torch.float32)) for _ in range(4)]
            for a_, b_ in sets:
                N.call("v100_rows_copy_probe", a_, b_, B_PER_GPU, 2048, 1024)
            sync()
            e0.record()
            for i in range(16):
                a_, b_ = sets[i % 4]
                N.call("v100_rows_copy_probe", a_, b_, B_PER_GPU, 2048, 1024)
            e1.record(); sync()
            pat_gbs = 16 * 2 * B_PER_GPU * 2048 * 1024 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del sets
        except Exception:                                            # noqa: BLE001
            pat_gbs = None
        frames = B_PER_GPU * T_FRAMES * world * args.steps
        dw_bytes, _ = encoder_dw_bytes(B_PER_GPU, T_FRAMES)
        roof = None
        if "dw_fwd" in kt:
            n, ms, nbytes = kt["dw_fwd"]          # bytes are summed per launch by the library (T varies with timestretch)
            achieved = nbytes / (ms * 1e-3) / 1e9
            # PMC counters cannot be read from inside this process: the per-launch HBM bytes are the algorithmic bytes of THIS
            # run's launches times the ratio rocprofv3 measured for these kernels (separate --pmc passes, 2*FETCH_SIZE +
            # WRITE_SIZE with the gfx950 correction; tools/pmc_passes.sh), null when that file is absent
            pmc = json.load(open(PMC_FILE)) if os.path.exists(PMC_FILE) else None
            if pmc is not None and pmc.get("kernel_src_sha") != dw_source_sha():
                pmc = None                      # the kernels changed since the counters were taken: no stale ratio
            act16 = F_.get_activation_storage() if args.precision == "bf16" else 0
            roof = {"bound": "hbm", "kernel": "dwconv_fwd16_stream_kernel / dwconv_mfma_kernel (depthwise forward: 8 Toeplitz-MFMA launches + the stride-2 first layer per step, "
                                              "rows <= 512 outputs on the streaming kernel, time-stretched longer rows on the general one; "
                                              + ("hidden activations stored as bf16: algorithmic bytes at 2 B/sample)" if act16 else "fp32 activations)"),
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None, "traffic_source": None, "traffic_per_launch": None,
                    "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2),
                    "algorithmic_bytes_per_launch": round(nbytes / n), "algorithmic_bytes_nominal_step": dw_bytes,
                    "measured_copy_gbs": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                    "copy_probe_gbs": round(probe_gbs, 1), "frac_of_copy_probe": round(achieved / probe_gbs, 4),
                    "pattern_copy_gbs": round(pat_gbs, 1) if pat_gbs else None,
                    "frac_of_pattern_copy": round(achieved / pat_gbs, 4) if pat_gbs else None,
                    "note": ("ceiling of this layout: a pure copy with the kernel's own access pattern ([B][C][T] bf16 rows, one workgroup per "
                             "channel, one row per wave, measured live = pattern_copy_gbs) reaches "
                             + (f"{pat_gbs / HBM_PEAK_GBS:.2f}" if pat_gbs else "0.56-0.64") + " of 8 TB/s; the 0.60 target needs channel-major "
                             "hidden tensors, which cost the K <= 512 GEMMs more than the depthwise kernels gain (profiles/r04_layout_ab.txt)")}
            # `traffic` only when the PMC file lists FETCH_SIZE / WRITE_SIZE for EVERY forward launch the nominal step dispatches
            # (one row per layer, matched by kernel size): the ratio is the bytes-weighted mean over those rows
            rows = (pmc or {}).get("launches") or []
            _, per_layer = encoder_dw_bytes(B_PER_GPU, T_FRAMES)
            ks = [11, 19, 27, 35, 51, 59, 67, 75, 83]
            byk = {r.get("k"): r for r in rows if r.get("hbm_bytes") and r.get("algorithmic_bytes")}
            if pmc and all(k in byk for k in ks):
                w_ratio = sum(byk[k]["hbm_bytes"] for k in ks) / sum(byk[k]["algorithmic_bytes"] for k in ks)
                roof["traffic"] = round(w_ratio * nbytes / n)
                roof["traffic_source"] = (f"{w_ratio:.4f} x algorithmic bytes: bytes-weighted over the 9 forward launches of a nominal step, each "
                                          f"with its own FETCH_SIZE / WRITE_SIZE rows (rocprofv3 PMC over bench.py itself, separate passes, "
                                          f"2*FETCH_SIZE + WRITE_SIZE; profiles/{os.path.basename(PMC_FILE)})")
                roof["traffic_per_launch"] = [{"k": k, "kernel": byk[k]["kernel"], "ratio": round(byk[k]["hbm_bytes"] / byk[k]["algorithmic_bytes"], 4)} for k in ks]
            if "dw_fwd" in kt_nom:               # the nominal step's nine launches alone (separate pass, time-stretch off): review item 3c
                nn, nms, nby = kt_nom["dw_fwd"]
                roof["frac_nominal_step"] = round(nby / (nms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                # the depthwise FAMILY (round-5 review, item 1d): forward + fused backward (backward-data + the weight gradient inside
                # it; the stride-2 opener's separate kernels included), algorithmic bytes of both directions / their summed time
                def _fam(t):
                    keys = [k for k in ("dw_fwd", "dw_bwd_data", "dw_wgrad") if k in t]
                    ms_ = sum(t[k][1] for k in keys)
                    return (sum(t[k][2] for k in keys) / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_ > 0 and "dw_bwd_data" in t else None
                ff, ffn = _fam(kt_all_raw), _fam(kt_nom)
                roof["frac_family"] = round(ff, 4) if ff else None
                roof["frac_family_nominal_step"] = round(ffn, 4) if ffn else None
                if "dw_bwd_data" in kt_nom:
                    bn_, bms_, bby_ = kt_nom["dw_bwd_data"]
                    roof["frac_bwd_nominal_step"] = round(bby_ / (bms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                roof["frac_note"] = ("frac: every forward launch of the timed region (~30 % of the seeded steps are time-stretched: longer rows on the "
                                     "three-tile / general kernels); frac_nominal_step: the nine launches of a 1024-frame step, timed in a separate pass; "
                                     "frac_family: forward + backward depthwise launches together (algorithmic bytes of both / their time) over the seeded "
                                     "sequence, frac_family_nominal_step: on the 1024-frame step")
        # Step-level roofline (round-4 review, item 2): algorithmic bytes and 1x1-GEMM flops of ONE nominal step from the shape model
        # (tools/step_model.py: every operand read once, every result written once, in the step's storage formats), the kernel time of a
        # nominal step measured in THIS run (HIP events in the dispatch packets of every library launch), and the HBM bytes rocprofv3's
        # counters saw for the same step (profiles/step_pmc.json, quoted only for the kernel sources it was measured on).
        roof_step = None
        try:
            from tools import step_model
            fam = step_model.by_family(step_model.step_rows(B_PER_GPU, T_FRAMES))
            b_alg = sum(f["bytes"] for f in fam.values())
            fl = sum(f["flops"] for f in fam.values())
            k_ms = sum(v[1] for v in kt_nom.values()) if kt_nom else None
            fam_ms = {k: v[1] for k, v in kt_nom.items()}
            spmc = None
            sp = os.path.join(ROOT, "profiles", "step_pmc.json")
            if os.path.exists(sp):
                spmc = json.load(open(sp))
                if spmc.get("csrc_sha") != csrc_sha():
                    spmc = None
            roof_step = {"bytes_algorithmic": round(b_alg), "bytes_measured": (spmc or {}).get("bytes_measured_per_step"),
                         "kernel_ms": round(k_ms, 3) if k_ms else None,
                         "achieved_gbs": round(b_alg / (k_ms * 1e-3) / 1e9, 1) if k_ms else None,
                         "frac_of_8TBs": round(b_alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms else None,
                         "hbm_floor_ms": round(b_alg / (HBM_PEAK_GBS * 1e9) * 1e3, 3), "mfma_floor_ms": round(fl / 2.5e15 * 1e3, 3),
                         "gemm_tflops": round(fl / 1e12, 3),
                         "launches": round(sum(v[0] for v in kt_nom.values()), 1) if kt_nom else None,
                         "note": "nominal step (B = 32 x T = 1024, time-stretch off), library launches only; bytes_algorithmic / flops: tools/step_model.py; "
                                 "bytes_measured: rocprofv3 PMC 2*FETCH_SIZE + WRITE_SIZE over every kernel of the step (profiles/step_pmc.json, "
                                 "null when the kernel sources changed since); kernel_ms: HIP events of every launch in this run",
                         "families_ms": {k: round(v, 3) for k, v in fam_ms.items()},
                         "families_algorithmic_mb": {k: round(f["bytes"] / 1e6, 1) for k, f in fam.items()}}
        except Exception as e:                                       # noqa: BLE001
            roof_step = {"error": f"{type(e).__name__}: {e}"}
        out = {
            "metric": "audio frames/sec (fwd+bwd) asr_en_base, B=32x1024-frame mel" + (" [DIAGNOSTIC: time-stretch off]" if args.diag_no_timestretch else "") + (f" [DIAGNOSTIC: stretch {args.diag_stretch_rate}% every step]" if args.diag_stretch_rate else ""),
            "value": round(frames / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "world_size": world,
            "collective": (f"RCCL {'.'.join(str(v) for v in torch.cuda.nccl.version())} all-reduce of 46.5 MB fp32 gradients, "
                           "16 MB buckets overlapped with backward") if world > 1 else None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: asr_en_base (AudioToTextCTC 64/512/29/512) training step, "
                                   "batch=32 x 1024-frame synthetic log-mel + CTC targets [32,100] per GPU, "
                                   "augmentation+dropout on, Adam; 1x1 GEMM operands bf16 (fp32 accumulate); activation storage level "
                                   + str(F_.get_activation_storage()) + " (0 fp32 everywhere; >= 1 the tensors internal to a block -- and their "
                                   "gradients from 2 -- stored as bf16; 4 adds a bf16 copy of each block output as the next block's GEMM "
                                   "operand; 5: the forward residual stream inside a stack lives in that bf16 copy only, as under the "
                                   "reference's autocast); gradients between blocks, BatchNorm statistics and all accumulation fp32"
                                   if args.precision == "bf16" else
                                   "configs[1] at fp32 throughout",
                       "global_batch": B_PER_GPU * world, "frames_per_utterance": T_FRAMES, "parallelism": f"dp{world}"},
            "loss": round(float(loss), 4),
            # host time to enqueue a step (the loop without the final synchronise): the step is GPU-bound while this stays below ms_per_step
            "host_enqueue_ms_per_step": round(host_loop / args.steps * 1e3, 3),
            "host_under_contention": contended,
            "launches_per_step": {"value": round(launches, 1), "source": "v100_launch_count() over the timed steps of this run (library launches; the "
                                                                      "time-stretched steps of the seeded sequence take other kernel variants, not other counts)"},
            "windows_ms_per_step": ({"n": len(windows), "median": round(float(np.median(windows)), 3), "min": round(min(windows), 3),
                                     "max": round(max(windows), 3)} if windows else None),
            "roofline": roof,
            "roofline_step": roof_step,
            "kernel_ms_per_step": kt_all,
            "kernel_ms_method": "dispatch-packet timestamps, all timed launches of 10 steps of the seeded sequence (r04 and earlier: event markers "
                                "around each 1x1-GEMM call, ~+0.2 ms on pw_*)",
            "sustained": sustained,
            "fp32_ms_per_step": fp32_line,
            "dp_path_single_rank": dp_line,
            "launch_graph_diagnostic": graph_diag,
        }
        # the extras must never cost the primary line: any failure in them is reported inside the JSON instead
        if world == 1 and not args.no_other_configs and args.precision == "bf16":
            try:
                out["other_configs"] = other_configs(device, N, F_)
            except Exception as e:                                   # noqa: BLE001
                out["other_configs"] = {"error": f"{type(e).__name__}: {e}"}
        out["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:                                   # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        # ONE line on stdout, kept SHORT (a harness that keeps only a few KB of a run's tail must still see "metric" and "value" at its
        # start): every contract key, the roofline and cpu_baseline objects with their figures, the extras as figures only.  The full
        # record (per-launch PMC rows, prose provenance of every number, every key of the other configs) goes to stderr and, where the
        # directory is writable, to gpurun_out/bench_details.json.
        sys.stderr.write("bench details: " + json.dumps(out) + "\n")
        try:
            os.makedirs("gpurun_out", exist_ok=True)
            with open(os.path.join("gpurun_out", "bench_details.json"), "w") as f:
                json.dump(out, f)
        except OSError:
            pass
        json_out.write(json.dumps(lean_line(out)) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
