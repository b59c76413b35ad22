"""Inference side of the hot path: the configs[2] TTS chain on the device, and the pure data-parallel scatter of
inference requests over the GPUs of a node (SURVEY.md 8e "Inference": replicas only, no collective on the data path).

Reference call sites this serves:
  * voice100/update_samples.py:30-90 -- text -> align model -> align() -> audio model predict -> vocoder glue, one batch
    of sample sentences (here: `TTSPipeline`, every step INCLUDING the pyworld synthesis on the GPU: it ends in a waveform);
  * voice100/models/asr.py:110-116 -- AudioToTextCTC.forward over independent utterances / 1-second chunks
    (here: `ASRPipeline`, log-mel -> encoder -> logits -> greedy CTC ids);
  * BASELINE.json configs[4]: "streaming 1-second chunks at 16 kHz, 8 x MI355X" -- chunks are independent, so rank r of
    `world` takes every world-th chunk (`shard_indices(..., "round_robin")`) or a contiguous slice of a request batch,
    runs its replica, and only the small results (token ids, WORLD features) are gathered on rank 0 (`scatter_run`).

Nothing here computes on the CPU: the pipelines call the HIP-backed modules; `scatter_run` is plumbing over
torch.distributed ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests, where a stock-op stand-in model is used).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

__all__ = ["shard_indices", "scatter_run", "TTSPipeline", "ASRPipeline", "GraphedForward"]


def shard_indices(n_items: int, rank: int, world: int, mode: str = "contiguous") -> torch.Tensor:
    """Indices (int64, ascending) of the items rank `rank` of `world` processes.
    "contiguous": one slice per rank, ceil(n / world) items each (a request batch);
    "round_robin": items rank, rank + world, ... (a stream of chunks: every rank stays equally loaded as chunks arrive)."""
    if not 0 <= rank < world:
        raise ValueError("rank must be in [0, world)")
    if mode == "contiguous":
        per = (n_items + world - 1) // world
        lo = min(n_items, rank * per)
        return torch.arange(lo, min(n_items, lo + per), dtype=torch.int64)
    if mode == "round_robin":
        return torch.arange(rank, max(n_items, rank), world, dtype=torch.int64)
    raise ValueError("mode must be 'contiguous' or 'round_robin'")


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def scatter_run(fn: Callable[..., Sequence[torch.Tensor]], inputs: Sequence[torch.Tensor], mode: str = "contiguous",
                group=None, pad_value=0, out_specs=None) -> Optional[List[torch.Tensor]]:
    """Run `fn` data-parallel over dim 0 of `inputs` and gather its outputs on rank 0.

    Every rank holds (or can index) the full `inputs`; rank r calls fn(*[x[idx_r] for x in inputs]) on its shard -- no
    collective touches the data path -- and gets back a sequence of tensors whose dim 0 is the shard.  Outputs may be
    ragged in dim 1 across ranks (aligned-text / WORLD-frame counts differ): they are padded with `pad_value` to the
    global maximum before the gather.  Returns the outputs in the ORIGINAL item order on the group's rank 0 (whatever its
    global rank is) and None elsewhere.  With no process group (or a group of one) it is just fn(*inputs).
    `out_specs` = [(trailing shape, dtype), ...] per output when the caller knows them (fixed-size outputs: token-id
    matrices, per-chunk features): the shapes then need no agreement round, i.e. no object all-gather and no host
    synchronisation per call -- what a timed streaming loop wants."""
    rank, world = _world(group)
    n = inputs[0].shape[0]
    if any(x.shape[0] != n for x in inputs):
        raise ValueError("scatter_run: inputs must share dim 0")
    if world == 1:
        return [o for o in fn(*inputs)]
    idx = shard_indices(n, rank, world, mode)
    per = (n + world - 1) // world                       # every rank pads its shard to `per` items so gathers are regular
    if idx.numel():
        outs = [o for o in fn(*[x[idx.to(x.device)] for x in inputs])]
    else:
        outs = None
    # Shapes and dtypes are agreed through rank 0's view of a rank that has work; ranks without items (n < world) build
    # zero-sized placeholders after learning the trailing shapes.
    if out_specs is not None:
        specs = [(tuple(sh), str(dt)) for sh, dt in out_specs]
        if outs is not None and any(tuple(o.shape[1:]) != sp[0] or str(o.dtype) != sp[1] for o, sp in zip(outs, specs)):
            raise ValueError("scatter_run: fn's outputs do not match out_specs")
        meta = [specs] * world
    else:
        meta = [None] * world
        dist.all_gather_object(meta, None if outs is None else [(tuple(o.shape[1:]), str(o.dtype)) for o in outs], group=group)
    ref = next((m for m in meta if m is not None), None)
    if ref is None:
        return [] if rank == 0 else None
    # dist.gather's dst is a GLOBAL rank: the group's rank 0 may be any process of the job
    dst = dist.get_global_rank(group, 0) if group is not None else 0
    dev = outs[0].device if outs is not None else inputs[0].device
    result = []
    for k, (_, dtname) in enumerate(ref):
        dt = getattr(torch, dtname.replace("torch.", ""))
        trailing = [m[k][0] for m in meta if m is not None]
        nd = len(trailing[0])
        full = tuple(max(t[d] for t in trailing) for d in range(nd))           # pad every ragged trailing dim to its maximum
        buf = torch.full((per,) + full, pad_value, dtype=dt, device=dev)
        if outs is not None:
            o = outs[k]
            buf[(slice(0, o.shape[0]),) + tuple(slice(0, s) for s in o.shape[1:])] = o
        gathered = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, gathered, dst=dst, group=group)
        if rank == 0:
            merged = torch.full((n,) + full, pad_value, dtype=dt, device=dev)
            for r in range(world):
                ir = shard_indices(n, r, world, mode)
                if ir.numel():
                    merged[ir.to(dev)] = gathered[r][:ir.numel()]
            result.append(merged)
    return result if rank == 0 else None


class TTSPipeline:
    """BASELINE configs[2] end to end on the device (update_samples.py:46-84, the pyworld synthesis included):

        text [B, L] int64, text_len [B]
          -> TextToAlignTextModel.forward                 [B, L, 2]  log(gap + 1), log(len + 1)        (tts.py:79-87)
          -> align = max(exp(pred) - 1, 0)                (_align_v2.py:39-46; v1 has no predict(): it trains on log(align + 1),
                                                           tts.py:126.  Negative values are clamped: the reference's Python loop
                                                           would index from the END of the tensor for a negative start)
          -> align() per utterance, on the device         aligntext [B, La] int64, lens                (tts.py:89-110, bit-exact)
          -> AlignTextToAudioModel.predict                f0 [B, 2La-1], logspc|mcep, codeap           (tts.py:192-201)
          -> (use_mcep) logspc = mcep @ mc2sp             one fp32 MFMA GEMM over all B x T frames      (vocoder.py:95)
          -> spc = max(exp(logspc) - 1e-15, 0)                                                         (vocoder.py:99)
          -> ap = decode_aperiodicity(codeap); waveform = synthesize(f0, spc, ap)                      (vocoder.py:100-101; round 4,
                                                           csrc/world.hip -- parity unpinned, see voice100_amd/vocoder.py)

    configs[2] therefore ends in a waveform [B, samples] on the device ("wave", zero beyond each utterance's "wave_len")."""

    def __init__(self, align_model, audio_model, vocoder=None, head: int = 5, tail: int = 5, synthesize: bool = True,
                 f0_ceil: float = 1000.0):
        self.align_model, self.audio_model, self.vocoder = align_model, audio_model, vocoder
        self.head, self.tail = head, tail
        self.synthesize, self.f0_ceil = synthesize, f0_ceil

    @torch.no_grad()
    def __call__(self, text: torch.Tensor, text_len: torch.Tensor):
        from .decode import align_expand
        pred = self.align_model(text)                                        # [B, L, 2]
        align = torch.clamp_min(torch.exp(pred) - 1.0, 0.0)
        aligntext, at_len = align_expand(text, align, text_len, self.head, self.tail)
        # (align_expand sizes the batch exactly as wide as its longest utterance -- pad_sequence semantics, update_samples.py:66 --
        # so that one ends at the convolutions' zero padding, not at extra blank tokens)
        f0, feat, codeap = self.audio_model.predict(aligntext)               # 2 * La - 1 frames
        v = self.vocoder
        if v is not None and v.use_mcep:
            B, T, C = feat.shape
            logspc = v.mcep_to_logspc(feat.reshape(B * T, C)).reshape(B, T, -1)
        else:
            logspc = feat
        spc = v.logspc_to_spc(logspc) if v is not None else None
        # valid WORLD frames per utterance: 2 * len - 1 (update_samples.py:81 slices 2 * len, which yields the same)
        frames = torch.clamp_min(2 * at_len - 1, 0)
        out = {"align": align, "aligntext": aligntext, "aligntext_len": at_len, "f0": f0, "logspc": logspc, "spc": spc,
               "codeap": codeap, "frames": frames}
        if v is not None and self.synthesize and v.n_fft == 512 and f0.shape[1] >= 2:
            wave, npulses = v.synthesize(f0, spc, frames=frames, f0_ceil=self.f0_ceil, codeap=codeap)
            out["wave"], out["n_pulses"] = wave, npulses
            out["wave_len"] = (frames.to(torch.float64) * v.frame_period * v.sample_rate / 1000).to(torch.int64)
        return out


class ASRPipeline:
    """BASELINE configs[4] per replica: waveform chunks [B, N] fp32 (16 kHz) -> log-mel [B, T, 64] -> AudioToTextCTC logits
    -> greedy CTC ids [B, T'] int64 (zero padded) + lengths (data_modules.py:276-291, asr.py:110-116, text.py:99-104)."""

    def __init__(self, model, mel=None):
        self.model, self.mel = model, mel

    @torch.no_grad()
    def __call__(self, wav_or_mel: torch.Tensor):
        from .decode import ctc_greedy_decode
        feats = self.mel(wav_or_mel) if self.mel is not None else wav_or_mel
        ids, n = ctc_greedy_decode(self.model(feats))
        return ids, n


class GraphedForward:
    """One eval-mode forward recorded as a HIP graph (torch.cuda.CUDAGraph) and replayed per call.

    The small shapes are launch-bound, not GPU-bound: configs[0] (AudioToTextCTC on 2 x 256 frames) is ~30 launches of a few microseconds
    each, so the host's per-launch cost IS the latency; a graph submits them as one unit.  Large batches gain nothing (the GPU is the
    bottleneck there) and the training step is not capturable (its augmentation draws upload fresh host data every iteration).

    fn: a callable of CUDA tensors whose shapes never change (a model in eval mode, `model.predict`, a pipeline without host read-backs);
    example inputs fix the shapes.  Calls copy their arguments into the graph's static inputs and return the graph's static outputs --
    OVERWRITTEN by the next call (clone what must survive).  Weights are read in place: an optimizer step or load_state_dict between calls
    is seen by the next replay only if no derived copy is involved -- re-record after changing weights."""

    def __init__(self, fn, *example: torch.Tensor, warmup: int = 3):
        if not example or not all(isinstance(e, torch.Tensor) and e.is_cuda for e in example):
            raise RuntimeError("GraphedForward: example inputs must be CUDA tensors")
        self.static_in = [e.clone() for e in example]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):            # one-time work (derived weight copies, plans, table uploads) happens here, not in the graph
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *args: torch.Tensor):
        if len(args) != len(self.static_in):
            raise TypeError(f"GraphedForward: expected {len(self.static_in)} inputs")
        for dst, a in zip(self.static_in, args):
            if a.shape != dst.shape or a.dtype != dst.dtype:
                raise RuntimeError(f"GraphedForward: recorded for {tuple(dst.shape)} {dst.dtype}, got {tuple(a.shape)} {a.dtype} (record another graph)")
            dst.copy_(a, non_blocking=True)
        self.graph.replay()
        return self.static_out
