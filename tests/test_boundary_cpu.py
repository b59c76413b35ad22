"""The drop-in boundary beyond forward(): graph recording (torch.jit.trace / ONNX lowering) and the
LightningModule surface (hparams, load_from_checkpoint) -- SURVEY.md 8b, reference call sites
voice100/export_onnx_v1.py:35-57, 60-84, 96-125 and voice100/models/_base.py:3-7.  CPU only: a trace runs
the in-package stock-op restatement (voice100_amd/_stock.py), which is checked here against the oracle
and the reference-generated goldens."""
import io
import warnings

import numpy as np
import pytest
import torch
from torch import nn

from conftest import load_golden, sub, rel_err
from oracle import cnn
from voice100_amd._base import Voice100ModelBase, checkpoint_dict, save_checkpoint, HAVE_LIGHTNING
from voice100_amd.asr import AudioToTextCTC
from voice100_amd.tts import AlignTextToAudioModel, TextToAlignTextModel

TOL = 1e-5


def _state(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def _randomize_bn(m, seed=3):
    """Non-trivial running statistics / affine terms so a wrong BN fold cannot hide behind mean 0 / var 1."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm1d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.2)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)


class AlignTextToAudioPredictModel(nn.Module):
    """The wrapper export_onnx_v1.py:87-93 puts around AlignTextToAudioModel.predict."""

    def __init__(self, model) -> None:
        super().__init__()
        self.model = model

    def forward(self, aligntext):
        return self.model.predict(aligntext)


def test_eager_cpu_forward_still_refuses():
    """Outside a trace there is no CPU path: the product fails loudly (no silent fallback)."""
    m = AudioToTextCTC(64, 32, 29, 32).eval()
    with pytest.raises(RuntimeError):
        m(torch.rand(1, 20, 64))


def test_trace_asr_matches_golden_and_oracle():
    """torch.jit.trace of the eval model with the exporter's dummy input (export_onnx_v1.py:41), replayed on the
    golden's batch (other batch size and length: the graph is shape-agnostic) -> reference logits and token ids."""
    g = load_golden("asr_tiny.npz")
    state = sub(g, "state/")
    hid = state["encoder.layers.8.conv.2.weight"].shape[0]
    hidden = state["encoder.layers.4.conv.2.weight"].shape[0]
    vocab = state["decoder.layers.1.weight"].shape[0]
    m = AudioToTextCTC(64, hid, vocab, hidden)
    m.load_state_dict(state, strict=True)
    m.eval()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traced = torch.jit.trace(m, torch.rand(size=[1, 100, 64], dtype=torch.float32), check_trace=False)   # the check
        # re-runs the module OUTSIDE the trace, i.e. on the HIP path: GPU only (tests/test_gpu_models.py does that)
    audio = torch.from_numpy(g["audio"])
    logits = traced(audio)
    assert rel_err(logits, g["logits_eval"]) < TOL
    assert np.array_equal(logits.argmax(-1).numpy(), g["argmax_eval"])
    assert rel_err(logits, cnn.audio_to_text_ctc_forward(audio, state, training=False)) < TOL
    kinds = {n.kind() for n in traced.graph.nodes()} | {n.kind() for n in traced.inlined_graph.nodes()}
    assert not any("voice100" in k or "PythonOp" in k for k in kinds)


@pytest.mark.parametrize("use_mcep", [False, True])
def test_trace_tts_predict_matches_golden(use_mcep):
    g = load_golden("tts_tiny_mcep.npz" if use_mcep else "tts_tiny_logspc.npz")
    state = sub(g, "state/")
    vocab, hidden = state["embedding.weight"].shape
    m = AlignTextToAudioModel(vocab, hidden, 1e-3, use_mcep=use_mcep)
    m.load_state_dict(state, strict=True)
    m.eval()
    wrapped = AlignTextToAudioPredictModel(m)
    aligntext = torch.from_numpy(g["aligntext"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traced = torch.jit.trace(wrapped, torch.randint(low=0, high=vocab, size=(5, 100)), check_trace=False)
    f0, logspc, codeap = traced(aligntext)
    ref = cnn.align_text_to_audio_predict(aligntext, state)
    for got, want in zip((f0, logspc, codeap), ref):
        assert rel_err(got, want) < TOL
    assert np.array_equal((f0 == 0).numpy(), (ref[0] == 0).numpy())          # F0 gate: exact
    for k, got in (("predict/f0", f0), ("predict/logspc", logspc), ("predict/codeap", codeap)):
        assert rel_err(got, g[k]) < TOL


def test_trace_text_to_align_text():
    g = load_golden("align_tiny.npz")
    state = sub(g, "state/")
    vocab, hidden = state["embedding.weight"].shape
    m = TextToAlignTextModel(vocab_size=vocab, hidden_size=hidden, learning_rate=1e-3)
    m.load_state_dict(state, strict=True)
    m.eval()
    text = torch.from_numpy(g["text"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traced = torch.jit.trace(m, torch.randint(low=0, high=vocab, size=(5, 100)), check_trace=False)
    assert rel_err(traced(text), g["pred_eval"]) < TOL
    assert rel_err(traced(text), cnn.text_to_align_text_forward(text, state, training=False)) < TOL


def _onnx_graph(model, args, input_names, output_names, dynamic_axes):
    """Lower to ONNX opset 13 exactly as torch.onnx.export would (export_onnx_v1.py:43-57).  The last step of export --
    serialising the protobuf -- needs the `onnx` wheel, which this image lacks; everything before it (tracing under
    is_in_onnx_export(), symbolic lowering, dynamic axes) runs here and yields the ONNX graph."""
    try:
        import onnx  # noqa: F401
        f = io.BytesIO()
        torch.onnx.export(model, args, f, export_params=True, opset_version=13, do_constant_folding=True, dynamo=False,
                          input_names=input_names, output_names=output_names, dynamic_axes=dynamic_axes)
        return [n.op_type for n in onnx.load_from_string(f.getvalue()).graph.node]
    except ImportError:
        from torch.onnx._internal.torchscript_exporter import utils as U
        from torch.onnx._internal.torchscript_exporter._globals import GLOBALS
        GLOBALS.export_onnx_opset_version = 13
        with U.exporter_context(model, torch.onnx.TrainingMode.EVAL, False):
            graph, params, _ = U._model_to_graph(model, args, input_names=input_names, output_names=output_names,
                                                 dynamic_axes=dynamic_axes, do_constant_folding=True)
        kinds = [n.kind() for n in graph.nodes()]
        assert all(k.startswith("onnx::") for k in kinds), kinds
        assert params, "export_params: weights must travel with the graph"
        return [k[len("onnx::"):] for k in kinds]


def test_onnx_lowering_asr_and_tts():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = AudioToTextCTC(64, 32, 29, 32).eval()
        ops = _onnx_graph(m, (torch.rand(size=[1, 100, 64]),), ["audio"], ["logits"],
                          {"audio": {0: "batch_size", 1: "audio_len"}, "logits": {0: "batch_size", 1: "logits_len"}})
        assert ops.count("Conv") == 28 and "Transpose" in ops          # 9 blocks x 3 convs (BN folded) + the char decoder
        t = AlignTextToAudioPredictModel(AlignTextToAudioModel(29, 32, 1e-3).eval())
        ops = _onnx_graph(t, (torch.randint(low=0, high=29, size=(5, 100)),), ["aligntext"], ["f0", "logspc", "codeap"],
                          {"aligntext": {0: "batch_size", 1: "aligntext_len"}, "f0": {0: "batch_size", 1: "audio_len"},
                           "logspc": {0: "batch_size", 1: "audio_len"}, "codeap": {0: "batch_size", 1: "audio_len"}})
        assert "ConvTranspose" in ops and "Gather" in ops and "Where" in ops


@pytest.mark.parametrize("cls,kw", [
    (AudioToTextCTC, dict(audio_size=64, embed_size=32, vocab_size=29, hidden_size=32, learning_rate=2e-3, weight_decay=1e-5)),
    (TextToAlignTextModel, dict(vocab_size=29, hidden_size=32, learning_rate=3e-3)),
    (AlignTextToAudioModel, dict(vocab_size=29, hidden_size=32, learning_rate=4e-3, use_mcep=True)),
])
def test_checkpoint_round_trip(tmp_path, cls, kw):
    """save_hyperparameters() -> Lightning-layout checkpoint -> load_from_checkpoint (export_onnx_v1.py:38,61,99)."""
    torch.manual_seed(11)
    m = cls(**kw)
    _randomize_bn(m)
    assert isinstance(m, Voice100ModelBase)
    assert dict(m.hparams) == kw
    assert m.hparams["vocab_size"] == 29 and m.hparams.learning_rate == kw["learning_rate"]
    ckpt = checkpoint_dict(m)
    assert set(ckpt) >= {"state_dict", "hyper_parameters", "pytorch-lightning_version"}
    path = tmp_path / "last.ckpt"
    if HAVE_LIGHTNING:
        torch.save(ckpt, path)
    else:
        save_checkpoint(m, path)
    m2 = cls.load_from_checkpoint(str(path))
    assert dict(m2.hparams) == kw
    s1, s2 = m.state_dict(), m2.state_dict()
    assert list(s1) == list(s2)
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    if cls is AudioToTextCTC:
        opt = m2.configure_optimizers()["optimizer"]
        assert opt.defaults["lr"] == 2e-3 and opt.defaults["weight_decay"] == 1e-5


def test_reference_keyed_checkpoint_loads_strict(tmp_path):
    """A checkpoint as the reference's Trainer writes it (golden state_dict + hyper_parameters) loads with strict=True."""
    g = load_golden("asr_tiny.npz")
    state = sub(g, "state/")
    hp = dict(audio_size=64, embed_size=state["encoder.layers.8.conv.2.weight"].shape[0],
              vocab_size=state["decoder.layers.1.weight"].shape[0],
              hidden_size=state["encoder.layers.4.conv.2.weight"].shape[0], learning_rate=1e-3, weight_decay=4e-5)
    path = tmp_path / "ref.ckpt"
    torch.save({"state_dict": state, "hyper_parameters": hp, "pytorch-lightning_version": "1.8.6", "epoch": 3}, path)
    m = AudioToTextCTC.load_from_checkpoint(str(path))
    for k, v in m.state_dict().items():
        assert torch.equal(v.reshape(-1), state[k].reshape(-1)), k


def test_eval_block_with_autograd_has_no_cpu_path_either():
    """An eval-mode block that takes part in autograd runs the differentiable frozen-statistics path -- on the GPU only, like everything
    else: a CPU tensor is refused, not silently computed elsewhere or returned detached."""
    from voice100_amd.layers import InvertedResidual
    blk = InvertedResidual(8, 8, kernel_size=5).eval()
    with pytest.warns(UserWarning, match="frozen-statistics"):
        with pytest.raises(RuntimeError, match="GPU only"):
            blk(torch.rand(1, 8, 16, requires_grad=True))


def test_normalize_matches_reference_formula():
    """AudioToTextCTC.normalize (asr.py:124-131): masked mean / std over time per utterance and feature."""
    torch.manual_seed(2)
    m = AudioToTextCTC(6, 8, 5, 8)
    audio = torch.randn(3, 11, 6) * 2 + 1
    lens = torch.tensor([11, 7, 4])
    out = m.normalize(audio, lens)
    for b, n in enumerate(lens.tolist()):
        seg = audio[b, :n]
        mean = seg.mean(0, keepdim=True)
        std = torch.sqrt(((seg - mean) ** 2).mean(0, keepdim=True))
        assert torch.allclose(out[b, :n], (seg - mean) / (std + 1e-15), atol=1e-5)
        assert torch.count_nonzero(out[b, n:]) == 0


def test_derived_tensor_cache_follows_versions_and_lifetimes():
    """functional._derived (16-bit weight copies, re-laid-out matrices kept per parameter OBJECT): rebuilt when the parameter's version
    counter moves -- which FusedAdam and the BatchNorm-updating forwards advance themselves (functional._touched), since their kernels
    write through raw pointers --, never served to another tensor, dropped with the parameter."""
    import gc
    from voice100_amd import functional as F_
    w = torch.nn.Parameter(torch.zeros(4, 4))
    built = []

    def build():
        built.append(1)
        return object()
    a = F_._derived(w, "t", build)
    assert F_._derived(w, "t", build) is a and len(built) == 1
    assert F_._derived(w, "other tag", build) is not a and len(built) == 2
    F_._touched([w])                                   # what optim.FusedAdam.step does after its kernel
    b = F_._derived(w, "t", build)
    assert b is not a and len(built) == 3
    with torch.no_grad():
        w.mul_(2.0)                                    # any in-place write through torch does the same
    assert F_._derived(w, "t", build) is not b
    w2 = torch.nn.Parameter(torch.zeros(4, 4))
    assert F_._derived(w2, "t", build) is not b        # another object: its own entry
    n = len(F_._DERIVED)
    del w
    gc.collect()
    assert len(F_._DERIVED) == n - 2                   # both tags of the dead parameter are gone


def test_bench_line_stays_under_3kb():
    """bench.py prints ONE JSON line for the driver; the full record goes to stderr / gpurun_out.  The printed form of the last committed
    record (profiles/r05_bench_details.json) must keep every contract key and stay under 3 KB."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    rec = os.path.join(root, "profiles", "r05_bench_details.json")
    if not os.path.exists(rec):
        pytest.skip("no committed bench record")
    line = bench.lean_line(json.load(open(rec)))
    text = json.dumps(line)
    assert len(text) < 3072, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k


def test_trainer_precision_values_of_the_reference_recipes():
    """`--trainer.precision` as the reference's README passes it (README.md:189, 212, 233, 284: `16`; Lightning's other spellings):
    16 maps to "bf16" (INTEGRATION.md section 1), anything else Lightning does not know is refused."""
    from voice100_amd.trainer import resolve_precision
    for v, want in ((32, "fp32"), ("32", "fp32"), ("32-true", "fp32"), (16, "bf16"), ("16", "bf16"), ("16-mixed", "bf16"),
                    ("bf16", "bf16"), ("bf16-mixed", "bf16"), ("BF16", "bf16")):
        assert resolve_precision(v) == want
    for bad in (64, "fp8", None, "half"):
        with pytest.raises(ValueError):
            resolve_precision(bad)
