"""Voice100ModelBase: the base class of the three v1 models (voice100/models/_base.py:3-7).

The reference derives its models from `pytorch_lightning.LightningModule`, and its callers rely on
that surface: `Trainer.fit(model, ...)` (train_asr.py:36-38), `Model.load_from_checkpoint(path)` and
`model.hparams["vocab_size"]` (export_onnx_v1.py:35-57, 60-62, 99-101), `self.save_hyperparameters()`,
`self.log_dict(...)` (asr.py:99, 156-165).

* pytorch_lightning importable  -> the base IS `pl.LightningModule`; nothing here replaces it.
* pytorch_lightning missing     -> a small nn.Module shim with the same members, so the modules,
  the exporter flow and voice100_amd.trainer.TrainStep work without it.  Checkpoints use
  Lightning's layout (`state_dict`, `hyper_parameters`, `pytorch-lightning_version`), so a file
  written by either side loads on the other.
"""
import inspect
from typing import Any, Dict, Optional

import torch
from torch import nn

try:                                            # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    _LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except ImportError:
    pl = None
    _LightningModule = None
    HAVE_LIGHTNING = False

CHECKPOINT_HPARAMS_KEY = "hyper_parameters"     # pytorch_lightning.LightningModule.CHECKPOINT_HYPER_PARAMS_KEY


class AttributeDict(dict):
    """dict with attribute access: `hparams.learning_rate` (asr.py:172) and `hparams["vocab_size"]`
    (export_onnx_v1.py:62) both work, like Lightning's own AttributeDict."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(f"Missing attribute '{key}'") from None

    def __setattr__(self, key, value):
        self[key] = value


class _ShimModule(nn.Module):
    """The LightningModule members the reference's v1 code path touches, on a plain nn.Module."""

    CHECKPOINT_HYPER_PARAMS_KEY = CHECKPOINT_HPARAMS_KEY

    def __init__(self):
        super().__init__()
        self._hparams = AttributeDict()
        self._logged: Dict[str, Any] = {}
        self.trainer = None

    # -- hyper-parameters -------------------------------------------------------------------------
    @property
    def hparams(self) -> AttributeDict:
        return self._hparams

    def save_hyperparameters(self, *names, ignore=()) -> None:
        """Record the calling __init__'s arguments (all of them, or those named) as `self.hparams`."""
        frame = inspect.currentframe().f_back
        args = inspect.getargvalues(frame)
        init_args = {k: args.locals[k] for k in args.args if k != "self"}
        if args.keywords:
            init_args.update(args.locals[args.keywords])
        if names:
            init_args = {k: v for k, v in init_args.items() if k in names}
        ignore = (ignore,) if isinstance(ignore, str) else tuple(ignore)
        for k, v in init_args.items():
            if k not in ignore:
                self._hparams[k] = v

    # -- logging (metrics are kept so a step loop can read them; no logger backend) ----------------
    def log(self, name: str, value, *args, **kwargs) -> None:
        self._logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

    def log_dict(self, metrics: Dict[str, Any], *args, **kwargs) -> None:
        for k, v in metrics.items():
            self.log(k, v)

    @property
    def logged_metrics(self) -> Dict[str, Any]:
        return self._logged

    # -- checkpoints ------------------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict: bool = True, **kwargs):
        """Build the model from the checkpoint's `hyper_parameters` (overridable by kwargs) and load
        its `state_dict` -- the call export_onnx_v1.py:38 / :61 / :99 makes."""
        ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
        return _load_state(cls, ckpt, strict, **kwargs)

    def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        pass

    def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        pass


def _load_state(cls, ckpt: Dict[str, Any], strict: bool = True, **kwargs):
    hp = dict(ckpt.get(CHECKPOINT_HPARAMS_KEY, {}))
    hp.update(kwargs)
    sig = inspect.signature(cls.__init__)
    if not any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values()):
        hp = {k: v for k, v in hp.items() if k in sig.parameters}
    model = cls(**hp)
    model.on_load_checkpoint(ckpt)
    model.load_state_dict(ckpt["state_dict"], strict=strict)
    return model


def checkpoint_dict(model: nn.Module, epoch: int = 0, global_step: int = 0) -> Dict[str, Any]:
    """A Lightning-layout checkpoint of `model` (what `Trainer.save_checkpoint` writes, minus optimizer state)."""
    ckpt = {
        "epoch": epoch, "global_step": global_step,
        "pytorch-lightning_version": pl.__version__ if HAVE_LIGHTNING else "1.8.6",   # the reference's pin (poetry.lock)
        "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
        CHECKPOINT_HPARAMS_KEY: dict(model.hparams),
    }
    model.on_save_checkpoint(ckpt)
    return ckpt


def save_checkpoint(model: nn.Module, path, epoch: int = 0, global_step: int = 0) -> None:
    torch.save(checkpoint_dict(model, epoch, global_step), path)


Voice100ModelBase = _LightningModule if HAVE_LIGHTNING else _ShimModule


def tracing() -> bool:
    """True while torch.jit.trace / torch.onnx.export records the graph: the modules then run their stock-op
    restatement (voice100_amd/_stock.py) so the exporter sees plain aten/onnx ops instead of opaque HIP calls."""
    return torch.jit.is_tracing() or torch.onnx.is_in_onnx_export()
