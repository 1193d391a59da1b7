"""Checkpoint interop for the module on the hot path.

The reference writes two dialects and its callers pick by model type
(``/root/reference/gnnepcsaft/demo/utils.py:42-50``, ``evaluate_ensemble.py:75-76``, ``train.py:172-173``):

* legacy dict of ``train/utils.py:109-119``: ``{"model_state_dict", "optimizer_state_dict", "scaler_state_dict",
  "step"}`` with bare ``PNAPCSAFT`` keys (``convs.0.lin.weight`` ...);
* Lightning ``.ckpt``: ``{"state_dict", "optimizer_states", "lr_schedulers", "global_step", "epoch", ...}`` whose
  ``state_dict`` keys carry the ``model.`` prefix of ``PNApcsaftL.model``.

``load_checkpoint`` accepts either dialect for either module type.  Files are read with
``torch.load(weights_only=True)`` only: nothing in a checkpoint is executed.
"""

from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch

from .models import PNAPCSAFT, PNApcsaftL

_PREFIX = "model."


def _read(path_or_dict) -> Dict[str, Any]:
    if isinstance(path_or_dict, dict):
        return path_or_dict
    return torch.load(os.fspath(path_or_dict), map_location="cpu", weights_only=True)


def extract_state_dict(ckpt: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    """The PNAPCSAFT state_dict (bare keys) held by a checkpoint of either dialect, or by a bare state_dict."""
    if "model_state_dict" in ckpt:
        sd = ckpt["model_state_dict"]
    elif "state_dict" in ckpt:
        sd = ckpt["state_dict"]
    else:
        sd = ckpt
    if not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError("checkpoint holds neither 'model_state_dict', 'state_dict' nor a bare state_dict")
    if sd and all(k.startswith(_PREFIX) for k in sd):
        sd = {k[len(_PREFIX):]: v for k, v in sd.items()}
    return dict(sd)


def load_checkpoint(model, path_or_dict, strict: bool = True) -> Dict[str, Any]:
    """Loads the weights of either dialect into a ``PNAPCSAFT`` or ``PNApcsaftL``; returns the checkpoint dict
    (optimizer / scheduler / step entries untouched, for ``resume``)."""
    ckpt = _read(path_or_dict)
    sd = extract_state_dict(ckpt)
    target = model.model if isinstance(model, PNApcsaftL) else model
    if not isinstance(target, PNAPCSAFT):
        raise TypeError("load_checkpoint expects a PNAPCSAFT or PNApcsaftL")
    want = target.state_dict()
    for k, v in sd.items():        # float64 checkpoints (evaluate_ensemble.py:68 casts the model) load as float32
        if k in want and v.dtype != want[k].dtype and v.is_floating_point():
            sd[k] = v.to(want[k].dtype)
    target.load_state_dict(sd, strict=strict)
    return ckpt


def lightning_checkpoint(lit: PNApcsaftL, optimizer=None, scheduler=None, global_step: int = 0,
                         epoch: int = 0) -> Dict[str, Any]:
    """The subset of Lightning's ``.ckpt`` layout the reference's loaders read back (``state_dict`` with the
    ``model.`` prefix, ``global_step``, ``epoch``) plus optimizer / scheduler state in Lightning's slots."""
    out: Dict[str, Any] = {
        "state_dict": {k: v.detach().cpu().clone() for k, v in lit.state_dict().items()},
        "global_step": int(global_step),
        "epoch": int(epoch),
    }
    if optimizer is not None:
        out["optimizer_states"] = [_to_cpu(optimizer.state_dict())]
    if scheduler is not None:
        out["lr_schedulers"] = [scheduler.state_dict()]
    return out


def legacy_checkpoint(model: PNAPCSAFT, optimizer=None, step: int = 0) -> Dict[str, Any]:
    """``savemodel`` of train/utils.py:109-119 (the GradScaler slot stays empty: the path is float32)."""
    return {
        "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        "optimizer_state_dict": _to_cpu(optimizer.state_dict()) if optimizer is not None else {},
        "scaler_state_dict": {},
        "step": int(step),
    }


def save_checkpoint(ckpt: Dict[str, Any], path) -> None:
    path = os.fspath(path)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)      # a killed run never leaves a truncated checkpoint under the final name


def resume(ckpt: Dict[str, Any], optimizer=None, scheduler=None) -> int:
    """Restores optimizer / scheduler state of a checkpoint written here; returns the step to continue from."""
    if optimizer is not None:
        if ckpt.get("optimizer_states"):
            optimizer.load_state_dict(ckpt["optimizer_states"][0])
        elif ckpt.get("optimizer_state_dict"):
            optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and ckpt.get("lr_schedulers"):
        scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    return int(ckpt.get("global_step", ckpt.get("step", 0)))


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj
