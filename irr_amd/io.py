"""Host-side formats either side of the hot path: Middlebury ``.flo`` flow
files and reference-compatible checkpoints.

* ``write_flow`` / ``read_flo_as_float32`` -- byte-compatible with utils/flow.py:11-34 and datasets/common.py:19-27
  (tag 202021.25 as float32, int32 width, int32 height, row-major interleaved (u, v) float32).
* ``CheckpointSaver`` -- same file layout as configuration.py:192-314: ``torch.save({'state_dict': ..., **stats})`` to
  ``<prefix>_latest.ckpt`` (+ ``.json`` stats, optional ``_best`` copy), keys in the ``_model.`` namespace of
  ``ModelAndLoss``; ``restore`` applies fnmatch include/exclude filters.  Unlike the reference it can also carry the
  optimizer state (the reference resumes Adam from scratch, scripts/IRR-PWC_sintel_train.sh:33,62).
"""
from __future__ import annotations

import fnmatch
import json
import os
import shutil
from typing import Iterable, Sequence, Union

import numpy as np
import torch

FLO_TAG = np.array([202021.25], np.float32)


def write_flow(filename: str, uv: np.ndarray, v: np.ndarray = None) -> None:
    if v is None:
        if uv.ndim != 3 or uv.shape[2] != 2:
            raise ValueError("expected an (H, W, 2) array")
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u = uv
    if u.shape != v.shape:
        raise ValueError("u and v must have the same shape")
    h, w = u.shape
    inter = np.empty((h, w, 2), np.float32)
    inter[:, :, 0] = u
    inter[:, :, 1] = v
    with open(filename, "wb") as f:
        FLO_TAG.tofile(f)
        np.array([w, h], np.int32).tofile(f)
        inter.tofile(f)


def read_flo_as_float32(filename: str) -> np.ndarray:
    with open(filename, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != FLO_TAG[0]:
            raise ValueError("Magic number incorrect. Invalid .flo file")
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * h * w)
    if data.size != 2 * h * w:
        raise ValueError("truncated .flo file")
    return data.reshape(h, w, 2)


def flow_tensor_to_flo(filename: str, flow: torch.Tensor) -> None:
    """(2, H, W) or (1, 2, H, W) tensor -> .flo (runtime.py:318-327 writes ``flow.transpose(1, 2, 0)``)."""
    if flow.dim() == 4:
        flow = flow[0]
    write_flow(filename, flow.detach().float().cpu().numpy().transpose(1, 2, 0))


def _filter(keys: Iterable[str], include: Union[str, Sequence[str]] = "*", exclude: Sequence[str] = ()):
    """tools.filter_list_of_strings semantics: union of include patterns minus any exclude pattern."""
    inc = [include] if isinstance(include, str) else list(include)
    exc = [exclude] if isinstance(exclude, str) else list(exclude)
    out = []
    for k in keys:
        if any(fnmatch.fnmatch(k, p) for p in inc) and not any(fnmatch.fnmatch(k, p) for p in exc):
            out.append(k)
    return out


class CheckpointSaver:
    def __init__(self, prefix="checkpoint", latest_postfix="_latest", best_postfix="_best", model_key="state_dict",
                 extension=".ckpt"):
        self._prefix, self._latest, self._best = prefix, latest_postfix, best_postfix
        self._model_key, self._ext = model_key, extension

    def _path(self, directory, postfix, ext=None):
        return os.path.join(directory, self._prefix + postfix + (ext or self._ext))

    def save_latest(self, directory, model_and_loss, stats_dict, store_as_best=False, optimizer_state=None):
        os.makedirs(directory, exist_ok=True)
        save = dict(stats_dict)
        save[self._model_key] = {k: v.detach().cpu() for k, v in model_and_loss.state_dict().items()}
        if optimizer_state is not None:
            save["optimizer_state"] = optimizer_state
        latest = self._path(directory, self._latest)
        torch.save(save, latest)
        with open(self._path(directory, self._latest, ".json"), "w") as f:
            json.dump(stats_dict, f, sort_keys=True, indent=2)
        if store_as_best:
            shutil.copyfile(latest, self._path(directory, self._best))
            shutil.copyfile(self._path(directory, self._latest, ".json"), self._path(directory, self._best, ".json"))
        return latest

    def restore(self, filename, model_and_loss, include_params="*", exclude_params=()):
        if not os.path.isfile(filename):
            raise FileNotFoundError(f"Could not find checkpoint file '{filename}'!")
        ckpt = torch.load(filename, map_location="cpu", weights_only=False)
        state = ckpt[self._model_key]
        keep = set(_filter(state.keys(), include_params, exclude_params))
        own = model_and_loss.state_dict()
        for name, value in state.items():
            if name not in keep:
                continue
            if name not in own:
                raise KeyError(f'unexpected key "{name}" in state_dict')
            if own[name].shape != value.shape:
                raise RuntimeError(f"While copying the parameter named {name}, whose dimensions in the model are "
                                   f"{tuple(own[name].shape)} and whose dimensions in the checkpoint are {tuple(value.shape)}.")
            own[name].copy_(value)
        missing = set(own.keys()) - set(state.keys())
        if missing:
            raise KeyError(f'missing keys in state_dict: "{missing}"')
        stats = {k: v for k, v in ckpt.items() if k != self._model_key}
        return stats, filename

    def restore_latest(self, directory, model_and_loss, include_params="*", exclude_params=()):
        return self.restore(self._path(directory, self._latest), model_and_loss, include_params, exclude_params)

    def restore_best(self, directory, model_and_loss, include_params="*", exclude_params=()):
        return self.restore(self._path(directory, self._best), model_and_loss, include_params, exclude_params)
