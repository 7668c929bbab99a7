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


# ----------------------------------------------------------------------------------------------
# PNG formats of the evaluation path.  The reference writes them through pypng / scipy.misc
# (utils/flow.py:37-62, runtime.py:318-343), neither of which is a dependency here: a PNG is a zlib stream of
# filter-0 scanlines between three chunks, so the codec below is ~40 lines of struct + zlib.
# ----------------------------------------------------------------------------------------------
import struct  # noqa: E402
import zlib  # noqa: E402

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _png_chunk(kind: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)


def write_png(filename: str, img: np.ndarray) -> None:
    """(H, W, 3) uint8 or uint16 array -> RGB PNG (bit depth 8 / 16, no interlace, filter type 0)."""
    if img.ndim != 3 or img.shape[2] != 3 or img.dtype not in (np.uint8, np.uint16):
        raise ValueError("expected an (H, W, 3) uint8 / uint16 array")
    h, w, _ = img.shape
    depth = 8 if img.dtype == np.uint8 else 16
    rows = img.astype(">u2" if depth == 16 else np.uint8).reshape(h, -1).view(np.uint8)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()          # filter byte 0 per scanline
    with open(filename, "wb") as f:
        f.write(_PNG_SIG)
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 2, 0, 0, 0)))
        f.write(_png_chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(_png_chunk(b"IEND", b""))


def read_png(filename: str) -> np.ndarray:
    """RGB PNG (8 / 16 bit, non-interlaced) -> (H, W, 3) uint8 / uint16; all five scanline filters are supported."""
    data = open(filename, "rb").read()
    if data[:8] != _PNG_SIG:
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if ctype != 2 or depth not in (8, 16) or interlace:
        raise ValueError("only non-interlaced 8/16-bit RGB PNGs are supported")
    bpp = 3 * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:                                             # sub / average / paeth need the running left neighbour
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    if depth == 8:
        return out.reshape(h, w, 3)
    return out.view(">u2").astype(np.uint16).reshape(h, w, 3)


def write_flow_png(filename: str, uv: np.ndarray, v: np.ndarray = None, mask: np.ndarray = None) -> None:
    """KITTI flow PNG (utils/flow.py:37-62): 16-bit RGB = (u*64 + 2^15, v*64 + 2^15, valid), values clipped to uint16."""
    if v is None:
        if uv.ndim != 3 or uv.shape[2] != 2:
            raise ValueError("expected an (H, W, 2) array")
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u = uv
    if u.shape != v.shape:
        raise ValueError("u and v must have the same shape")
    valid = np.ones(u.shape) if mask is None else mask
    fu = np.clip(u * 64 + 2 ** 15, 0.0, 65535.0).astype(np.uint16)
    fv = np.clip(v * 64 + 2 ** 15, 0.0, 65535.0).astype(np.uint16)
    write_png(filename, np.stack((fu, fv, valid.astype(np.uint16)), axis=-1))


def read_png_flow(filename: str):
    """KITTI flow PNG -> ((H, W, 2) float64 flow, (H, W, 1) validity) -- datasets/kitti_combined.py:19-34."""
    img = read_png(filename).astype(np.float64)
    invalid = img[:, :, 2] == 0
    flow = (img[:, :, 0:2] - 2 ** 15) / 64.0
    flow[invalid] = 0
    return flow, (1 - invalid * 1)[:, :, None]


def make_color_wheel() -> np.ndarray:
    """Middlebury colour wheel: 55 hues over the six segments RY, YG, GC, CB, BM, MR (utils/flow.py:124-172)."""
    segs = [(15, (255, None, 0)), (6, (-1, 255, 0)), (4, (0, 255, None)), (11, (0, -1, 255)), (13, (None, 0, 255)), (6, (255, 0, -1))]
    wheel = np.zeros((sum(n for n, _ in segs), 3))
    col = 0
    for n, spec in segs:
        ramp = np.floor(255 * np.arange(n) / n)
        for ch, s in enumerate(spec):
            wheel[col:col + n, ch] = ramp if s is None else (255 - ramp if s == -1 else s)
        col += n
    return wheel


def flow_to_png_middlebury(flow: np.ndarray) -> np.ndarray:
    """(2, H, W) flow -> (H, W, 3) uint8 Middlebury colour coding, normalised by the largest magnitude
    (utils/flow.py:80-121,175-210)."""
    u, v = flow[0].copy(), flow[1].copy()                   # the normalising radius is taken in the INPUT precision,
    unknown = (np.abs(u) > 1e7) | (np.abs(v) > 1e7)         # as the reference does (utils/flow.py:197-203)
    u[unknown] = 0
    v[unknown] = 0
    maxrad = max(-1, np.max(np.sqrt(u ** 2 + v ** 2)))
    u = u / (maxrad + np.finfo(float).eps)
    v = v / (maxrad + np.finfo(float).eps)
    nan = np.isnan(u) | np.isnan(v)
    u[nan] = 0
    v[nan] = 0
    wheel = make_color_wheel()
    ncols = wheel.shape[0]
    rad = np.sqrt(u ** 2 + v ** 2)
    fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (ncols - 1) + 1
    k0 = np.floor(fk).astype(int)
    k1 = k0 + 1
    k1[k1 == ncols + 1] = 1
    f = fk - k0
    img = np.zeros(u.shape + (3,))
    for i in range(3):
        col = (1 - f) * (wheel[k0 - 1, i] / 255) + f * (wheel[k1 - 1, i] / 255)
        small = rad <= 1
        col[small] = 1 - rad[small] * (1 - col[small])
        col[~small] *= 0.75
        img[:, :, i] = np.uint8(np.floor(255 * col * (1 - nan)))
    img[np.repeat(unknown[:, :, None], 3, axis=2)] = 0
    return np.uint8(img)


def save_outputs(args, example_dict, output_dict) -> list:
    """EvaluationEpoch.save_outputs (runtime.py:276-343): writes, per sample, ``<save>/img/[basedir/]<basename>_flow.png``
    (+ ``_occ.png``, ``_flow_b.png``, ``_occ_b.png``) and ``<save>/flo/[basedir/]<basename>.flo`` / ``.png`` according to
    ``args.save_result_{img,flo,png,occ,bidirection}``.  Returns the list of files written."""
    written = []
    flow_f = output_dict["flow"].detach().float().cpu().numpy()
    bidir = bool(getattr(args, "save_result_bidirection", False))
    flow_b = output_dict["flow_b"].detach().float().cpu().numpy() if bidir else None
    b_size = flow_f.shape[0]
    occ = occ_b = None
    if getattr(args, "save_result_occ", False):
        def occ_img(t):
            p = torch.sigmoid(t.detach().float()).expand(-1, 3, -1, -1).cpu().numpy().transpose(0, 2, 3, 1)
            return np.uint8(np.round(p) * 255)
        occ = occ_img(output_dict["occ"])
        if bidir:
            occ_b = occ_img(output_dict["occ_b"])
    names_img, names_flo = [], []
    for ii in range(b_size):
        sub = (example_dict["basedir"][ii] + "/") if "basedir" in example_dict else "/"
        names_img.append(args.save + "/img/" + sub + str(example_dict["basename"][ii]))
        names_flo.append(args.save + "/flo/" + sub + str(example_dict["basename"][ii]))
        for n in (names_img[-1], names_flo[-1]):
            os.makedirs(os.path.dirname(n), exist_ok=True)

    def put(fn, writer, *a):
        writer(fn, *a)
        written.append(fn)

    if getattr(args, "save_result_img", False):
        for ii in range(b_size):
            if occ is not None:
                put(names_img[ii] + "_occ.png", write_png, occ[ii])
                if bidir:
                    put(names_img[ii] + "_occ_b.png", write_png, occ_b[ii])
            put(names_img[ii] + "_flow.png", write_png, flow_to_png_middlebury(flow_f[ii]))
            if bidir:
                put(names_img[ii] + "_flow_b.png", write_png, flow_to_png_middlebury(flow_b[ii]))
    for ii in range(b_size):
        hwc = flow_f[ii].transpose(1, 2, 0)
        if getattr(args, "save_result_flo", False):
            put(names_flo[ii] + ".flo", write_flow, hwc)
        if getattr(args, "save_result_png", False):
            put(names_flo[ii] + ".png", write_flow_png, hwc)
    return written
