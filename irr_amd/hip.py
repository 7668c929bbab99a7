"""ctypes binding of libirr_hip.so (the C ABI declared in include/irr_hip.h).

The argument types are parsed from the header itself, so the header is the single source of truth for
the boundary.  There is NO fallback: if the library is missing, or a call returns non-zero, this raises.
PyTorch is used only to own device memory and streams; every tensor crosses as ``data_ptr()``.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading
from typing import Dict, List, Tuple

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_PKG), "include", "irr_hip.h")
# IRR_HIP_LIB: load another build of the SAME ABI (ablation / trace builds of irr_amd.build with IRR_BUILD_TAG)
LIB_PATH = os.environ.get("IRR_HIP_LIB") or os.path.join(_PKG, "lib", "libirr_hip.so")

ABI_VERSION = 12         # irr_abi_version() of the library this binding was written against (csrc/misc.hip)

_CTYPES = {
    "const float*": ctypes.c_void_p, "float*": ctypes.c_void_p, "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p,
    "const int*": ctypes.c_void_p, "int*": ctypes.c_void_p,
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """-> {function: (return type, [(ctype string, arg name), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"\b(int|long)\s+(irr_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                ty = mm.group(1).strip().replace(" *", "*")
                parsed.append((ty, mm.group(2)))
        out[name] = (ret, parsed)
    return out


class _Lib:
    def __init__(self):
        self._lock = threading.Lock()
        self._lib = None
        self.protos = parse_header()

    def load(self) -> ctypes.CDLL:
        if self._lib is None:
            with self._lock:
                if self._lib is None:
                    if not os.path.exists(LIB_PATH):
                        raise RuntimeError(
                            f"{LIB_PATH} is missing: the HIP extension is the product path and has no fallback. "
                            "Build it with `python -m irr_amd.build` (hipcc, gfx950).")
                    lib = ctypes.CDLL(LIB_PATH)
                    lib.irr_abi_version.restype = ctypes.c_int
                    got = lib.irr_abi_version()
                    if got != ABI_VERSION:
                        # argument types are taken from the HEADER in the tree: a library built from other sources would be
                        # called with the wrong signatures (e.g. double vs float scalars) without any error
                        raise RuntimeError(
                            f"{LIB_PATH} has ABI version {got}, include/irr_hip.h / irr_amd.hip expect {ABI_VERSION}: the "
                            "library was built from other sources -- rebuild it with `python -m irr_amd.build --force`")
                    for name, (ret, args) in self.protos.items():
                        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
                        fn.restype = ctypes.c_long if ret == "long" else ctypes.c_int
                        fn.argtypes = [_CTYPES[t] for t, _ in args]
                    self._lib = lib
        return self._lib


_LIB = _Lib()


def lib() -> ctypes.CDLL:
    return _LIB.load()


def prototypes():
    return _LIB.protos


class HipError(RuntimeError):
    pass


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point; non-zero return -> HipError (mirrors the reference's
    AT_ERROR("CUDA call failed") path, models/correlation_package/correlation_cuda.cc:78-80)."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise HipError(f"{name} failed with code {rc}")


def ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    return t.data_ptr()


def stream() -> int:
    """hipStream_t of the CURRENT device's current stream.  Kernels are launched on the calling thread's current device, so
    every entry into the library happens under ``device_of`` / ``Function`` below (the reference's counterpart is
    ``with torch.cuda.device_of(input1)``, models/correlation_package/correlation.py:21,34)."""
    return torch.cuda.current_stream().cuda_stream


class device_of:
    """``with hip.device_of(t):`` -- makes t's device current for the launches inside (no-op when it already is, or when t
    is not a HIP tensor: the operators then raise their own 'no CPU fallback' error)."""

    __slots__ = ("idx", "prev")

    def __init__(self, t):
        self.idx = t.device.index if (isinstance(t, torch.Tensor) and t.is_cuda) else -1
        self.prev = -1

    def __enter__(self):
        if self.idx >= 0 and self.idx != torch.cuda.current_device():
            self.prev = torch.cuda.current_device()
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


class Function(torch.autograd.Function):
    """autograd.Function whose forward runs with the device of its first HIP tensor argument current (the autograd engine
    already does the same for backward: its worker threads are per device)."""

    @classmethod
    def apply(cls, *args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with device_of(a):
                        return super().apply(*args, **kwargs)
                break
        return super().apply(*args, **kwargs)


def check_plane_dense(t: torch.Tensor) -> int:
    """Validates the NCHW-with-dense-planes layout of include/irr_hip.h and returns the batch stride."""
    assert t.dim() == 4 and t.dtype == torch.float32 and t.is_cuda, (t.shape, t.dtype, t.device)
    b, c, h, w = t.shape
    sb, sc, sh, sw = t.stride()
    if b == 1:
        sb = c * h * w if c * h * w > 0 else 0
    assert (sw == 1 or w == 1) and (sh == w or h == 1) and (sc == h * w or c == 1), \
        f"tensor must have dense H*W planes, got shape {tuple(t.shape)} stride {t.stride()}"
    return sb


def bs(t: torch.Tensor) -> int:
    return check_plane_dense(t)
