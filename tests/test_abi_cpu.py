"""CPU-only: the C-ABI library loads and exports every symbol include/irr_hip.h declares."""
import ctypes
import os

from irr_amd import hip


def test_header_parses():
    protos = hip.prototypes()
    assert "irr_corr81_fwd_f32" in protos and "irr_warp_bwd_f32" in protos
    for name, (ret, args) in protos.items():
        assert ret in ("int", "long")
        for ty, _ in args:
            assert ty in hip._CTYPES, (name, ty)     # plain pointers / sizes only, no torch types


def test_library_exports_all_declared_symbols():
    assert os.path.exists(hip.LIB_PATH), "run `python -m irr_amd.build` (or __graft_entry__.build()) first"
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in hip.prototypes():
        assert hasattr(lib, name), f"{name} declared in include/irr_hip.h but not exported"
    assert lib.irr_abi_version() >= 1


def test_argument_validation_without_gpu():
    # launchers reject null / non-positive arguments before touching the device
    hip.lib()
    try:
        hip.call("irr_corr81_fwd_f32", None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, None)
    except hip.HipError as e:
        assert "-22" in str(e)
    else:
        raise AssertionError("expected IRR_EINVAL")
