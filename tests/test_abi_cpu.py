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
    assert lib.irr_abi_version() == hip.ABI_VERSION


def test_argument_validation_without_gpu():
    # launchers reject null / non-positive arguments before touching the device
    hip.lib()
    try:
        hip.call("irr_corr81_fwd_f32", None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, None)
    except hip.HipError as e:
        assert "-22" in str(e)
    else:
        raise AssertionError("expected IRR_EINVAL")


def test_binding_refuses_a_library_of_another_abi(monkeypatch):
    """the argument types come from the header in the tree, so a stale library must be refused at load time."""
    import pytest
    fresh = hip._Lib()
    monkeypatch.setattr(hip, "ABI_VERSION", hip.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        fresh.load()


def test_build_is_keyed_on_content_not_on_mtime(tmp_path):
    """objects newer than changed sources (the copy to the GPU box) must still be rebuilt; the stamp next to the
    library names the sources it was LINKED from."""
    from irr_amd import build as B
    assert not B.stale(), "conftest builds the library"
    assert B.built_hash() == B.source_hash()
    objdir = os.path.join(os.path.dirname(hip.LIB_PATH), "obj")
    stamp = os.path.join(objdir, "misc.o.hash")
    good = open(stamp).read()
    try:
        with open(stamp, "w") as f:                 # as if misc.hip had other contents when misc.o was compiled
            f.write("0" * 16)
        os.utime(os.path.join(objdir, "misc.o"))    # ... and the object is NEWER than every source
        t0 = os.path.getmtime(hip.LIB_PATH)
        B.build(verbose=False)
        assert open(stamp).read() == good, "the object was not recompiled although its content key differed"
        assert os.path.getmtime(hip.LIB_PATH) > t0, "the library was not relinked"
    finally:
        if open(stamp).read() != good:
            B.build(force=True, verbose=False)
    assert B.built_hash() == B.source_hash()


def test_streaming_kernel_h2_form_is_a_host_side_switch():
    """Routing is host code: the streaming 32-channel problems report code 9001 to both eligibility functions of the library, and
    irr_amd.conv puts them on the fp16x2 form unless conv.X3S_H2 is switched off (default on since round 5: DESIGN.md 5.2 /
    profiles/NOTES.md C.5)."""
    from irr_amd import conv as C, hip
    import os
    assert C.X3S_H2 == bool(int(os.environ.get("IRR_X3S_H2", "1")))
    old_math = C.MATH
    C.set_math("h2")
    try:
        shape = (64, 32, 384, 448, 32, 3, 1, 1)               # B, Cin, H, W, Cout, k, stride, dil: a full-resolution 32 -> 32 layer
        assert hip.lib().irr_conv2d_x3_eligible(*shape) == 9001 and hip.lib().irr_conv2d_h2_eligible(*shape) == 9001
        assert C.x3_code(*shape) == 9001
        old = C.set_x3s_h2(False)
        try:
            assert C.h2_code(*shape) == 0
            C.set_x3s_h2(True)
            assert C.h2_code(*shape) == 9001
            big = (64, 128, 96, 112, 128, 3, 1, 1)            # a conv_x3_kernel problem is on the fp16x2 form either way
            assert C.h2_code(*big) == C.x3_code(*big) != 0
            C.set_x3s_h2(False)
            assert C.h2_code(*big) == C.x3_code(*big)
        finally:
            C.set_x3s_h2(old)
    finally:
        C.set_math(old_math)
