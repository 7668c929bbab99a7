"""Round 4 (profiles/NOTES.md C.3): the built library holds no packed fp32 instruction of the form that returned wrong values beside
MFMA waves of another kernel -- a v_pk_{fma,mul,add}_f32 whose LOW result reads the HIGH half of its own destination pair (what the
SLP vectoriser makes of a lane swap; tools/pkfma_swap.py reproduces the deviation stand-alone).  The build flags keep the vectorisers
out (tests/test_pair_gpu.py); this looks at the machine code itself, so a hand-written packed operation of that form is caught too."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner():
    spec = importlib.util.spec_from_file_location("scan_pk_swap", os.path.join(ROOT, "tools", "scan_pk_swap.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_low_reads_high_recognises_the_forms_that_deviated():
    s = _scanner()
    assert s.low_reads_high("v_pk_fma_f32 v[30:31], s[12:13], v[88:89], v[30:31] op_sel:[0,0,1] op_sel_hi:[0,1,0]") == [2]
    assert s.low_reads_high("v_pk_add_f32 v[2:3], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]") == [1]
    assert s.low_reads_high("v_pk_fma_f32 v[2:3], s[4:5], v[4:5], v[2:3] op_sel:[0,0,1] op_sel_hi:[0,1,1]") == [2]
    # in place, a scalar broadcast, the high half of ANOTHER pair: none of these deviated
    assert s.low_reads_high("v_pk_fma_f32 v[2:3], s[4:5], v[4:5], v[2:3] op_sel_hi:[0,1,1]") == []
    assert s.low_reads_high("v_pk_fma_f32 v[48:49], s[48:49], v[116:117], v[48:49] op_sel:[1,0,0]") == []
    assert s.low_reads_high("v_pk_fma_f32 v[94:95], v[196:197], v[190:191], v[94:95] op_sel:[1,0,0]") == []
    assert s.low_reads_high("v_pk_fma_f32 v[2:3], s[4:5], v[4:5], v[2:3] op_sel:[0,1,0] op_sel_hi:[0,0,1]") == []


def test_built_library_has_no_packed_instruction_of_that_form():
    s = _scanner()
    lib = os.path.join(ROOT, "irr_amd", "lib", "libirr_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    if not os.path.exists(os.path.join(s.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    total, hits = s.scan(lib)
    assert not hits, hits[:8]
