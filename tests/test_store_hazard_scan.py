"""Round 5 (profiles/NOTES.md D.5, profiles/r5_store_hazard_ab.txt): on gfx950 a 16-byte buffer store with an SGPR soffset whose data
registers are overwritten by the next VALU instruction stored the NEW contents in lanes 12-15 of every 16-lane row whenever another
kernel or process loaded the memory pipeline (hipcc inserts wait states behind every wide store EXCEPT that form) -- the fault behind
round 4's two-rank NaN and round 5's shared-GPU corruption of conv_x3s_kernel.  The kernel's stores now pin eight wait states
(irr_buffer_store_b128_guarded, csrc/common.h); this test looks at the machine code of the built library: no store of more than
64 bits anywhere in it has its data registers rewritten within the two wait states the compiler gives every other form."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    spec = importlib.util.spec_from_file_location("scan_store_hazard", os.path.join(ROOT, "tools", "scan_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_operand_parsing():
    s = _scanner()
    assert s.data_regs("buffer_store_dwordx4 v[8:11], v12, s[80:83], s68 offen") == {8, 9, 10, 11}
    assert s.data_regs("global_store_dwordx4 v[114:115], v[4:7], off") == {4, 5, 6, 7}
    assert s.dest_regs("v_add_f32_e32 v8, v140, v182") == {8}
    assert s.dest_regs("v_cmp_gt_u32_e64 s[8:9], v14, v12") == set()
    assert s.dest_regs("buffer_load_dwordx4 v[0:3], v4, s[8:11], 0 offen") == {0, 1, 2, 3}
    assert s.WIDE.match("buffer_store_dwordx4 v[8:11], v12, s[80:83], s68 offen") and not s.WIDE.match("buffer_store_dwordx2 v[8:9], v12, s[80:83], 0 offen")


def test_built_library_keeps_wait_states_behind_every_wide_store():
    s = _scanner()
    lib = os.path.join(ROOT, "irr_amd", "lib", "libirr_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    if not os.path.exists(os.path.join(s.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    total, hits = s.scan(lib, 2)
    assert total > 100                      # (the scan did see the library's stores)
    assert not hits, hits[:6]
