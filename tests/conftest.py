import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is git-ignored: build it on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from irr_amd import build, hip
    if build.stale(hip.LIB_PATH):          # missing, or any csrc/ file, header or the build recipe is newer than the library
        if os.environ.get("IRR_HIP_LIB"):
            raise RuntimeError("IRR_HIP_LIB points to a library that is older than its sources: " + hip.LIB_PATH)
        build.build(verbose=False)
    yield


@pytest.fixture(params=["default", "x3_all", "h2_all"])
def routing(request):
    """Kernel-family routing of the conv layers for the reference-fixture tests.  ``default``: what the library picks for
    the (small) test shapes -- mostly the fp32-MFMA family.  ``x3_all`` / ``h2_all``: irr_conv_x3_set_min_blocks(0), i.e. every
    layer the split-operand family ACCEPTS runs on it (conv_x3 / conv_x3s forward and data gradient incl. the combined DenseNet
    column packs, conv_wgrad_x3) in its bf16x3 form / in the fp16x2 form wherever that exists -- h2_all is the routing
    bench.py's BASELINE-size step gets by default."""
    from irr_amd import conv as C, hip
    C.x3_code(1, 64, 8, 8, 64, 3, 1, 1)                  # applies IRR_X3_MIN_BLOCKS once, if set
    C.LAUNCHES.clear()
    if request.param == "default":
        yield "default"
        return
    old = hip.lib().irr_conv_x3_set_min_blocks(0)
    C.set_math("x3" if request.param == "x3_all" else "h2")
    try:
        yield request.param
        fam = "x3" if request.param == "x3_all" else "h2"
        n = sum(v for k, v in C.LAUNCHES.items() if fam in k)
        assert n > 0, f"{request.param} routing launched no {fam} kernel: {dict(C.LAUNCHES)}"
    finally:
        hip.lib().irr_conv_x3_set_min_blocks(old)
        C.set_math(C.DEFAULT_MATH)
