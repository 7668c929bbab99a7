"""Round 4: the kernel pair behind the lane deviation, and the build flags that keep it away (irr_amd/build.py, profiles/NOTES.md C.3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_smallco_data_gradient_is_bit_stable_beside_the_dilated_weight_gradient():
    """The launch pair that produced round 4's lane deviation: the Cout = 2 data gradient of conv_last (accumulate + mask over a
    563-channel buffer) on the main stream while the dilation-16 weight gradient -- four-wave blocks, the one weight-gradient
    shape that leaves register room for other waves on its SIMDs -- runs on a second stream.  With v_pk_fma_f32 in the data-gradient
    kernel 30-39 of 40 launches came out 4e-3 off; built without the vectorisers every launch is bit-equal to the lone one."""
    from irr_amd import conv as C
    torch.manual_seed(0)
    B, H, W = 8, 96, 112
    side = torch.cuda.Stream()
    g_est = torch.randn(B, 2, H, W, device="cuda") * 1e-5
    w_last = torch.randn(2, 563, 3, 3, device="cuda") * 0.02
    G0 = torch.randn(B, 565, H, W, device="cuda") * 1e-6
    buf = torch.randn(B, 565, H, W, device="cuda")
    x16 = torch.randn(B, 96, H, W, device="cuda")
    g16 = torch.randn(B, 64, H, W, device="cuda") * 1e-6
    gw16 = torch.zeros(64, 96, 3, 3, device="cuda")

    def victim():
        G = G0.clone()
        C.conv_dgrad(g_est, w_last, 1, 1, (H, W), gx=G[:, :563], accumulate=True, mask=buf[:, :563], nmask=32)
        return G

    ref = victim()
    torch.cuda.synchronize()
    xa, ga = (C.amax_measure(x16), C.amax_measure(g16)) if C.MATH == "h2" else (None, None)
    bad = 0
    for rep in range(24):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                C.conv_wgrad(x16, g16, gw16.shape, 1, 16, gw=gw16, x_amax=xa, gy_amax=ga)
        if rep % 4:
            torch.cuda._sleep(20000 * (rep % 4))
        out = victim()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    assert bad == 0, bad


def test_library_has_no_packed_fp32_math_outside_the_forward_cost_volume():
    """the build flags that keep the vectorisers out (only corr81_fwd4_kernel -- forward pass, nothing of another stream beside it --
    keeps explicit packed math)"""
    from irr_amd import build
    assert "-fno-slp-vectorize" in build.COMMON and "-fno-vectorize" in build.COMMON
