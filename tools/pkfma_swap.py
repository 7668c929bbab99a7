"""Round 4, NOTES C.3: the stand-alone victim of tools/pkfma_swap.hip (a loop of four packed FMAs, one of them in the form under
test) on the main stream, beside -- on a second stream -- the library's dilation-16 weight gradient, a synthetic kernel that only
streams MFMAs, or a synthetic kernel of scalar FMAs in the same block shape.  Every launch is bit-compared with the lone launch.
FORMS=0,2,5 selects forms.  (Builds tools/_pkfma_swap.so with hipcc when it is missing.)"""
import ctypes, os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C

_here = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(os.path.join(_here, "_pkfma_swap.so")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(_here, "pkfma_swap.hip"),
                    "-o", os.path.join(_here, "_pkfma_swap.so")], check=True)
so = ctypes.CDLL(os.path.join(_here, "_pkfma_swap.so"))
so.launch_swap_victim.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
torch.manual_seed(0)
B, H, W = 8, 96, 112
NBLK, ITERS = 8192, 600
side = torch.cuda.Stream()
x16 = torch.randn(B, 96, H, W, device="cuda")
g16 = torch.randn(B, 64, H, W, device="cuda") * 1e-6
gw16 = torch.zeros(64, 96, 3, 3, device="cuda")
vin = torch.rand(NBLK * 256 * 4, device="cuda") * 0.8 + 0.1
vw = torch.rand(32 * 12, device="cuda") * 0.2 + 0.05
C.set_math("h2")


def victim(form):
    out = torch.empty_like(vin)
    rc = so.launch_swap_victim(vin.data_ptr(), vw.data_ptr(), out.data_ptr(), NBLK, ITERS, form, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out


so.launch_mfma_aggressor.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
so.launch_valu_aggressor.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.zeros(4096, device="cuda")
FORMS = {0: "accumulator halves swapped, between s_load_dword and s_waitcnt lgkmcnt(0) (as in the kernel)", 1: "no swap",
         2: "swap, no scalar load / wait", 3: "swap, s_nop 7 before the wait", 4: "swap, scalar load issued four instructions earlier",
         5: "halves of the vector multiplicand swapped instead", 6: "swap, wait with nothing outstanding",
         7: "swap, one independent VALU instruction before the wait", 8: "swap is the second to last VALU instruction",
         9: "swap, s_nop 0 before the wait", 10: "v_pk_add_f32, addend halves swapped", 11: "v_pk_mul_f32, second factor swapped",
         12: "accumulator: high half to both results", 13: "swap, all three sources in VGPRs", 14: "accumulator: low half to both results",
         15: "swap, 1 wait state before it", 16: "swap, 2 wait states before it", 17: "swap, 4 wait states before it",
         18: "swap, 8 wait states before it", 19: "swap, 4 wait states right behind the pair's writer (one instruction earlier)",
         20: "swap, the pair's writer DIRECTLY before it", 21: "swap, writer directly before it + 4 wait states",
         22: "swap, 16 wait states before it", 23: "swap, four independent VALU instructions before it"}
KINDS = [k.strip() for k in os.environ.get("KINDS", "alone,d16 weight gradient,synthetic MFMA kernel,synthetic VALU kernel").split(",")]


def aggress(kind):
    if kind == "d16 weight gradient":
        xa, ga = C.amax_measure(x16), C.amax_measure(g16)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                C.conv_wgrad(x16, g16, gw16.shape, 1, 16, gw=gw16, x_amax=xa, gy_amax=ga)
    elif kind == "synthetic MFMA kernel":
        side.wait_stream(torch.cuda.current_stream())
        for _ in range(3):
            assert so.launch_mfma_aggressor(sink.data_ptr(), 256, 4000, side.cuda_stream) == 0
    elif kind == "synthetic VALU kernel":
        side.wait_stream(torch.cuda.current_stream())
        for _ in range(3):
            assert so.launch_valu_aggressor(sink.data_ptr(), 256, 4000, side.cuda_stream) == 0


for form in [int(f) for f in os.environ.get("FORMS", "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23").split(",")]:
    ref = victim(form)
    torch.cuda.synchronize()
    for kind in KINDS:
        bad, lanes, comps = 0, set(), set()
        for rep in range(30):
            aggress(kind)
            if kind != "alone" and rep % 4:
                torch.cuda._sleep(20000 * (rep % 4))
            out = victim(form)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                ne = (out != ref).nonzero().flatten()
                lanes |= set(((ne // 4) % 64).tolist())
                comps |= set((ne % 4).tolist())
        print(f"form {form} ({FORMS[form]}), {kind}: {bad} of 30 launches differ"
              + (f"; lanes {min(lanes)}..{max(lanes)} ({len(lanes)}); components {sorted(comps)}" if bad else ""), flush=True)
