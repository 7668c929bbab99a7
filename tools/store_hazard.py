"""Round 5, NOTES D.5: the wide-store hazard of gfx950 stand-alone (tools/store_hazard.hip).  Every thread stores (1, 1, 1, 1) with
`buffer_store_dwordx4 v[10:13], ...` and overwrites v10 with 2.0 after WS wait states; a slot whose first component reads 2.0 was stored
from the rewritten register.  FORM 0: SGPR soffset (hipcc inserts no wait states behind it), FORM 1: literal soffset 0 (hipcc: two).
Run alone, beside a streaming copy kernel on a second stream, and beside the library's 32 -> 32 weight gradient (the kernel that ran beside
conv_x3s_kernel when the fault showed inside one process).   python3 tools/store_hazard.py [launches per setting]"""
import ctypes, os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C

_here = os.path.dirname(os.path.abspath(__file__))
_so = os.path.join(_here, "_store_hazard.so")
if not os.path.exists(_so) or os.path.getmtime(_so) < os.path.getmtime(os.path.join(_here, "store_hazard.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(_here, "store_hazard.hip"), "-o", _so], check=True)
so = ctypes.CDLL(_so)
so.launch_store_victim.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
so.launch_mem_aggressor.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
NBLK, ITERS = 2048, 96
nthr = NBLK * 256
out = torch.empty(ITERS * nthr * 4, device="cuda")
side = torch.cuda.Stream()
big_a = torch.randn(64 * 1024 * 1024, device="cuda")            # 256 MB each
big_b = torch.empty_like(big_a)
C.set_math("h2")
B, H, W = 16, 384, 448
x = torch.randn(B, 32, H, W, device="cuda")
gy = torch.randn(B, 32, H, W, device="cuda") * 1e-4
gw = torch.zeros(32, 32, 3, 3, device="cuda")
xa, ga = C.amax_measure(x), C.amax_measure(gy)


def aggress(kind):
    if kind == "alone":
        return
    side.wait_stream(torch.cuda.current_stream())
    if kind == "copy kernel":
        assert so.launch_mem_aggressor(big_a.data_ptr(), big_b.data_ptr(), big_a.numel() // 4, 3, 1024, side.cuda_stream) == 0
    else:
        with torch.cuda.stream(side):
            for _ in range(3):
                C.conv_wgrad(x, gy, gw.shape, 1, 1, gw=gw, x_amax=xa, gy_amax=ga)


print(f"{NBLK} blocks x 256 threads x {ITERS} stores of 16 B per launch, {N} launches per setting; wrong = slots whose component 0 reads 2.0")
for form, fname in ((0, "SGPR soffset"), (1, "literal soffset 0")):
    for ws in (0, 1, 2, 4, 8):
        row = []
        for kind in ("alone", "copy kernel", "32->32 weight gradient"):
            wrong = launches_hit = other = 0
            for rep in range(N):
                out.zero_()
                aggress(kind)
                if rep % 3:
                    torch.cuda._sleep(100000 * (rep % 3))
                assert so.launch_store_victim(out.data_ptr(), NBLK, ITERS, nthr * 16, form, ws, torch.cuda.current_stream().cuda_stream) == 0
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                v = out.view(-1, 4)
                w0 = int((v[:, 0] == 2.0).sum())
                oth = int((v[:, 1:] != 1.0).sum()) + int(((v[:, 0] != 1.0) & (v[:, 0] != 2.0)).sum())
                wrong += w0; other += oth; launches_hit += int(w0 > 0)
            row.append(f"{kind}: {wrong} wrong slots in {launches_hit} of {N} launches" + (f" (+{other} other mismatches)" if other else ""))
        print(f"{fname:18s} {ws} wait states | " + " | ".join(row), flush=True)
# the narrower stores with an SGPR soffset and NO wait state (dword / dwordx2: outside the hazard rule; dwordx3: inside it)
so.launch_store_victim_narrow.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
for width in (1, 2, 3):
    row = []
    for kind in ("alone", "copy kernel", "32->32 weight gradient"):
        wrong = 0
        for rep in range(N):
            out.zero_()
            aggress(kind)
            assert so.launch_store_victim_narrow(out.data_ptr(), NBLK, ITERS, nthr * 16, width, torch.cuda.current_stream().cuda_stream) == 0
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            wrong += int((out.view(-1, 4)[:, 0] == 2.0).sum())
        row.append(f"{kind}: {wrong} wrong slots")
    print(f"buffer_store_dword{'' if width == 1 else 'x' + str(width)}, SGPR soffset, 0 wait states | " + " | ".join(row), flush=True)
# LDS: ds_write_b128 followed at once by a VALU write of its first data register
so.launch_lds_victim.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
row = []
for kind in ("alone", "copy kernel", "32->32 weight gradient"):
    wrong = other = 0
    for rep in range(N):
        out.zero_()
        aggress(kind)
        assert so.launch_lds_victim(out.data_ptr(), NBLK, 2000, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        v = out[:nthr * 4].view(-1, 4)
        wrong += int(v[:, 0].sum()); other += int(v[:, 1].sum())
    row.append(f"{kind}: {wrong} wrong slots" + (f" (+{other} other)" if other else ""))
print("ds_write_b128, overwrite of its first data register in the next cycle | " + " | ".join(row), flush=True)
# where in a wave do the wrong slots sit (last setting that showed any: form 0, no wait states, beside the weight gradient)
out.zero_()
aggress("32->32 weight gradient")
so.launch_store_victim(out.data_ptr(), NBLK, ITERS, nthr * 16, 0, 0, torch.cuda.current_stream().cuda_stream)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
idx = (out.view(-1, 4)[:, 0] == 2.0).nonzero().flatten()
if idx.numel():
    lanes = (idx % 64).tolist()
    hist = [0] * 64
    for l in lanes:
        hist[l] += 1
    print("lane histogram of the wrong slots (form 0, no wait states, beside the weight gradient):", hist)
