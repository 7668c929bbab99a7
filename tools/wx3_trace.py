"""s_memtime trace of conv_wgrad_x3_kernel (tagged build with IRR_WX3_TRACE=1): per-unit cycle counts of wave 0 and wave 4 of
one block on the 565 -> 128 layer at 96x112x64: compute (1->2), publish (2->3), issue (3->4), barrier wait (4->5), loop top (5->1)."""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip
shapes = {"565->128": (565, 128), "467->64": (467, 64), "32->32": (32, 32)}
name = sys.argv[1] if len(sys.argv) > 1 else "565->128"
cin, cout = shapes[name]
B, H, W = (64, 96, 112) if cin > 32 else (64, 384, 448)
x = torch.randn(B, cin, H, W, device="cuda"); gy = torch.randn(B, cout, H, W, device="cuda")
gw = torch.zeros(cout, cin, 3, 3, device="cuda")
for _ in range(2):
    C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, 1, gw=gw)
torch.cuda.synchronize()
buf = np.zeros(16 * 400, np.uint64)
lib = ctypes.CDLL(hip.LIB_PATH)
lib.irr_wx3_trace_dump.argtypes = [ctypes.c_void_p]
print(name, "rc", lib.irr_wx3_trace_dump(buf.ctypes.data))
buf = buf.reshape(16, 400)
for wave in range(16):
    ev = [(int(v >> 56), int(v & ((1 << 56) - 1))) for v in buf[wave] if v]
    if not ev:
        continue
    d12 = [t1 - t0 for (s0, t0), (s1, t1) in zip(ev[20:-1], ev[21:]) if (s0, s1) == (1, 2)]
    d45 = [t1 - t0 for (s0, t0), (s1, t1) in zip(ev[20:-1], ev[21:]) if (s0, s1) == (4, 5)]
    print(f"wave {wave:2d}: compute {np.mean(d12):7.0f}  barrier wait {np.mean(d45):7.0f}")
for wave in (0, 4):
    ev = [(int(v >> 56), int(v & ((1 << 56) - 1))) for v in buf[wave] if v]
    print(f"wave {wave}: {len(ev)} events")
    deltas = {}
    for (s0, t0), (s1, t1) in zip(ev[20:-1], ev[21:]):
        deltas.setdefault((s0, s1), []).append(t1 - t0)
    for k, v in sorted(deltas.items()):
        print(f"   {k[0]} -> {k[1]}: mean {np.mean(v):8.0f}  median {np.median(v):8.0f}  max {np.max(v):8.0f} cycles (n={len(v)})")
