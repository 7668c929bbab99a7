"""s_memtime trace of conv_x3s_kernel (build with IRR_X3S_TRACE=1): per-phase cycle counts of one MFMA wave and one
producer wave of block 7."""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip
B, cin, cout, H, W = 64, 32, 32, 384, 448
x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
for _ in range(2):
    C.conv_forward(x, w, b, 1, 1, True)
torch.cuda.synchronize()
buf = np.zeros(8 * 400, np.uint64)
lib = ctypes.CDLL(hip.LIB_PATH)
lib.irr_x3s_trace_dump.argtypes = [ctypes.c_void_p]
rc = lib.irr_x3s_trace_dump(buf.ctypes.data)
print("rc", rc)
buf = buf.reshape(8, 400)
for wave, name in ((0, "MFMA wave 0"), (4, "producer wave 4")):
    ev = [(int(v >> 56), int(v & ((1 << 56) - 1))) for v in buf[wave] if v]
    print(name, len(ev), "events")
    # per-slot deltas averaged over tiles 3.. (skip warm-up)
    deltas = {}
    for (s0, t0), (s1, t1) in zip(ev[30:-1], ev[31:]):
        deltas.setdefault((s0, s1), []).append(t1 - t0)
    for k, v in sorted(deltas.items()):
        print(f"   {k[0]:2d} -> {k[1]:2d}: mean {np.mean(v):8.0f} cycles  (n={len(v)})")
