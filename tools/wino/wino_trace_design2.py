"""s_memtime timeline of one block of conv_wino_kernel (a build with -DWINO_TRACE=1: bash tools/wino_abl_build.sh 0 -DWINO_TRACE=1, then
IRR_HIP_LIB=irr_amd/lib_wabl0/libirr_hip.so python tools/wino_trace.py).  Stamps per chunk: 0 loop top, 1 raw published, 2 barrier A
passed, 3 transform done (waves that run it first), 4 MFMA phase done, 7 transform done (the others), 5 barrier B passed; 6 = kernel end.  Prints the mean duration of each segment per wave
(shader cycles) over the chunks of the 565 -> 128 launch."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip
from tools.wino_check import pack, wino_forward

cin, cout = int(os.environ.get("CIN", 565)), int(os.environ.get("COUT", 128))
x = torch.randn(64, cin, 96, 112, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
C.set_math("h2")
xa = C.amax_measure(x); pk = pack(w)
for _ in range(3):
    wino_forward(x, pk, b, True, xa)
buf = (ctypes.c_ulonglong * (8 * 1024 + 8))()
fn = hip.lib().irr_wino_trace_read
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.addressof(buf)) == 0
a = np.frombuffer(buf, dtype=np.uint64)
names = {(0, 1): "barrier", (1, 2): "DMA issue + fragments of tile group 0", (2, 3): "wait U + 12 MFMAs + fragments of tile group 1",
         (3, 4): "12 MFMAs + U issue", (4, 0): "loop back"}
for wv in range(8):
    n = int(a[8 * 1024 + wv])
    st = a[wv * 1024: wv * 1024 + n]
    slot = (st >> np.uint64(56)).astype(int); t = (st & np.uint64((1 << 56) - 1)).astype(np.int64)
    seg = {}
    for i in range(n - 1):
        seg.setdefault((slot[i], slot[i + 1]), []).append(t[i + 1] - t[i])
    tot = t[-1] - t[0]
    print(f"wave {wv}: {n} stamps, {tot} cycles total; per chunk " +
          "; ".join(f"{names.get(k, k)} {np.mean(v):.0f}" for k, v in sorted(seg.items()) if len(v) > 3))
    ep = seg.get((4, 6))
    if ep:
        print(f"        epilogue {ep[0]} cycles")
