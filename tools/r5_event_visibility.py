"""Round 5, NOTES D.2: is a kernel on an IDLE second stream that waits for an event of the main stream guaranteed to see everything the
main stream's last kernel wrote -- also while another process keeps the GPU busy?  (The two-rank NaN needs the lane's weight-gradient
launches to start right at the event, on an idle lane: profiles/r5_nan_ab.txt, calls 6 and 8.)

Per iteration k: main stream: a large persistent-style write of the value k into `buf` (our own streaming conv kernel writing a
32-channel map, or a plain fill); event; second stream: wait, then count elements that differ from the expected result.  Any
mismatch = the consumer saw stale bytes.   python tools/r5_event_visibility.py [iters] [conv|fill]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mode = sys.argv[2] if len(sys.argv) > 2 else "conv"          # conv | fill | conv_x3 (bf16x3 streaming kernel) | conv64 (64 -> 64: conv_x3_kernel, 39 KiB LDS)
check = sys.argv[3] if len(sys.argv) > 3 else "event"        # event: consumer on a second stream behind an event | same: consumer on the SAME stream
dev = torch.device("cuda", 0)
main, side = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
B, H, W = 16, 224, 512
bad = torch.zeros((), device=dev, dtype=torch.int64)
checked = 0
samples = []
CH = 64 if mode == "conv64" else 32
if mode.startswith("conv"):
    from irr_amd import conv as C, hip
    C.set_math("x3" if mode == "conv_x3" else "h2"); C.set_x3s_h2(True)
    w = torch.zeros(CH, CH, 3, 3, device=dev)
    for c in range(CH):
        w[c, c, 1, 1] = 1.0                                  # identity conv: out == in (exactly, in either operand form)
    x = [torch.full((B, CH, H, W), float(k + 1), device=dev) for k in range(2)]
    xa = [C.amax_measure(t) for t in x]
    print("kernel code", C.x3_code(B, CH, H, W, CH, 3, 1, 1), "h2", bool(C.h2_code(B, CH, H, W, CH, 3, 1, 1)), flush=True)
for k in range(N):
    if mode.startswith("conv"):
        out = torch.empty(B, CH, H, W, device=dev)           # fresh memory each time, like a backward pass's gradient maps
        C.conv_forward(x[k & 1], w, None, 1, 1, False, out=out, x_amax=xa[k & 1] if mode != "conv_x3" else None)
        want = float((k & 1) + 1)
    else:
        out = torch.empty(B, 32, H, W, device=dev)
        out.fill_(float(k))
        want = float(k)
    if check == "same":
        bad += (out != want).sum()
        idx = (out != want).flatten().nonzero()[:6, 0]
        samples.append((k, want, out.flatten()[idx], idx))
        done = torch.cuda.Event(); done.record(main)
    else:
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            bad += (out != want).sum()
            idx = (out != want).flatten().nonzero()[:6, 0]
            samples.append((k, want, out.flatten()[idx], idx))
            done = torch.cuda.Event(); done.record(side)
        out.record_stream(side)
    checked += out.numel()
    if k % 8 == 7:
        done.synchronize()                                   # let the side stream go IDLE again before the next producers
torch.cuda.synchronize()
print(f"{mode} / {check}: {N} producer/consumer pairs, {checked} elements checked, {int(bad)} stale", flush=True)
shown = 0
for k, want, vals, idx in samples:
    if vals is not None and vals.numel() and shown < 6:
        pos = [(int(i) // (CH * H * W), (int(i) // (H * W)) % CH, (int(i) // W) % H, int(i) % W) for i in idx.tolist()]
        print(f"  iteration {k}: expected {want}, found {vals.tolist()} at (b, c, y, x) {pos}", flush=True)
        shown += 1
