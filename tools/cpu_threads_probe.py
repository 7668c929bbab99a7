import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import irr_pwc_oracle as O
B = 2
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    P = O.make_trainable(O.synthetic_params(0)); opt = O.make_adam(P)
    batch = O.synthetic_batch(B, 384, 448, 1234)
    O.train_step(P, opt, batch)
    t0 = time.perf_counter(); O.train_step(P, opt, batch); dt = time.perf_counter() - t0
    print(th, "threads:", round(dt, 2), "s/step", round(B / dt, 3), "pairs/s", flush=True)
