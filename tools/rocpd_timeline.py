"""Occupancy timeline of the LAST train step in a rocprofv3 --kernel-trace rocpd database: how much of the step has
no kernel running, exactly one running with a grid that cannot fill 256 CUs, or two lanes running.
usage: python tools/rocpd_timeline.py <results.db> [--dump FROM_MS TO_MS]"""
import sqlite3
import sys

from rocpd_stats import short


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    if "--cols" in sys.argv:
        print(cols)
    def pick(*subs):
        for c in cols:
            if all(s in c.lower() for s in subs):
                return c
        return None
    c_start, c_end = pick("start"), pick("end")
    gx, gy, gz = pick("grid", "x"), pick("grid", "y"), pick("grid", "z")
    wx, wy, wz = pick("workgroup", "x"), pick("workgroup", "y"), pick("workgroup", "z")
    c_stream = pick("stream") or pick("queue")
    q = f"select name, {c_start}, {c_end}, {gx}, {gy}, {gz}, {wx}, {wy}, {wz}, {c_stream} from kernels order by {c_start}"
    rows = db.execute(q).fetchall()
    # last step = from the last 'adam' kernel but one to the last one
    adam = [i for i, r in enumerate(rows) if "adam" in r[0].lower()]
    if len(adam) >= 2:
        rows = rows[adam[-2] + 1:adam[-1] + 1]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    ev = []
    for name, s, e, a, b, c, x, y, z, st in rows:
        blocks = (a // max(x, 1)) * (b // max(y, 1)) * (c // max(z, 1)) if a >= x else a * b * c
        ev.append((s, 1, blocks, name, st))
        ev.append((e, -1, blocks, name, st))
    ev.sort(key=lambda t: (t[0], t[1]))
    active = {}
    last = t0
    idle = single_small = single_big = multi = 0
    small_by = {}
    for t, d, blocks, name, st in ev:
        dt = t - last
        if dt > 0:
            n = len(active)
            if n == 0:
                idle += dt
            elif n == 1:
                (k, (bl, nm)), = active.items()
                if bl < 256:
                    single_small += dt
                    key = "%s [%d blocks]" % (short(nm)[:70], bl)
                    small_by[key] = small_by.get(key, 0) + dt
                else:
                    single_big += dt
            else:
                multi += dt
        last = t
        key = (name, st, blocks, d > 0)
        if d > 0:
            active[(name, st, t)] = (blocks, name)
        else:
            for k in list(active):
                if k[0] == name and k[1] == st:
                    del active[k]
                    break
    span = t1 - t0
    per = {}
    for name, s_, e_, a_, b_, c_, x_, y_, z_, st in rows:
        v = per.setdefault(st, [s_, e_, 0, 0])
        v[0] = min(v[0], s_); v[1] = max(v[1], e_); v[2] += e_ - s_; v[3] += 1
    for st, v in sorted(per.items(), key=lambda kv: -kv[1][2]):
        print(f"stream {st}: first start +{(v[0] - t0) / 1e6:.2f} ms, last end +{(v[1] - t0) / 1e6:.2f} ms, kernel time {v[2] / 1e6:.2f} ms, {v[3]} dispatches")
    cnt = {}
    for r in rows:
        k = short(r[0])[:90]
        c2 = cnt.setdefault(k, [0, 0]); c2[0] += 1; c2[1] += r[2] - r[1]
    tk = [v for k, v in cnt.items() if "at::native" in k or "rocclr" in k]
    print(f"torch / runtime kernels of the step (at::native::*, __amd_rocclr_*): {sum(v[0] for v in tk)} dispatches, "
          f"{sum(v[1] for v in tk) / 1e6:.2f} ms of kernel time; library kernels: {len(rows) - sum(v[0] for v in tk)} dispatches")
    print("dispatches of the step by kernel (count, total ms):")
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"   {v[0]:5d}  {v[1] / 1e6:7.2f} ms  {k}")
    # tail: what runs in the last 12 ms of the step
    tail = [(r[1], r[2], r[0], r[9]) for r in rows if r[2] > t1 - 12e6]
    agg = {}
    for s_, e_, nm, st in tail:
        k = (st, short(nm)[:60])
        a2 = agg.setdefault(k, [0, 0]); a2[0] += 1; a2[1] += e_ - max(s_, t1 - 12e6)
    print("last 12 ms of the step:")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   stream {k[0]}  {v[1] / 1e6:6.2f} ms  {v[0]:4d}x  {k[1]}")
    print(f"step span {span / 1e6:.2f} ms: idle {idle / 1e6:.2f} ms, one kernel < 256 blocks {single_small / 1e6:.2f} ms, "
          f"one kernel >= 256 blocks {single_big / 1e6:.2f} ms, >= 2 kernels {multi / 1e6:.2f} ms  ({len(rows)} dispatches)")
    for k, v in sorted(small_by.items(), key=lambda kv: -kv[1])[:45]:
        print(f"   {v / 1e6:7.2f} ms  {k}")
    if "--dump" in sys.argv:         # --dump FROM_MS TO_MS (relative to the step's end when negative): every dispatch in the window
        i = sys.argv.index("--dump")
        lo, hi = float(sys.argv[i + 1]) * 1e6, float(sys.argv[i + 2]) * 1e6
        lo = t1 + lo if lo < 0 else t0 + lo
        hi = t1 + hi if hi <= 0 else t0 + hi
        print("dispatches in the window (start offset us, duration us, gap to the previous end on the same stream us, stream, blocks, kernel):")
        prev_end = {}
        for name, s_, e_, a_, b_, c_, x_, y_, z_, st in rows:
            blocks = (a_ // max(x_, 1)) * (b_ // max(y_, 1)) * (c_ // max(z_, 1)) if a_ >= x_ else a_ * b_ * c_
            gap = (s_ - prev_end[st]) / 1e3 if st in prev_end else 0.0
            prev_end[st] = e_
            if s_ >= lo and s_ <= hi:
                print(f"  {(s_ - t0) / 1e3:10.1f} {(e_ - s_) / 1e3:8.1f} {gap:8.1f}  s{st} {blocks:6d}  {short(name)[:80]}")


if __name__ == "__main__":
    main()
