"""Kernel sequence of the LAST pass in a rocprofv3 --kernel-trace rocpd database whose passes are separated by a spin kernel
(tools/fwd_levels_trace.py): start offset, duration, gap to the end of the previous kernel, blocks.
usage: python tools/rocpd_seq.py <results.db> [from_index]"""
import sqlite3
import sys

from rocpd_stats import short


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]

    def pick(*subs):
        for c in cols:
            if all(s in c.lower() for s in subs):
                return c
        return None
    c_start, c_end = pick("start"), pick("end")
    gx, gy, gz = pick("grid", "x"), pick("grid", "y"), pick("grid", "z")
    wx, wy, wz = pick("workgroup", "x"), pick("workgroup", "y"), pick("workgroup", "z")
    rows = db.execute(f"select name, {c_start}, {c_end}, {gx}, {gy}, {gz}, {wx}, {wy}, {wz} from kernels order by {c_start}").fetchall()
    marks = [i for i, r in enumerate(rows) if "spin" in r[0].lower() or "sleep" in r[0].lower()]
    rows = rows[marks[-2] + 1:marks[-1]]
    t0 = rows[0][1]
    prev_end = t0
    busy = gap_sum = 0
    print(f"{'#':>4s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'blocks':>7s} {'thr':>4s}  kernel")
    for i, (name, s, e, a, b, c, x, y, z) in enumerate(rows):
        blocks = (a // max(x, 1)) * (b // max(y, 1)) * (c // max(z, 1))
        gap = (s - prev_end) / 1e3
        print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {blocks:7d} {x * y * z:4d}  {short(name)[:90]}")
        busy += e - s
        gap_sum += max(0, s - prev_end)
        prev_end = max(prev_end, e)
    print(f"# {len(rows)} kernels, span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, gaps {gap_sum / 1e3:.1f} us")


if __name__ == "__main__":
    main()
