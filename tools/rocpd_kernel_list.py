"""Every dispatch of the kernels whose name contains <substring> in a rocprofv3 --kernel-trace rocpd database: grid, duration.
usage: python tools/rocpd_kernel_list.py <results.db> <substring> [last_n]"""
import sqlite3
import sys

from rocpd_stats import short


def main():
    db = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
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
    rows = [r for r in rows if sub in r[0]]
    if last:
        rows = rows[-last:]
    for name, s, e, a, b, c, x, y, z in rows:
        print(f"{(e - s) / 1e3:9.1f} us  grid {a // max(x, 1):5d} x {b // max(y, 1):4d} x {c // max(z, 1):4d}  {short(name)[:60]}")


if __name__ == "__main__":
    main()
