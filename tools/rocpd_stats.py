"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace --stats) as a kernel table.
usage: python tools/rocpd_stats.py <results.db> [top_n]  > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# kernels: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.3f} ms total GPU kernel time")
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, calls, tot, avg, mn, mx in rows[:top]:
        print(f"{short(name):110s} {calls:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f}")
    aten = [r for r in rows if "at::native" in r[0] or "rocclr" in r[0]]
    print(f"# torch / runtime kernels (at::native::*, __amd_rocclr_*): {sum(r[1] for r in aten)} dispatches, "
          f"{sum(r[2] for r in aten) / 1e6:.3f} ms; library kernels: {sum(r[1] for r in rows) - sum(r[1] for r in aten)} dispatches")
    rest = rows[top:]
    if rest:
        print(f"{'(other ' + str(len(rest)) + ' kernels)':110s} {sum(r[1] for r in rest):7d} {sum(r[2] for r in rest) / 1e6:10.3f}")


if __name__ == "__main__":
    main()
