"""Same-queue overlaps in a rocprofv3 --kernel-trace rocpd database: consecutive kernels of ONE queue / stream whose execution
intervals intersect.  usage: python tools/rocpd_overlap.py <results.db>"""
import sqlite3
import sys

from rocpd_stats import short

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)


def pick(*subs):
    for c in cols:
        if all(s in c.lower() for s in subs):
            return c
    return None


c_start, c_end = pick("start"), pick("end")
c_q = pick("queue") or pick("stream")
c_s = pick("stream") or c_q
rows = db.execute(f"select name, {c_start}, {c_end}, {c_q}, {c_s} from kernels order by {c_start}").fetchall()
byq = {}
for name, s, e, q, st in rows:
    byq.setdefault((q, st), []).append((s, e, name))
for key, ks in byq.items():
    ks.sort()
    n_ov, worst = 0, []
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        if s1 < e0:
            n_ov += 1
            worst.append(((e0 - s1) / 1e3, short(n0)[:60], short(n1)[:60], (e0 - s0) / 1e3))
    worst.sort(reverse=True)
    print(f"queue/stream {key}: {len(ks)} kernels, {n_ov} consecutive pairs overlap")
    for ov, a, b, d0 in worst[:12]:
        print(f"    {b}  started {ov:8.1f} us before the end of  {a} ({d0:.1f} us long)")
