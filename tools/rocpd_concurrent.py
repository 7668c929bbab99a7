"""Kernels of OTHER streams that run while a named kernel runs (rocprofv3 --kernel-trace rocpd database).
usage: python tools/rocpd_concurrent.py <results.db> <substring of the kernel name>"""
import sqlite3
import sys
from collections import Counter

from rocpd_stats import short

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
hits = [(n, s, e, st) for n, s, e, st in rows if sys.argv[2] in n]
print(f"{len(hits)} launches match")
tot = Counter()
for n, s, e, st in hits[-8:]:
    over = [(short(n2)[:70], max(s, s2), min(e, e2)) for n2, s2, e2, st2 in rows if st2 != st and s2 < e and e2 > s]
    print(f"  {short(n)[:60]} ({(e - s) / 1e3:.0f} us) runs beside: " + "; ".join(f"{k} [{(b - a) / 1e3:.0f} us]" for k, a, b in over))
    for k, a, b in over:
        tot[k] += 1
print("most frequent neighbours:", tot.most_common(8))
