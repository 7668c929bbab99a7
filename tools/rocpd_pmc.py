"""Per-kernel PMC summary from a rocprofv3 rocpd database (--pmc run).  usage: rocpd_pmc.py db [name-substring]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else "conv_"
rows = db.execute("select name, dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events "
                  "where name like ? group by name, dispatch_id, counter_name", (f"%{sub}%",)).fetchall()
agg = {}
for name, disp, cn, val, dur in rows:
    short = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    short = re.sub(r"\(.*\)$", "", short)[:60]
    a = agg.setdefault(short, {"n": set(), "dur": {}, "c": {}})
    a["n"].add(disp)
    a["dur"][disp] = dur
    a["c"][cn] = a["c"].get(cn, 0.0) + val
for k, a in agg.items():
    n = len(a["n"])
    print(f"{k}  dispatches={n} avg_us={sum(a['dur'].values()) / n / 1e3:.1f}")
    for cn, v in sorted(a["c"].items()):
        print(f"    {cn:28s} {v / n:16.0f} per dispatch")
