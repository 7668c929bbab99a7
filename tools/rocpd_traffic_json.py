"""hbm_traffic.json from the two PMC passes (same aggregation as rocpd_traffic.py): {kernel: {launches, avg_us,
fetch_MB_per_launch_corrected, write_MB_per_launch}}.  usage: rocpd_traffic_json.py fetch.db write.db > profiles/hbm_traffic.json
(the json also records the hash of the library sources the passes ran on: irr_amd.build.source_hash())"""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import build as _build  # noqa: E402


def agg(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select name, dispatch_id, sum(counter_value), max(duration) from pmc_events "
                      "where counter_name=? group by name, dispatch_id", (counter,)).fetchall()
    out = {}
    for name, _d, v, dur in rows:
        short = re.sub(r"\(anonymous namespace\)::|void ", "", name)
        short = re.sub(r"\(.*\)$", "", short)
        short = re.sub(r", (false|true)>", lambda m: ">" if m.group(1) == "false" else ", true>", short)[:60]
        a = out.setdefault(short, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += v
        a[2] += dur
    return out


f = agg(sys.argv[1], "FETCH_SIZE")
w = agg(sys.argv[2], "WRITE_SIZE")
res = {"_source_hash": _build.source_hash()}      # bench.py reports these bytes only for a library built from the same sources
for k in sorted(f, key=lambda k: -f[k][2])[:40]:
    n = f[k][0]
    wn = max(w.get(k, [1])[0], 1)
    res[k] = {"launches": n, "avg_us": round(f[k][2] / n / 1e3, 1),
              "fetch_MB_per_launch_corrected": round(f[k][1] / n * 1024 * 2 / 1e6, 2),
              "write_MB_per_launch": round(w.get(k, [1, 0.0, 0.0])[1] / wn * 1024 / 1e6, 2)}
print(json.dumps(res, indent=1))
