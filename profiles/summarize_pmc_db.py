"""Per-kernel average of one PMC counter from a rocprofv3 --pmc result database.
usage: python summarize_pmc_db.py <results.db> <COUNTER>"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? "
                   "group by kernel_name, dispatch_id", (sys.argv[2],)).fetchall()
agg = {}
for name, _, v in rows:
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("sdpb::", "")
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += v
print(f"# kernel, launches, avg {sys.argv[2]} per launch [KB]")
for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1] / kv[1][0])[:70]:
    print(f"{name[:80]:80s} {n:6d} {tot / n:16.1f}")
