"""print per-kernel PMC counter averages from a rocprofv3 rocpd database."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "conv"
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select * from counters_collection").fetchall()
ci = {c: i for i, c in enumerate(cols)}
agg = {}
for r in rows:
    name = r[ci.get("kernel_name", ci.get("name", 0))] if ("kernel_name" in ci or "name" in ci) else "?"
    if pat not in str(name): continue
    cn = r[ci["counter_name"]]; v = r[ci["value"]]
    short = str(name).replace("(anonymous namespace)::", "").replace("aclgan::", "").replace("void ", "")
    a = agg.setdefault((re.sub(r"\(.*$", "", short)[:60], cn), [0, 0.0]); a[0] += 1; a[1] += v
for (k, cn), (n, v) in sorted(agg.items()):
    print("%-62s %-28s n=%d avg=%.4g" % (k, cn, n, v / n))
if not agg: print("columns:", cols)
