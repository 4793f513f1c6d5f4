#!/usr/bin/env python3
"""Idle time of the GPU inside a traced run (rocprofv3 --kernel-trace rocpd database): the union of all kernel intervals against the wall span,
the gaps (no kernel running on any queue) by size class, and the largest gaps with the kernels around them.
  python scripts/rocpd_gaps.py trace.db [skip_first_n_kernels] [min_gap_us]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
if skip < 0:      # -N: the window of N whole training steps that ends with the LAST-BUT-ONE pair of adam_kernel launches (an update ends with its Adam launch)
    adam = [e for (n, s_, e) in rows if "adam_kernel" in n]
    nsteps = -skip
    hi = adam[-3]                      # end of a gen_update two updates before the last traced one (keeps clear of the post-run probes)
    lo = adam[-3 - 2 * nsteps]
    rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
    print("window: %d steps between adam launches, %.3f ms per step" % (nsteps, (hi - lo) / 1e6 / nsteps))
else:
    rows = rows[skip:]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("aclgan::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:60]


t0, t1 = rows[0][1], max(r[2] for r in rows)
busy = 0
cur_end = rows[0][1]
last_name = None
gaps = []
for name, s, e in rows:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, last_name, short(name)))
        busy += 0
        cur_end = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        last_name = short(name)
span = t1 - t0
print("kernels %d, wall span %.3f ms, GPU busy (union of kernel intervals) %.3f ms = %.1f %%, idle %.3f ms" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 1e9)):
    g = [x for x in gaps if lo * 1e3 <= x[0] < hi * 1e3]
    print("  gaps %5s .. %5s us: %6d, %.3f ms" % (lo, hi if hi < 1e9 else "inf", len(g), sum(x[0] for x in g) / 1e6))
print("largest gaps (us, at ms, after kernel -> before kernel):")
for g in sorted(gaps, reverse=True)[:25]:
    if g[0] / 1e3 >= min_gap:
        print("  %8.1f  at %9.3f  %s -> %s" % (g[0] / 1e3, g[1] / 1e6, g[2], g[3]))
