#!/usr/bin/env python3
"""Kernel time per roctx range (network pass) from a `rocprofv3 --kernel-trace --marker-trace` database of a run with ACLGAN_ROCTX=1
(csrc/engine.hip: one range per forward pass of a network, "gen_BA.decode#2", and per block of backward closures, "bwd:gen_BA.decode#2").
Kernels are attributed through their position in the launch order: ranges are host intervals, kernels are enqueued by the same host thread in
program order, so the n-th launch issued inside a range is the n-th kernel (ordered by dispatch) after the launches before the range."""
import sqlite3, sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]


def cols(t):
    return [r[1] for r in cur.execute("pragma table_info('%s')" % t)]


kt = "kernels" if "kernels" in tables else [t for t in tables if "kernel" in t.lower()][0]
kc = cols(kt)
name_col = "name" if "name" in kc else [c for c in kc if "name" in c][0]
kern = cur.execute("select %s, start, end from %s order by start" % (name_col, kt)).fetchall()
# marker / region table: anything with start, end and a name-like column that holds our labels
marks = []
for t in tables:
    c = cols(t)
    if "start" in c and "end" in c and t != kt:
        ncol = [x for x in c if x in ("name", "message", "region_name", "label")] or [x for x in c if "name" in x]
        if not ncol:
            continue
        try:
            rows = cur.execute("select %s, start, end from '%s'" % (ncol[0], t)).fetchall()
        except Exception:
            continue
        rows = [r for r in rows if r[0] and any(str(r[0]).startswith(p) for p in ("gen_", "dis_", "bwd:"))]
        if rows:
            marks = rows
            break
if not marks:
    print("no roctx ranges found; tables:", tables)
    sys.exit(0)
# host ranges and device kernels live on different clocks in general; rocprofv3 reports both in the same (system) domain, and a kernel
# cannot START before it was enqueued: attribute a kernel to the range whose host interval contains its enqueue ... which the kernel trace
# does not record.  Approximation that is exact for a serialised stream: kernels in start order are consumed range by range, each range
# taking the kernels that start before the NEXT range's first kernel; the split point is found by matching counts through the gaps between
# ranges (every launch happens inside some range except the few glue kernels between passes, which are reported as "(between passes)").
marks.sort(key=lambda r: r[1])
out = defaultdict(lambda: [0, 0.0])
ki = 0
for i, (label, s, e) in enumerate(marks):
    nxt = marks[i + 1][1] if i + 1 < len(marks) else None
    # kernels that started before this range opened on the host belong to what came before
    while ki < len(kern) and kern[ki][1] < s:
        out["(between passes)"][0] += 1; out["(between passes)"][1] += kern[ki][2] - kern[ki][1]; ki += 1
    while ki < len(kern) and (nxt is None or kern[ki][1] < nxt):
        out[str(label)][0] += 1; out[str(label)][1] += kern[ki][2] - kern[ki][1]; ki += 1
tot = sum(v[1] for v in out.values())
print("%-34s %8s %12s %7s" % ("range", "kernels", "kernel_us", "pct"))
for k, (n, t) in sorted(out.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %8d %12.1f %7.2f" % (k[:34], n, t / 1e3, 100.0 * t / max(tot, 1)))
print("%-34s %8d %12.1f" % ("TOTAL", sum(v[0] for v in out.values()), tot / 1e3))
