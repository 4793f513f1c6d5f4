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
import json
marks = []
if "regions" in tables:
    for ext, st, en in cur.execute("select extdata, start, end from regions order by start").fetchall():
        try:
            msg = json.loads(ext).get("message")
        except Exception:
            msg = None
        if msg and "@" in msg:
            marks.append((msg, st, en))
if not marks:
    print("no roctx ranges found (run with ACLGAN_ROCTX=1 and --marker-trace); tables:", [t for t in tables if "_0000" not in t])
    sys.exit(0)
# Kernels are asynchronous: a range's host interval says nothing about when its kernels ran.  Every range label carries the library's launch
# counter at its opening ("name#k@N"), a "~end@N" marker follows its close: the range owns the library launches [N_open, N_end) in LAUNCH
# ORDER = start order of the library's kernels on a single stream (trace with ACLGAN_SIDE_STREAM=0).  torch's own kernels and runtime fills
# are not counted by the library and are left out on both sides.
import re
lib = [k for k in kern if "at::native" not in str(k[0]) and "rocclr" not in str(k[0])]
out = defaultdict(lambda: [0, 0.0, 0])
allm = sorted(set(marks), key=lambda r: r[1])
covered = 0
for i, (label, s0, e0) in enumerate(allm):
    label = str(label)
    if label.startswith("~end"):
        continue
    m = re.match(r"(.*)@(\d+)$", label)
    if not m or i + 1 >= len(allm):
        continue
    n0 = int(m.group(2))
    m2 = re.match(r"~end@(\d+)$", str(allm[i + 1][0]))
    if not m2:
        continue
    n1 = int(m2.group(1))
    base = re.sub(r"#\d+$", lambda mm: mm.group(0), m.group(1))
    t = sum(k[2] - k[1] for k in lib[n0:n1])
    out[base][0] += n1 - n0; out[base][1] += t; out[base][2] += 1
    covered += n1 - n0
rest = len(lib) - covered
tot = sum(v[1] for v in out.values())
print("%-34s %6s %9s %12s %7s   (%d library kernels in the trace, %d outside any range)" % ("range", "times", "kernels", "kernel_us", "pct", len(lib), rest))
for k, (n, t, occ) in sorted(out.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %6d %9d %12.1f %7.2f" % (k[:34], occ, n, t / 1e3, 100.0 * t / max(tot, 1)))
print("%-34s %6s %9d %12.1f" % ("TOTAL (inside ranges)", "", sum(v[0] for v in out.values()), tot / 1e3))
