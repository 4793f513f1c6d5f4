#!/usr/bin/env python3
"""idle time between consecutive kernels of a rocprofv3 rocpd kernel trace: how much of the wall time no kernel was running
(launch / dispatch gaps), and its distribution.  usage: rocpd_gaps.py DB"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = sorted(cur.execute("select start, end from kernels").fetchall())
busy = sum(e - s for s, e in rows)
gaps = []
cur_end = rows[0][1]
for s, e in rows[1:]:
    if s > cur_end: gaps.append(s - cur_end)
    cur_end = max(cur_end, e)
span = rows[-1][1] - rows[0][0]
small = [g for g in gaps if g < 50_000]          # < 50 us: dispatch gaps; larger ones are host-side pauses (set-up, sync points)
print("kernels %d  span %.1f ms  busy %.1f ms  idle %.1f ms" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
print("gaps < 50 us: %d, total %.2f ms, mean %.2f us, median %.2f us" % (len(small), sum(small) / 1e6, sum(small) / max(1, len(small)) / 1e3,
                                                                     sorted(small)[len(small) // 2] / 1e3 if small else 0))
print("gaps >= 50 us: %d, total %.1f ms" % (len(gaps) - len(small), (sum(gaps) - sum(small)) / 1e6))
