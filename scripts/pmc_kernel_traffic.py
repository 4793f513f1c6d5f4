#!/usr/bin/env python3
"""memory-side bytes per launch of ONE kernel from two rocprofv3 PMC databases -> json bound to the build:
    python scripts/pmc_kernel_traffic.py <fetch.db> <write.db> <kernel substring> <out.json>
bytes_per_launch = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes, averaged over the launches."""
import hashlib, json, os, sqlite3, sys


def avg(dbp, counter, pat):
    cur = sqlite3.connect(dbp).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ci = {c: i for i, c in enumerate(cols)}
    n, tot = 0, 0.0
    for r in cur.execute("select * from counters_collection"):
        if r[ci["counter_name"]] == counter and pat in str(r[ci.get("kernel_name", ci.get("name", 0))]):
            n += 1; tot += float(r[ci["value"]])
    return n, (tot / n if n else 0.0)


fdb, wdb, pat, out = sys.argv[1:5]
nf, f = avg(fdb, "FETCH_SIZE", pat); nw, w = avg(wdb, "WRITE_SIZE", pat)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ent = {"kernel": pat, "launches": [nf, nw], "fetch_bytes": 2.0 * 1024.0 * f, "write_bytes": 1024.0 * w, "bytes_per_launch": 2.0 * 1024.0 * f + 1024.0 * w,
       "lib_md5": hashlib.md5(open(os.path.join(root, "acl-gan_amd", "libaclgan_hip.so"), "rb").read()).hexdigest(), "head": os.environ.get("ACLGAN_HEAD"),
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/probe_wino.py fwd; 2 x FETCH_SIZE + WRITE_SIZE"}
json.dump(ent, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(ent))
