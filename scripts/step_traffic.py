#!/usr/bin/env python3
"""Memory-side bytes of ONE training step from two rocprofv3 PMC databases (FETCH_SIZE and WRITE_SIZE need separate passes):

    rocprofv3 --pmc FETCH_SIZE -d /tmp/f -o p -- python scripts/probe_step.py fp32 256 8 2
    rocprofv3 --pmc WRITE_SIZE -d /tmp/w -o p -- python scripts/probe_step.py fp32 256 8 2
    python scripts/step_traffic.py <fetch.db> <write.db> <steps> <key> [out.json] [summary.txt]

Sums the counter over every kernel the library launched (torch's own init / fill kernels excluded), divides by the number of steps,
applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read: x2) and merges
{key: {"bytes_per_step", "fetch_bytes", "write_bytes", ...}} into out.json (default profiles/r05_step_traffic.json).  Counter unit: KiB.
The entry is bound to the build it was measured on: md5 of acl-gan_amd/libaclgan_hip.so, the library's own launch count per step
(scripts/probe_step.py writes it to $PROBE_STEP_JSON) and the commit ($ACLGAN_HEAD: .git does not travel to the GPU box); bench.py compares
them with the running library and reports traffic_stale on a mismatch."""
import json, os, re, sqlite3, sys
from collections import defaultdict


def load(dbp, counter):
    db = sqlite3.connect(dbp); cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ci = {c: i for i, c in enumerate(cols)}
    per = defaultdict(lambda: [0, 0.0])
    for r in cur.execute("select * from counters_collection"):
        if r[ci["counter_name"]] != counter:
            continue
        name = str(r[ci.get("kernel_name", ci.get("name", 0))])
        if "at::native" in name or "rocclr" in name and "fillBuffer" not in name and "copyBuffer" not in name:
            continue
        short = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("aclgan::", "").replace("void ", ""))
        per[short][0] += 1; per[short][1] += float(r[ci["value"]])
    return per


def main():
    fdb, wdb, steps, key = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_step_traffic.json")
    summary = sys.argv[6] if len(sys.argv) > 6 else None
    f, w = load(fdb, "FETCH_SIZE"), load(wdb, "WRITE_SIZE")
    fetch = 2.0 * 1024.0 * sum(v[1] for v in f.values()) / steps      # gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x
    write = 1024.0 * sum(v[1] for v in w.values()) / steps
    ent = {"bytes_per_step": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "steps_traced": steps,
           "kernels_per_step": sum(v[0] for v in f.values()) / steps,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/probe_step.py; 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes"}
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        ent["lib_md5"] = hashlib.md5(open(os.path.join(root, "acl-gan_amd", "libaclgan_hip.so"), "rb").read()).hexdigest()
    except OSError:
        ent["lib_md5"] = None
    ent["head"] = os.environ.get("ACLGAN_HEAD")
    try:
        ent["launches_per_step"] = json.load(open(os.environ["PROBE_STEP_JSON"]))["launches_per_step"]
    except Exception:
        ent["launches_per_step"] = None
    try:
        cur = json.load(open(out))
    except Exception:
        cur = {}
    cur[key] = ent
    json.dump(cur, open(out, "w"), indent=1, sort_keys=True)
    lines = ["# %s: memory-side bytes per step = 2 x FETCH_SIZE + WRITE_SIZE summed over every kernel of %d traced steps / %d" % (key, steps, steps),
             "# total %.2f GB per step (fetch %.2f GB, write %.2f GB)" % ((fetch + write) / 1e9, fetch / 1e9, write / 1e9),
             "%-70s %8s %12s %12s" % ("kernel", "calls/st", "fetch MB/st", "write MB/st")]
    names = sorted(set(f) | set(w), key=lambda n: -(2 * f.get(n, [0, 0])[1] + w.get(n, [0, 0])[1]))
    for n in names[:40]:
        lines.append("%-70s %8.1f %12.1f %12.1f" % (n[:70], f.get(n, [0, 0])[0] / steps, 2 * 1024 * f.get(n, [0, 0])[1] / steps / 1e6, 1024 * w.get(n, [0, 0])[1] / steps / 1e6))
    txt = "\n".join(lines)
    print(txt)
    if summary:
        open(summary, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
