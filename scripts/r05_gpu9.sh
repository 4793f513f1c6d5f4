#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r05_9; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "floor", c["launch_bound_floor_ms_per_step"], "small", (c.get("small_batch") or {}).get("ms_per_step"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
timeout 300 $B > $OUT/a1.json 2>/dev/null; summ $OUT/a1.json
ACLGAN_NO_PRIVATE_SIDE=1 timeout 300 $B > $OUT/b1.json 2>/dev/null; summ $OUT/b1.json
timeout 300 $B > $OUT/a2.json 2>/dev/null; summ $OUT/a2.json
ACLGAN_NO_PRIVATE_SIDE=1 timeout 300 $B > $OUT/b2.json 2>/dev/null; summ $OUT/b2.json
