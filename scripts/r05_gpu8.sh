#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r05_8; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor"
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "median", c["ms_per_step_median"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
timeout 300 $B > $OUT/base.json 2>/dev/null; summ $OUT/base.json
ACLGAN_LANE_PRIO=-1 timeout 300 $B > $OUT/lane_low.json 2>/dev/null; summ $OUT/lane_low.json
ACLGAN_SIDE_PRIO=-1 timeout 300 $B > $OUT/side_low.json 2>/dev/null; summ $OUT/side_low.json
ACLGAN_SIDE_PRIO=-1 ACLGAN_LANE_PRIO=-1 timeout 300 $B > $OUT/both_low.json 2>/dev/null; summ $OUT/both_low.json
ACLGAN_SIDE_PRIO=1 timeout 300 $B > $OUT/side_high.json 2>/dev/null; summ $OUT/side_high.json
timeout 300 $B > $OUT/base2.json 2>/dev/null; summ $OUT/base2.json
