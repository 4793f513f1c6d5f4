#!/bin/bash
# round 6, GPU script 18: stream priorities of the parameter-gradient stream / lanes 1.. on the final build (same box, alternating)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_18; mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor"
for i in 1 2; do
  run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/${name}_$i.json 2>/dev/null; summ $OUT/${name}_$i.json; }
  run default A=1
  run side_lo ACLGAN_SIDE_PRIO=-1
  run side_hi ACLGAN_SIDE_PRIO=1
  run lanes_lo ACLGAN_LANE_PRIO=-1
  run side_lo_lanes_lo ACLGAN_SIDE_PRIO=-1 ACLGAN_LANE_PRIO=-1
done
