#!/bin/bash
# round 6, GPU script 11: the 16-bit switches tuned in rounds 3-5 on ONE queue, measured again under the lane scheduler (bf16 B=8, fp16 B=32)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_11; mkdir -p $OUT
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
for dt in bf16 fp16; do
  run() { name=$1; shift; env "$@" timeout 300 $B --dtype $dt > $OUT/${dt}_$name.json 2>/dev/null; summ $OUT/${dt}_$name.json; }
  run default A=1
  run dgrad_direct ACLGAN_DGRAD16S_DIRECT=1
  run tile4 ACLGAN_GLDS_TILE=4
  run patch0 ACLGAN_FWD16_PATCH=0
  run patch2 ACLGAN_FWD16_PATCH=2
  run co16_0 ACLGAN_CO16=0
  run act16_0 ACLGAN_ACT16=0
  run nostatfuse ACLGAN_NOSTATFUSE=1
  run lanes2 ACLGAN_LANES=2
  run prefill0 ACLGAN_PREFILL_LANE=0
  run default2 A=1
done
