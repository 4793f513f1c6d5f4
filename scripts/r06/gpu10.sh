#!/bin/bash
# round 6, GPU script 10: launch-floor A/B (prefill lane, thin kernels), streams created AFTER the trainer with / without aclgan_warm_streams
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_10; mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "floor", c.get("launch_bound_floor_ms_per_step"), "small", (c.get("small_batch") or {}).get("ms_per_step"), "launches", c["kernel_launches_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
F="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
timeout 400 $F > $OUT/floor_default.json 2>/dev/null; summ $OUT/floor_default.json
ACLGAN_PREFILL_LANE=0 timeout 400 $F > $OUT/floor_prefill_lane0.json 2>/dev/null; summ $OUT/floor_prefill_lane0.json
ACLGAN_THININ2=0 timeout 400 $F > $OUT/floor_thinin2_off.json 2>/dev/null; summ $OUT/floor_thinin2_off.json
ACLGAN_NOWINOS2=1 timeout 400 $F > $OUT/floor_s2k4_off.json 2>/dev/null; summ $OUT/floor_s2k4_off.json
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor"
for warm in 1 0; do for ps in 1 2; do for ln in 2 3; do
  ACLGAN_WARM_STREAMS=$warm timeout 300 $B --lanes $ln --post-streams $ps > $OUT/bench_lanes${ln}_post${ps}_warm${warm}.json 2>/dev/null; summ $OUT/bench_lanes${ln}_post${ps}_warm${warm}.json
done; done; done
ACLGAN_WARM_STREAMS=1 timeout 300 $B --lanes 3 --pre-streams 1 > $OUT/bench_lanes3_pre1_warm1.json 2>/dev/null; summ $OUT/bench_lanes3_pre1_warm1.json
