#!/bin/bash
# round 5, GPU call 2: process-wide stream pool + 3-lane plan, norm mask, hardware queues, 16-bit variants, new 16-bit parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r05_2; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "launches", c["kernel_launches_per_step"], "floor", c["launch_bound_floor_ms_per_step"], "small", (c.get("small_batch") or {}).get("ms_per_step"), "lanes", c.get("lanes"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
( timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_step16.py tests/test_gpu_ops_misc.py -x -q -s 2>&1 | grep -vE "^\s*$" | tail -60 ) > $OUT/tests.log 2>&1
for L in 2 3; do timeout 300 $B --lanes $L > $OUT/bench_lanes$L.json 2> $OUT/bench_lanes$L.err; summ $OUT/bench_lanes$L.json; done
ACLGAN_NORM_MASK=0 timeout 300 $B --lanes 3 --no-launch-floor > $OUT/bench_lanes3_nomask.json 2>/dev/null; summ $OUT/bench_lanes3_nomask.json
GPU_MAX_HW_QUEUES=8 timeout 300 $B --lanes 3 > $OUT/bench_lanes3_q8.json 2>/dev/null; summ $OUT/bench_lanes3_q8.json
GPU_MAX_HW_QUEUES=8 timeout 300 $B --lanes 4 > $OUT/bench_lanes4_q8.json 2>/dev/null; summ $OUT/bench_lanes4_q8.json
# 16-bit: lanes, tile / specialised variants after the round-4 barrier fix
B16="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor --config configs/selfie2anime.yaml"
timeout 300 $B16 --lanes 1 > $OUT/bench_bf16_lanes1.json 2>/dev/null; summ $OUT/bench_bf16_lanes1.json
timeout 300 $B16 --lanes 3 > $OUT/bench_bf16_lanes3.json 2>/dev/null; summ $OUT/bench_bf16_lanes3.json
ACLGAN_GLDS_TILE=4 timeout 300 $B16 --lanes 3 > $OUT/bench_bf16_tile4.json 2>/dev/null; summ $OUT/bench_bf16_tile4.json
ACLGAN_GLDS_TILE=2 ACLGAN_GLDS_SPEC=1 timeout 300 $B16 --lanes 3 > $OUT/bench_bf16_tile2_spec.json 2>/dev/null; summ $OUT/bench_bf16_tile2_spec.json
ACLGAN_GLDS_SPEC=1 timeout 300 $B16 --lanes 3 > $OUT/bench_bf16_spec.json 2>/dev/null; summ $OUT/bench_bf16_spec.json
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-launch-floor --dtype fp16 --lanes 3 > $OUT/bench_fp16_b32.json 2>/dev/null; summ $OUT/bench_fp16_b32.json
tail -25 $OUT/tests.log
