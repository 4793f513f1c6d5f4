#!/bin/bash
# round 5, GPU call 1: lane scheduler correctness + A/B of the scheduler switches on the headline workload
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r05_1; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_lanes.py "tests/test_gpu_ddp.py::test_bucket_callback_fires_after_the_last_writer" -x -q -s 2>&1 | tail -40 ) > $OUT/lanes_tests.log 2>&1
echo "lanes tests rc=$?" >> $OUT/lanes_tests.log
for L in 1 2 3 4; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --lanes $L > $OUT/bench_lanes$L.json 2> $OUT/bench_lanes$L.err
done
ACLGAN_U_BATCH=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --lanes 1 > $OUT/bench_lanes1_nobatch.json 2> $OUT/bench_lanes1_nobatch.err
ACLGAN_SIDE_STREAM=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor > $OUT/bench_noside.json 2> $OUT/bench_noside.err
( timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_step.py tests/test_gpu_graph.py -x -q -s 2>&1 | tail -60 ) > $OUT/det_step_tests.log 2>&1
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "launches", c["kernel_launches_per_step"], "floor", c["launch_bound_floor_ms_per_step"], "small", (c.get("small_batch") or {}).get("ms_per_step"), "lanes", c.get("lanes"))
except Exception as e: print("ERR", e)
PY
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt; tail -5 $OUT/lanes_tests.log; tail -5 $OUT/det_step_tests.log
