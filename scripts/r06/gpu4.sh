#!/bin/bash
# round 6, GPU script 4: band enumeration of the sub-pixel dgrad ring (tests, A/B, trace), capture repro
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_4; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_ops.py tests/test_gpu_ops16.py -m gpu -q -x -k "forward_only or tuning_change or conv_dgrad or dgrad" 2>&1 | tail -40) > $OUT/pytest_misc.log; tail -6 $OUT/pytest_misc.log | cut -c 1-300
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor"
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "frac", d["roofline"]["frac"], "launches", c["kernel_launches_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
timeout 300 $B > $OUT/bench_default.json 2>$OUT/bench_default.err; summ $OUT/bench_default.json
ACLGAN_UP5_BANDFOLD=0 timeout 300 $B > $OUT/bench_nobandfold.json 2>/dev/null; summ $OUT/bench_nobandfold.json
timeout 300 $B --lanes 1 > $OUT/bench_lanes1.json 2>/dev/null; summ $OUT/bench_lanes1.json
ACLGAN_UP5_BANDFOLD=0 timeout 300 $B --lanes 1 > $OUT/bench_lanes1_nobandfold.json 2>/dev/null; summ $OUT/bench_lanes1_nobandfold.json
timeout 300 $B --dtype bf16 > $OUT/bench_bf16.json 2>/dev/null; summ $OUT/bench_bf16.json
rm -rf /tmp/prof_t; ACLGAN_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor --no-other-configs > $OUT/prof.log 2>&1
DB=$(find /tmp/prof_t -name "*.db" | head -1); python scripts/rocpd_bygrid.py $DB 6 "" 100000 > $OUT/by_grid_ss0.txt 2>&1; python scripts/rocpd_stats.py $DB > $OUT/kernel_stats_ss0.txt 2>&1; grep -E "dgrad_fast|fold_band" $OUT/by_grid_ss0.txt | head -12; tail -1 $OUT/kernel_stats_ss0.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/capture_lanes scripts/debug/capture_lanes.hip > $OUT/capture_build.log 2>&1
timeout 300 /tmp/capture_lanes > $OUT/capture_lanes.txt 2>&1; cat $OUT/capture_lanes.txt
(ACLGAN_CAPTURE_LANES=1 timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -q -x -k "fp32" 2>&1 | tail -25) > $OUT/graph_with_lanes.log; tail -12 $OUT/graph_with_lanes.log | cut -c 1-400
