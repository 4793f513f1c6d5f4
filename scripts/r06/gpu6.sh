#!/bin/bash
# round 6, GPU script 6: capture rings, split-bf16 loop microbenchmark, lane / determinism / frozen-mask tests of the prefill-on-a-lane build, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_6; mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/capture_lanes scripts/debug/capture_lanes.hip > $OUT/capture_build.log 2>&1
timeout 600 /tmp/capture_lanes rings > $OUT/capture_rings.txt 2>&1; paste - - - < $OUT/capture_rings.txt | cut -c 1-200
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Iscripts/microbench -o /tmp/split_bf16_loop scripts/microbench/split_bf16_loop.hip > $OUT/split_build.log 2>&1
timeout 300 /tmp/split_bf16_loop > $OUT/split_bf16_loop.txt 2>&1; cat $OUT/split_bf16_loop.txt
(timeout 1500 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_determinism.py tests/test_gpu_maskfrozen.py tests/test_gpu_graph.py -m gpu -q -s 2>&1 | grep -vE "^\s*$" | cut -c 1-700) > $OUT/pytest.log; grep -E "passed|failed|worst gradient tensors, masks FROZEN|mean over seeds|Error|^E " $OUT/pytest.log | cut -c 1-400 | head -40
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor"
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "frac", r["frac"], "launches", c["kernel_launches_per_step"], "flop", r.get("flop_per_launch"), r.get("flop_per_launch_counter"), r.get("flop_counter_vs_model"), "small", (c.get("small_batch") or {}).get("ms_per_step"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
timeout 300 $B > $OUT/bench_a.json 2>$OUT/bench_a.err; summ $OUT/bench_a.json
timeout 300 $B > $OUT/bench_b.json 2>/dev/null; summ $OUT/bench_b.json
timeout 300 $B --lanes 2 > $OUT/bench_lanes2.json 2>/dev/null; summ $OUT/bench_lanes2.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --batch 3 > $OUT/bench_b3.json 2>/dev/null; summ $OUT/bench_b3.json
