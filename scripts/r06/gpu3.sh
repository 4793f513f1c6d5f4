#!/bin/bash
# round 6, GPU script 3: frozen masks + loss signs, advisor tests (full logs), band fold of the sub-pixel dgrad ring (tests + A/B)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_3; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_ops.py -m gpu -q -x -k "forward_only or tuning_change or stride2 or conv_dgrad or winograd_fused_kernel" 2>&1 | tail -40) > $OUT/pytest_misc.log; tail -30 $OUT/pytest_misc.log | cut -c 1-400
(timeout 1200 python -m pytest tests/test_gpu_maskfrozen.py -m gpu -q -s 2>&1 | grep -vE "^\s*$" | cut -c 1-1200) > $OUT/pytest_maskfrozen.log; grep -E "passed|failed|error|matched|worst|frozen|sign|Error|assert" $OUT/pytest_maskfrozen.log | cut -c 1-500 | head -60
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
timeout 300 $B > $OUT/bench_default2.json 2>/dev/null; summ $OUT/bench_default2.json
timeout 300 $B --lanes 1 > $OUT/bench_lanes1.json 2>/dev/null; summ $OUT/bench_lanes1.json
ACLGAN_UP5_BANDFOLD=0 timeout 300 $B --lanes 1 > $OUT/bench_lanes1_nobandfold.json 2>/dev/null; summ $OUT/bench_lanes1_nobandfold.json
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "256_b8" 2>&1 | grep -E "worst|passed|failed|rel errors" | cut -c 1-700) | tee $OUT/pytest_fullsize_256.log
