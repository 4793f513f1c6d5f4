#!/bin/bash
# round 6, GPU script 1: the 4x4 stride-2 layers through the fused Winograd kernel -- operator tests, timing per shape, step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stride2 or conv_block_fwd or conv_fwd or conv_dgrad or winograd_fused_kernel" > $OUT/pytest_ops.log 2>&1; tail -5 $OUT/pytest_ops.log
timeout 300 python scripts/bench_s2k4.py > $OUT/bench_s2k4.txt 2>&1; cat $OUT/bench_s2k4.txt
rm -rf /tmp/prof_s2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s2 -o p -- python scripts/bench_s2k4.py > $OUT/prof_s2.log 2>&1
DB=$(find /tmp/prof_s2 -name "*.db" | head -1)
python scripts/rocpd_bygrid.py $DB 1 "" 80 > $OUT/s2k4_by_grid.txt 2>&1; head -60 $OUT/s2k4_by_grid.txt
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
ACLGAN_NOWINOS2=1 timeout 300 $B > $OUT/bench_nos2.json 2>/dev/null; summ $OUT/bench_nos2.json
ACLGAN_WINO_FUSED=2 timeout 300 $B > $OUT/bench_forced.json 2>/dev/null; summ $OUT/bench_forced.json
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest_fullsize.log 2>&1; tail -15 $OUT/pytest_fullsize.log
