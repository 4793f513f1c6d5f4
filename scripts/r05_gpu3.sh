#!/bin/bash
# round 5, GPU call 3: the patch-resident 16-bit forward kernel (parity + timing), stress test, a few switches under lanes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r05_5; mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; k=d["roofline"]["kernel"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "launches", c["kernel_launches_per_step"], "kernel", k["ms"], k["frac"], k["name"][:40])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
( timeout 600 python -m pytest tests/test_gpu_ops16s.py -x -q -k "fwd16p or epilogue_statistics or test_conv_fwd16s" 2>&1 | grep -vE "^\s*$" | tail -30 ) > $OUT/tests_patch.log 2>&1
tail -4 $OUT/tests_patch.log
B16="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor --config configs/selfie2anime.yaml"
ACLGAN_FWD16_PATCH=1 timeout 300 $B16 > $OUT/bench_bf16_patch1.json 2>/dev/null; summ $OUT/bench_bf16_patch1.json
ACLGAN_FWD16_PATCH=0 timeout 300 $B16 > $OUT/bench_bf16_patch0.json 2>/dev/null; summ $OUT/bench_bf16_patch0.json
ACLGAN_FWD16_PATCH=2 timeout 300 $B16 > $OUT/bench_bf16_patch2_lockstep.json 2>/dev/null; summ $OUT/bench_bf16_patch2_lockstep.json
ACLGAN_FWD16_PATCH=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-launch-floor --dtype fp16 > $OUT/bench_fp16_patch1.json 2>/dev/null; summ $OUT/bench_fp16_patch1.json
ACLGAN_FWD16_PATCH=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-launch-floor --dtype fp16 > $OUT/bench_fp16_patch0.json 2>/dev/null; summ $OUT/bench_fp16_patch0.json
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "stress" 2>&1 | tail -5 ) > $OUT/tests_stress.log 2>&1; tail -3 $OUT/tests_stress.log
( timeout 900 python -m pytest tests/test_gpu_step16.py -x -q -s -k "trajectory or per_gpu_batch" 2>&1 | grep -vE "^\s*$" | tail -120 ) > $OUT/tests_step16.log 2>&1; tail -6 $OUT/tests_step16.log
