#!/bin/bash
# round 6, GPU script 12: streaming kernels without per-element divisions (norm_apply / norm_bwd_apply / act_bwd / fold): tests + benches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_12; mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "floor", c.get("launch_bound_floor_ms_per_step"), "small", (c.get("small_batch") or {}).get("ms_per_step"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
(timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops16s.py tests/test_gpu_ops16.py tests/test_gpu_step16.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -5) | tee $OUT/tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
for i in 1 2; do timeout 400 $B > $OUT/fp32_$i.json 2>/dev/null; summ $OUT/fp32_$i.json; done
for i in 1 2; do timeout 400 $B --no-launch-floor --dtype bf16 > $OUT/bf16_$i.json 2>/dev/null; summ $OUT/bf16_$i.json; done
for i in 1 2; do timeout 400 $B --no-launch-floor --dtype fp16 > $OUT/fp16_$i.json 2>/dev/null; summ $OUT/fp16_$i.json; done
for i in 1 2; do ACLGAN_GLDS_TILE=4 timeout 400 $B --no-launch-floor --dtype fp16 > $OUT/fp16_tile4_$i.json 2>/dev/null; summ $OUT/fp16_tile4_$i.json; done
timeout 400 $B --no-launch-floor --config configs/glasses_removal.yaml > $OUT/fp32_512.json 2>/dev/null; summ $OUT/fp32_512.json
