#!/bin/bash
# round 6, GPU script 14: fp16 B=32 tile rule A/B in one box session (new default / forced 128-row tiles / largest-tile rule), + 16-bit op tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_14; mkdir -p $OUT
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
for i in 1 2; do
  timeout 400 $B --dtype fp16 > $OUT/fp16_default_$i.json 2>/dev/null; summ $OUT/fp16_default_$i.json
  ACLGAN_GLDS_TILE=1 timeout 400 $B --dtype fp16 > $OUT/fp16_tile1_$i.json 2>/dev/null; summ $OUT/fp16_tile1_$i.json
  ACLGAN_GLDS_TILE=4 timeout 400 $B --dtype fp16 > $OUT/fp16_tile4_$i.json 2>/dev/null; summ $OUT/fp16_tile4_$i.json
done
timeout 400 $B --dtype bf16 > $OUT/bf16_default.json 2>/dev/null; summ $OUT/bf16_default.json
(timeout 900 python -m pytest tests/test_gpu_ops16s.py tests/test_gpu_step16.py -m gpu -q -x 2>&1 | tail -3) | tee $OUT/tests.log
