#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_9; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sweep.py tests/test_gpu_step.py -m gpu -q -x -k "conv_fwd or conv_dgrad or conv_wgrad or thin or sweep or update_steps or forward_matches" 2>&1 | tail -15) | tee $OUT/pytest.log
timeout 300 python scripts/probe_thin.py 10 2>&1 | grep -v amdgpu | tee $OUT/probe_thin_new.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor"
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "frac", r["frac"], "launches", c["kernel_launches_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
timeout 300 $B > $OUT/bench_new.json 2>$OUT/bench_new.err; summ $OUT/bench_new.json
timeout 300 $B > $OUT/bench_new2.json 2>/dev/null; summ $OUT/bench_new2.json
timeout 300 $B --dtype bf16 > $OUT/bench_bf16_new.json 2>/dev/null; summ $OUT/bench_bf16_new.json
