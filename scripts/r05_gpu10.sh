#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r05_10; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor"
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"], "frac", d["roofline"]["frac"], "launches", c["kernel_launches_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
timeout 300 $B --deterministic > $OUT/det.json 2>/dev/null; summ $OUT/det.json
timeout 300 $B --lanes 1 > $OUT/lanes1.json 2>/dev/null; summ $OUT/lanes1.json
timeout 300 $B --lanes 2 > $OUT/lanes2.json 2>/dev/null; summ $OUT/lanes2.json
timeout 300 $B --config configs/selfie2anime.yaml --deterministic > $OUT/bf16_det.json 2>/dev/null; summ $OUT/bf16_det.json
timeout 300 $B --graph > $OUT/graph.json 2>/dev/null; summ $OUT/graph.json
