#!/bin/bash
# round 6, GPU script 16: 16-bit input gradient, interior-first enumeration with short frame tiles: tests + A/B (ACLGAN_DGRAD16S_RINGLAST=0 = raster order)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_16; mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dis/gen", c["ms_dis_update"], c["ms_gen_update"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
(timeout 900 python -m pytest tests/test_gpu_ops16s.py tests/test_gpu_ops16.py tests/test_gpu_step16.py -m gpu -q -x 2>&1 | tail -3) | tee $OUT/tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-launch-floor"
for i in 1 2; do
  for dt in bf16 fp16; do
    timeout 400 $B --dtype $dt > $OUT/${dt}_new_$i.json 2>/dev/null; summ $OUT/${dt}_new_$i.json
    ACLGAN_DGRAD16S_RINGLAST=0 timeout 400 $B --dtype $dt > $OUT/${dt}_raster_$i.json 2>/dev/null; summ $OUT/${dt}_raster_$i.json
  done
done
rm -rf /tmp/prof_d16
ACLGAN_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d16 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor --no-other-configs --dtype bf16 > $OUT/prof.log 2>&1
python scripts/rocpd_bygrid.py $(find /tmp/prof_d16 -name "*.db" | head -1) 6 "" 100000 2>/dev/null | grep -E "dgrad16s" | head -8 | cut -c1-150
