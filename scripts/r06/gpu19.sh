#!/bin/bash
# round 6, GPU script 19: where the GPU idles inside the default 3-lane step (union of kernel intervals against the wall span)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_19; mkdir -p $OUT
export TMPDIR=/tmp
for dt in fp32 bf16; do
  rm -rf /tmp/prof_g
  timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_g -o p -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor --no-other-configs --dtype $dt > $OUT/prof_$dt.log 2>&1
  DB=$(find /tmp/prof_g -name "*.db" | head -1)
  # window: four whole steps of the timed region (between Adam launches)
  python scripts/rocpd_gaps.py $DB -4 15 > $OUT/gaps_$dt.txt 2>&1
  cat $OUT/gaps_$dt.txt
done
