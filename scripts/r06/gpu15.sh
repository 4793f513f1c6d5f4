#!/bin/bash
# round 6, GPU script 15: the halo ring of the 3x3 ResBlock input gradient: tile shape x split count (kernel trace of the operator alone)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_15; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "0 0" "0 1" "0 2" "0 3" "0 4" "1 0" "1 2" "1 3" "2 0" "2 2" "2 3" "2 4" "3 0" "3 2" "3 3"; do
  set -- $cfg
  rm -rf /tmp/prof_h
  ACLGAN_HALO_TILE=$1 ACLGAN_HALO_SPLIT=$2 ONLY=res3x3 REPS=20 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o p -- python scripts/bench_conv.py dgrad > $OUT/log_$1_$2.txt 2>&1
  DB=$(find /tmp/prof_h -name "*.db" | head -1)
  echo "tile $1 split $2: $(python scripts/rocpd_stats.py $DB 2>/dev/null | grep -E "conv_dgrad_fast" | awk '{print $1, $2, $3, $4, "calls", $(NF-5), "avg_us", $(NF-3)}' | tr '\n' ';')  $(grep res3x3 $OUT/log_$1_$2.txt | tail -1)" | tee -a $OUT/summary.txt
done
