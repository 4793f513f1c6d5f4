#!/bin/bash
# round 6, GPU script 17: conv_dgrad16s alone (operator probe + kernel trace): interior-first enumeration against raster order, bf16 B=8 and fp16 B=32
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_17; mkdir -p $OUT
export TMPDIR=/tmp
for rl in 1 0 1 0; do
  for cfg in "bf16 ResBlock" "fp16 ResBlock B=32"; do
    set -- $cfg; dt=$1; shift; tag="$*"
    rm -rf /tmp/prof_p
    ACLGAN_DGRAD16S_RINGLAST=$rl PROBE_ONLY="$tag" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python scripts/probe16s.py $dt > $OUT/log.txt 2>&1
    DB=$(find /tmp/prof_p -name "*.db" | head -1)
    echo "ringlast=$rl $dt $tag: $(python scripts/rocpd_stats.py $DB 2>/dev/null | grep -E "conv_dgrad16s|conv_fold_st" | awk '{print $1, "avg_us", $(NF-3), "min", $(NF-2)}' | tr '\n' ';') | $(grep -E "dgrad16s" $OUT/log.txt | tail -1 | sed 's/.*dgrad16s+fold/dgrad16s+fold/' | cut -c1-40)" | tee -a $OUT/summary.txt
  done
done
