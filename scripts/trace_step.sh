#!/bin/bash
# kernel trace of the bench step -> per-kernel and per-grid tables.  usage: bash scripts/trace_step.sh <outdir> [bench args]; env (e.g. ACLGAN_SIDE_STREAM=0) passes through
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/trace}; shift
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/prof_step
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_step -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor "$@" > $O/prof.log 2>&1
DB=$(find /tmp/prof_step -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/kernel_stats.txt 2>&1
python scripts/rocpd_bygrid.py $DB 6 "" 100 > $O/by_grid.txt 2>&1
head -40 $O/kernel_stats.txt | cut -c 1-150; tail -1 $O/kernel_stats.txt
