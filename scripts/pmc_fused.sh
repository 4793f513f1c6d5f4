#!/bin/bash
# PMC passes + kernel trace of the fused Winograd forward on the ResBlock shape.  usage: bash scripts/pmc_fused.sh <outdir> [mode]
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/pmc_fused}; MODE=${2:-1}
mkdir -p $O
export TMPDIR=/tmp ACLGAN_WINO_FUSED=$MODE
rm -rf /tmp/pf_trace
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_trace -o p -- python scripts/probe_wino.py fwd > $O/trace.log 2>&1
DB=$(find /tmp/pf_trace -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/kernel_stats.txt 2>&1
head -8 $O/kernel_stats.txt | cut -c 1-150
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/pf_$i
    timeout 300 rocprofv3 --pmc $set -d /tmp/pf_$i -o p -- python scripts/probe_wino.py fwd > $O/pmc_$i.log 2>&1
    DB=$(find /tmp/pf_$i -name "*.db" | head -1)
    if [ -n "$DB" ]; then python scripts/pmc_dump.py $DB wino_fused >> $O/pmc.txt 2>&1; else echo "set $i failed: $set" >> $O/pmc.txt; tail -3 $O/pmc_$i.log >> $O/pmc.txt; fi
done
cat $O/pmc.txt
