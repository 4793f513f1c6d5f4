#!/bin/bash
# round 6, GPU script 7: the image-side (thin-channel) layers in isolation: timing, kernel trace, two PMC passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_7; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/probe_thin.py 10 > $OUT/probe_thin.txt 2>&1; cat $OUT/probe_thin.txt
rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python scripts/probe_thin.py 3 > /dev/null 2>&1
python scripts/rocpd_bygrid.py $(find /tmp/pt -name "*.db" | head -1) 5 "" 60 > $OUT/thin_by_grid.txt 2>&1; cat $OUT/thin_by_grid.txt | cut -c 1-150
rm -rf /tmp/pa /tmp/pb
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pa -o p -- python scripts/probe_thin.py 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d /tmp/pb -o p -- python scripts/probe_thin.py 2 > /dev/null 2>&1
(python scripts/pmc_dump.py $(find /tmp/pa -name "*.db" | head -1) conv_; python scripts/pmc_dump.py $(find /tmp/pb -name "*.db" | head -1) conv_) > $OUT/thin_pmc.txt 2>&1; cat $OUT/thin_pmc.txt | cut -c 1-140
