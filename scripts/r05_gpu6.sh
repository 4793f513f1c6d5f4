#!/bin/bash
# round 5, GPU call 6: PMC passes of the 16-bit ResBlock forward kernel, three schedules
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r05_6; mkdir -p $OUT
export TMPDIR=/tmp
for m in 0 1 2; do python scripts/probe_fwd16.py bf16 $m 2>&1 | grep fwd16 | tee -a $OUT/timing.txt; done
python scripts/probe_fwd16.py bf16 1 32 2>&1 | grep fwd16 | tee -a $OUT/timing.txt
python scripts/probe_fwd16.py bf16 0 32 2>&1 | grep fwd16 | tee -a $OUT/timing.txt
for m in 0 1 2; do
  rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pmc_a -o p -- python scripts/probe_fwd16.py bf16 $m > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d /tmp/pmc_b -o p -- python scripts/probe_fwd16.py bf16 $m > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT -d /tmp/pmc_c -o p -- python scripts/probe_fwd16.py bf16 $m > /dev/null 2>&1
  echo "== patch mode $m" >> $OUT/pmc_fwd16.txt
  for d in a b c; do python scripts/pmc_dump.py $(find /tmp/pmc_$d -name "*.db" | head -1) conv_fwd16 >> $OUT/pmc_fwd16.txt 2>&1; done
done
cat $OUT/pmc_fwd16.txt | cut -c1-140
