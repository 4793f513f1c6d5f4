# L2 hit rate and memory-side fetch of the 16-bit conv kernels on the ResBlock shape (separate PMC passes, no tracing)
export TMPDIR=/tmp; O=gpurun_out/r03_pmc16; mkdir -p $O; rm -f $O/*.txt
for sp in ${SPECS:-0}; do
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" ; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  PROBE_ONLY=ResBlock ACLGAN_GLDS_SPEC=$sp timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_$tag -o p -- python scripts/probe16s.py > $O/log_$tag.txt 2>&1
  DB=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  echo "## spec=$sp $c" >> $O/pmc.txt
  python scripts/pmc_dump.py $DB "16" >> $O/pmc.txt 2>&1
done; done
cut -c1-160 $O/pmc.txt
