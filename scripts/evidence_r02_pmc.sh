set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final
export TMPDIR=/tmp
mkdir -p $O
rm -f $O/pmc_winograd_traffic.txt $O/pmc_winograd_mfma.txt
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_$c -o p -- python scripts/probe_wino.py fwd > /dev/null 2>&1
    DB=$(find /tmp/pmc_$c -name "*.db" | head -1)
    echo "## $c (KiB per launch)" >> $O/pmc_winograd_traffic.txt
    python scripts/pmc_dump.py $DB "" | grep -v "at::native" >> $O/pmc_winograd_traffic.txt 2>&1
done
for w in fwd wgrad; do
    rm -rf /tmp/pmc_sq_$w
    timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA -d /tmp/pmc_sq_$w -o p -- python scripts/probe_wino.py $w > /dev/null 2>&1
    DB=$(find /tmp/pmc_sq_$w -name "*.db" | head -1)
    echo "## $w" >> $O/pmc_winograd_mfma.txt
    python scripts/pmc_dump.py $DB "" | grep -v "at::native\|fillBuffer" >> $O/pmc_winograd_mfma.txt 2>&1
done
cat $O/pmc_winograd_traffic.txt $O/pmc_winograd_mfma.txt | cut -c 1-150
