#!/bin/bash
# round 6, GPU script 2: frozen-mask backward parity, advisor fixes, kernel traces + PMC passes (traffic, MFMA instructions) of the new build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_2; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_maskfrozen.py -m gpu -q -s 2>&1 | grep -vE "^\s*$" | cut -c 1-1200) > $OUT/pytest_maskfrozen.log; grep -E "passed|failed|error|matched|worst|frozen|Error|assert" $OUT/pytest_maskfrozen.log | cut -c 1-600 | head -60
(timeout 600 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_ops.py -m gpu -q -x -k "forward_only or tuning_change or stride2 or conv_block_fwd" 2>&1 | tail -5) | tee $OUT/pytest_misc.log
STAGES="trace trace_ss0 traffic" bash scripts/evidence_r06.sh 2>&1 | tail -60
