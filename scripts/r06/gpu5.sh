#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_5; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -g -o /tmp/capture_lanes scripts/debug/capture_lanes.hip > $OUT/capture_build.log 2>&1
ulimit -s; timeout 600 /tmp/capture_lanes sweep > $OUT/capture_sweep.txt 2>&1; cat $OUT/capture_sweep.txt | paste - - | cut -c 1-150
echo "== with ulimit -s unlimited"; (ulimit -s unlimited; ulimit -s; CAP_N=2400 CAP_E=8 timeout 300 /tmp/capture_lanes 2>&1 | grep -A1 "variant 8") | tee $OUT/capture_unlimited_stack.txt
echo "== backtrace"; (CAP_N=2400 CAP_E=8 timeout 120 gdb -batch -ex "set follow-fork-mode child" -ex run -ex bt /tmp/capture_lanes 2>&1 | grep -E "^#|SIGSEGV" | head -40) | tee $OUT/capture_backtrace.txt
