#!/bin/bash
# ablations of conv_fwd16p (measurement build: -DACLGAN_FWD16P_ABLATION): what a k-tile costs without its MFMAs / fragment reads / copies
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r05_7; mkdir -p $OUT
for m in 1 17 33 65 97 113 81 49; do echo "mode $m (ablation bits $((m>>4)): 1 no MFMA, 2 no fragment reads, 4 no copies)"; python scripts/probe_fwd16.py bf16 $m 2>&1 | grep fwd16; done | tee $OUT/ablation.txt
