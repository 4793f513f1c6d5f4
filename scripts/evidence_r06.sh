#!/bin/bash
# Round-6 evidence run on one MI355X (through gpurun).  Stages selected with STAGES="tests bench trace ..." (default: all).
#   ACLGAN_HEAD=$(git rev-parse --short HEAD) /usr/local/graft/bin/gpurun --timeout 2400 -- "ACLGAN_HEAD=$ACLGAN_HEAD STAGES='tests bench' bash scripts/evidence_r06.sh"
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06${TAG:+_$TAG}
mkdir -p $O
export TMPDIR=/tmp
STAGES=${STAGES:-"tests smoke bench trace trace_ss0 trace16 trace16_ss0 traffic traffic16 traffic16f traffic512 pmcfused prestreams split thin s2k4 capture"}
has() { [[ " $STAGES " == *" $1 "* ]]; }
B="python bench.py --no-cpu-baseline"
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); c = d["config"]; r = d["roofline"]; k = r["kernel"]
    print(sys.argv[1], d["value"], "img/s", d["ms_per_step"], "ms", "dis", c["ms_dis_update"], "gen", c["ms_gen_update"], "launches", c.get("kernel_launches_per_step"),
          "floor", c.get("launch_bound_floor_ms_per_step"), "small", (c.get("small_batch") or {}).get("ms_per_step"), "frac", r["frac"], "alg_frac", r.get("algorithmic_frac"),
          "kernel", k["frac"], k["ms"], "traffic", r.get("traffic"), "stale", r.get("traffic_stale"), "alg_bytes", r.get("algorithmic_bytes"), "cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
echo "== stages: $STAGES head ${ACLGAN_HEAD:-?}" | tee $O/progress.log
if has tests; then
    (timeout ${TEST_TIMEOUT:-1500} python -m pytest ${PYTEST_PATHS:-tests} -m gpu -q -s --durations=15 ${PYTEST_X--x} ${PYTEST_ARGS:-} 2>&1 | grep -vE "^\s*$" | cut -c 1-900) > $O/tests_full.log
    grep -E "passed|failed|error" $O/tests_full.log | tail -3 | tee -a $O/progress.log
    grep -E "worst|passed|failed|rel errors|shard equivalence|chained|under the floor|Error|assert|train loop|s call" $O/tests_full.log | cut -c 1-700 > $O/tests_summary.log
fi
if has smoke; then (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) | tee $O/smoke.log | tee -a $O/progress.log; fi
if has bench; then timeout 600 python bench.py > $O/bench_256_fp32.json 2> $O/bench_256_fp32.err; summ $O/bench_256_fp32.json | tee -a $O/progress.log; fi
if has bench512; then $B --config configs/glasses_removal.yaml > $O/bench_512_fp32.json 2>/dev/null; summ $O/bench_512_fp32.json | tee -a $O/progress.log; fi
if has bench16; then
    $B --config configs/selfie2anime.yaml > $O/bench_256_bf16.json 2>/dev/null; summ $O/bench_256_bf16.json | tee -a $O/progress.log
    $B --dtype fp16 > $O/bench_256_fp16_b32.json 2>/dev/null; summ $O/bench_256_fp16_b32.json | tee -a $O/progress.log
fi
if has benchdet; then $B --deterministic > $O/bench_256_fp32_deterministic.json 2>/dev/null; summ $O/bench_256_fp32_deterministic.json | tee -a $O/progress.log; fi
if has benchab; then      # what the fused weight gradient and the backward side stream are worth, alone and together
    ACLGAN_SIDE_STREAM=0 $B --no-launch-floor > $O/bench_256_fp32_side_stream_off.json 2>/dev/null; summ $O/bench_256_fp32_side_stream_off.json | tee -a $O/progress.log
    ACLGAN_WINO_WGRAD_FUSED=0 $B --no-launch-floor > $O/bench_256_fp32_pipeline_wgrad.json 2>/dev/null; summ $O/bench_256_fp32_pipeline_wgrad.json | tee -a $O/progress.log
    ACLGAN_WINO_WGRAD_FUSED=0 ACLGAN_SIDE_STREAM=0 $B --no-launch-floor > $O/bench_256_fp32_pipeline_wgrad_side_stream_off.json 2>/dev/null; summ $O/bench_256_fp32_pipeline_wgrad_side_stream_off.json | tee -a $O/progress.log
fi
if has probewg; then (timeout 300 python scripts/probe_wgrad_fused.py 2>&1 | grep -v amdgpu.ids) > $O/probe_wgrad_fused.txt; tail -9 $O/probe_wgrad_fused.txt | tee -a $O/progress.log; fi
if has benchold; then ACLGAN_WINO_FUSED=0 $B > $O/bench_256_fp32_three_launch_winograd.json 2>/dev/null; summ $O/bench_256_fp32_three_launch_winograd.json | tee -a $O/progress.log; fi
trace() {   # tag, bench args (env passes through)
    rm -rf /tmp/prof_$1
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor --no-other-configs $2 > $O/prof_$1.log 2>&1
    DB=$(find /tmp/prof_$1 -name "*.db" | head -1)
    python scripts/rocpd_stats.py $DB > $O/kernel_stats_$1.txt 2>&1
    python scripts/rocpd_bygrid.py $DB 6 "" 100000 > $O/by_grid_$1.txt 2>&1      # (round 5: the FULL table, tail included)
    head -12 $O/kernel_stats_$1.txt | cut -c 1-140 | tee -a $O/progress.log; tail -1 $O/kernel_stats_$1.txt | tee -a $O/progress.log
}
if has trace; then trace 256_fp32 ""; fi
if has trace_ss0; then ACLGAN_SIDE_STREAM=0 trace 256_fp32_side_stream_off ""; fi
if has trace_lanes1; then trace 256_fp32_lanes1 "--lanes 1"; fi      # one queue: kernel times without co-residency effects
if has trace16; then trace 256_bf16 "--dtype bf16"; fi
if has trace16_ss0; then ACLGAN_SIDE_STREAM=0 trace 256_bf16_side_stream_off "--dtype bf16"; fi      # one queue: the bf16 family table
if has trace16f_ss0; then ACLGAN_SIDE_STREAM=0 trace 256_fp16_b32_side_stream_off "--dtype fp16"; fi
if has trace512; then trace 512_fp32 "--config configs/glasses_removal.yaml"; fi
traffic() {   # dtype size batch
    export PROBE_STEP_JSON=$O/probe_step_$1_$2_b$3.json
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${c}_$1
        timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_${c}_$1 -o p -- python scripts/probe_step.py $1 $2 $3 2 > $O/pmc_${c}_$1.log 2>&1
    done
    F=$(find /tmp/pmc_FETCH_SIZE_$1 -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE_$1 -name "*.db" | head -1)
    python scripts/step_traffic.py $F $W 2 ${1}_${2}_b${3} $O/step_traffic.json $O/step_traffic_${1}_${2}_b${3}.txt | head -14 | cut -c 1-120 | tee -a $O/progress.log
    # round 6: the matrix-pipe FLOPs of the same step from SQ_INSTS_MFMA (what roofline.flop_per_launch is checked against)
    rm -rf /tmp/pmc_MFMA_$1
    timeout 600 rocprofv3 --pmc SQ_INSTS_MFMA -d /tmp/pmc_MFMA_$1 -o p -- python scripts/probe_step.py $1 $2 $3 2 > $O/pmc_MFMA_$1.log 2>&1
    M=$(find /tmp/pmc_MFMA_$1 -name "*.db" | head -1)
    python scripts/step_mfma_flops.py $M 2 ${1}_${2}_b${3} $O/step_traffic.json $O/step_mfma_flops_${1}_${2}_b${3}.txt | head -12 | cut -c 1-120 | tee -a $O/progress.log
}
if has traffic; then traffic fp32 256 8; fi
if has traffic16; then traffic bf16 256 8; fi
if has traffic16f; then traffic fp16 256 32; fi
if has traffic512; then traffic fp32 512 4; fi
if has pmcfused; then
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmcf_$c
        timeout 300 rocprofv3 --pmc $c -d /tmp/pmcf_$c -o p -- python scripts/probe_wino.py fwd > $O/pmcf_$c.log 2>&1
    done
    python scripts/pmc_kernel_traffic.py $(find /tmp/pmcf_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmcf_WRITE_SIZE -name "*.db" | head -1) wino_fused_kernel $O/pmc_wino_fused.json | cut -c 1-300 | tee -a $O/progress.log
    rm -rf /tmp/pmcf_sq
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pmcf_sq -o p -- python scripts/probe_wino.py fwd > /dev/null 2>&1
    python scripts/pmc_dump.py $(find /tmp/pmcf_sq -name "*.db" | head -1) wino_fused > $O/pmc_wino_fused_sq.txt 2>&1
    rm -rf /tmp/pmcf_sq2
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d /tmp/pmcf_sq2 -o p -- python scripts/probe_wino.py fwd > /dev/null 2>&1
    python scripts/pmc_dump.py $(find /tmp/pmcf_sq2 -name "*.db" | head -1) wino_fused >> $O/pmc_wino_fused_sq.txt 2>&1
    cut -c 1-130 $O/pmc_wino_fused_sq.txt | tee -a $O/progress.log
fi
if has pmc16; then      # the 16-bit ResBlock forward: conv_fwd16s / patch kernel (lockstep, counter-phase)
    for m in 0 1 2; do
        rm -rf /tmp/pmc16_a /tmp/pmc16_b
        timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pmc16_a -o p -- python scripts/probe_fwd16.py bf16 $m > /dev/null 2>&1
        timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d /tmp/pmc16_b -o p -- python scripts/probe_fwd16.py bf16 $m > /dev/null 2>&1
        echo "== fwd16_patch $m" >> $O/pmc_fwd16.txt
        for d in a b; do python scripts/pmc_dump.py $(find /tmp/pmc16_$d -name "*.db" | head -1) conv_fwd16 >> $O/pmc_fwd16.txt 2>&1; done
        python scripts/probe_fwd16.py bf16 $m 2>&1 | grep fwd16 >> $O/pmc_fwd16.txt
    done
    grep -E "fwd16 bf16|VALU_MFMA_BUSY|GRBM" $O/pmc_fwd16.txt | cut -c 1-130 | tee -a $O/progress.log
fi
if has pmcwgrad; then
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmcw_$c
        timeout 300 rocprofv3 --pmc $c -d /tmp/pmcw_$c -o p -- python scripts/probe_wino.py wgrad > $O/pmcw_$c.log 2>&1
    done
    python scripts/pmc_kernel_traffic.py $(find /tmp/pmcw_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmcw_WRITE_SIZE -name "*.db" | head -1) wino_wgrad_fused_kernel $O/pmc_wino_wgrad_fused.json | cut -c 1-300 | tee -a $O/progress.log
    rm -rf /tmp/pmcw_sq
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT -d /tmp/pmcw_sq -o p -- python scripts/probe_wino.py wgrad > /dev/null 2>&1
    python scripts/pmc_dump.py $(find /tmp/pmcw_sq -name "*.db" | head -1) wgrad_fused > $O/pmc_wino_wgrad_fused_sq.txt 2>&1
    cut -c 1-130 $O/pmc_wino_wgrad_fused_sq.txt | tee -a $O/progress.log
fi
if has roctx; then
    rm -rf /tmp/prof_roctx
    ACLGAN_ROCTX=1 ACLGAN_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --marker-trace -d /tmp/prof_roctx -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-launch-floor --no-other-configs > $O/prof_roctx.log 2>&1
    DB=$(find /tmp/prof_roctx -name "*.db" | head -1)
    python scripts/rocpd_schema.py $DB > $O/roctx_schema.txt 2>&1
    python scripts/rocpd_bypass.py $DB > $O/kernel_time_by_pass.txt 2>&1
    head -50 $O/kernel_time_by_pass.txt | cut -c 1-150 | tee -a $O/progress.log
fi
if [ -n "${EXTRA:-}" ]; then echo "== extra: $EXTRA" | tee -a $O/progress.log; (eval "$EXTRA") 2>&1 | tail -${EXTRA_TAIL:-40} | tee -a $O/progress.log; fi
if has prestreams; then      # round 6: does a foreign stream created first (RCCL's, a loader's) shift the lanes' hardware queues?  lanes x pre-streams
    for ps in 0 1 2; do for ln in 2 3; do
        $B --no-other-configs --no-launch-floor --steps 10 --warmup 3 --lanes $ln --pre-streams $ps > $O/bench_256_fp32_lanes${ln}_prestreams${ps}.json 2>/dev/null; summ $O/bench_256_fp32_lanes${ln}_prestreams${ps}.json | tee -a $O/progress.log
    done; done
fi
if has split; then      # round 6, review item 4: the split-bf16 (6 products) inner loop against today's fp32 loop, register-resident
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Iscripts/microbench -o /tmp/split_bf16_loop scripts/microbench/split_bf16_loop.hip > $O/split_build.log 2>&1
    timeout 300 /tmp/split_bf16_loop > $O/microbench_split_bf16_loop.txt 2>&1; cat $O/microbench_split_bf16_loop.txt | tee -a $O/progress.log
fi
if has thin; then      # round 6: the image-side layers in isolation, new kernels against ACLGAN_THININ2=0
    (timeout 300 python scripts/probe_thin.py 10 2>&1 | grep -v amdgpu) > $O/probe_thin.txt; (ACLGAN_THININ2=0 timeout 300 python scripts/probe_thin.py 10 2>&1 | grep -v amdgpu) > $O/probe_thin_round5_kernels.txt
    paste -d'|' $O/probe_thin.txt $O/probe_thin_round5_kernels.txt | cut -c 1-200 | tee -a $O/progress.log
fi
if has s2k4; then (timeout 300 python scripts/bench_s2k4.py 2>&1 | grep -v amdgpu) > $O/s2k4_direct_vs_fused_per_shape.txt; cat $O/s2k4_direct_vs_fused_per_shape.txt | tee -a $O/progress.log; fi
if has capture; then      # round 6: minimal reproduction of the multi-stream capture crash (scripts/debug/capture_lanes.hip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/capture_lanes scripts/debug/capture_lanes.hip > $O/capture_build.log 2>&1
    timeout 300 /tmp/capture_lanes > $O/capture_variants.txt 2>&1; paste - - < $O/capture_variants.txt | cut -c 1-160 | tee -a $O/progress.log
    timeout 600 /tmp/capture_lanes rings > $O/capture_rings.txt 2>&1; grep -c CRASHED $O/capture_rings.txt | tee -a $O/progress.log
    timeout 900 /tmp/capture_lanes sweep > $O/capture_sweep.txt 2>&1; grep -c CRASHED $O/capture_sweep.txt | tee -a $O/progress.log
fi
echo "== done" | tee -a $O/progress.log
