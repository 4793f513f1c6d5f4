#!/bin/bash
# Round-3 evidence run on one MI355X (through gpurun).  Stages selected with STAGES="tests bench trace traffic bench16 trace16 ..." (default: all).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'STAGES="tests bench" bash scripts/evidence_r03.sh'
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03${TAG:+_$TAG}
mkdir -p $O
export TMPDIR=/tmp
STAGES=${STAGES:-"tests smoke bench trace traffic bench16 trace16"}
has() { [[ " $STAGES " == *" $1 "* ]]; }
B="python bench.py --no-cpu-baseline"
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); c = d["config"]; r = d["roofline"]
    print(sys.argv[1], d["value"], "img/s", d["ms_per_step"], "ms", "dis", c["ms_dis_update"], "gen", c["ms_gen_update"], "launches", c.get("kernel_launches_per_step"),
          "floor", c.get("launch_bound_floor_ms_per_step"), "frac", r["frac"], "alg_frac", r.get("algorithmic_frac"), "kernel", r["kernel"]["frac"], r["kernel"]["ms"],
          "traffic", r.get("traffic"), "alg_bytes", r.get("algorithmic_bytes"), "cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
echo "== stages: $STAGES" | tee $O/progress.log
if has tests; then
    (timeout ${TEST_TIMEOUT:-1500} python -m pytest ${PYTEST_PATHS:-tests} -m gpu -q -s ${PYTEST_X--x} ${PYTEST_ARGS:-} 2>&1 | grep -vE "^\s*$" | cut -c 1-900) > $O/tests_full.log
    grep -E "passed|failed|error" $O/tests_full.log | tail -3 | tee -a $O/progress.log
    grep -E "worst|passed|failed|rel errors|shard equivalence|chained|under the floor|Error|assert" $O/tests_full.log | cut -c 1-700 > $O/tests_summary.log
fi
if has smoke; then (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) | tee $O/smoke.log | tee -a $O/progress.log; fi
if has bench; then
    timeout 600 python bench.py > $O/bench_256_fp32.json 2> $O/bench_256_fp32.err; summ $O/bench_256_fp32.json | tee -a $O/progress.log
fi
if has bench512; then $B --config configs/glasses_removal.yaml > $O/bench_512_fp32.json 2>/dev/null; summ $O/bench_512_fp32.json | tee -a $O/progress.log; fi
if has bench16; then
    $B --config configs/selfie2anime.yaml > $O/bench_256_bf16.json 2>/dev/null; summ $O/bench_256_bf16.json | tee -a $O/progress.log
    $B --dtype fp16 > $O/bench_256_fp16_b32.json 2>/dev/null; summ $O/bench_256_fp16_b32.json | tee -a $O/progress.log
fi
if has benchdet; then
    $B --deterministic > $O/bench_256_fp32_deterministic.json 2>/dev/null; summ $O/bench_256_fp32_deterministic.json | tee -a $O/progress.log
    $B --dtype bf16 --deterministic > $O/bench_256_bf16_deterministic.json 2>/dev/null; summ $O/bench_256_bf16_deterministic.json | tee -a $O/progress.log
fi
trace() {   # tag, bench args
    rm -rf /tmp/prof_$1
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor $2 > $O/prof_$1.log 2>&1
    DB=$(find /tmp/prof_$1 -name "*.db" | head -1)
    python scripts/rocpd_stats.py $DB > $O/kernel_stats_$1.txt 2>&1
    python scripts/rocpd_bygrid.py $DB 6 "" 80 > $O/by_grid_$1.txt 2>&1
    head -12 $O/kernel_stats_$1.txt | cut -c 1-140 | tee -a $O/progress.log; tail -1 $O/kernel_stats_$1.txt | tee -a $O/progress.log
}
if has trace; then trace 256_fp32 ""; fi
if has trace16; then trace 256_bf16 "--dtype bf16"; fi
if has trace512; then trace 512_fp32 "--size 512 --batch 4"; fi
traffic() {   # dtype size batch
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${c}_$1
        timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_${c}_$1 -o p -- python scripts/probe_step.py $1 $2 $3 2 > $O/pmc_${c}_$1.log 2>&1
    done
    F=$(find /tmp/pmc_FETCH_SIZE_$1 -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE_$1 -name "*.db" | head -1)
    python scripts/step_traffic.py $F $W 2 ${1}_${2}_b${3} $O/step_traffic.json $O/step_traffic_${1}_${2}_b${3}.txt | head -14 | cut -c 1-120 | tee -a $O/progress.log
}
if has traffic; then traffic fp32 256 8; fi
if has traffic16; then traffic bf16 256 8; fi
if has pmcwino; then
    for w in fwd wgrad; do
        rm -rf /tmp/pmc_sq_$w
        timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA -d /tmp/pmc_sq_$w -o p -- python scripts/probe_wino.py $w > /dev/null 2>&1
        DB=$(find /tmp/pmc_sq_$w -name "*.db" | head -1)
        echo "## $w" >> $O/pmc_winograd_mfma.txt
        python scripts/pmc_dump.py $DB "" | grep -v "at::native\|fillBuffer" >> $O/pmc_winograd_mfma.txt 2>&1
    done
    cut -c 1-150 $O/pmc_winograd_mfma.txt | tee -a $O/progress.log
fi
if [ -n "${EXTRA:-}" ]; then echo "== extra: $EXTRA" | tee -a $O/progress.log; (eval "$EXTRA") 2>&1 | tail -${EXTRA_TAIL:-40} | tee -a $O/progress.log; fi
echo "== done" | tee -a $O/progress.log
