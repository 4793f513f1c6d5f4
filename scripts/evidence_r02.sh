#!/bin/bash
# Round-2 evidence run on one MI355X (through gpurun): parity tests, smoke, the bench lines of every BASELINE configuration that fits
# one GPU, rocprofv3 kernel traces and the memory-side PMC passes of the Winograd pipeline.  Everything lands in gpurun_out/final/.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash scripts/evidence_r02.sh'
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline"

echo "== tests" | tee $O/progress.log
if [ -z "${SKIP_TESTS:-}" ]; then
(timeout 1500 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -vE "^\s*$" | cut -c 1-600) > $O/tests_full.log
grep -E "passed|failed" $O/tests_full.log | tail -2 | tee -a $O/progress.log
grep -E "worst|passed|failed" $O/tests_full.log > $O/tests_summary.log
fi

echo "== smoke" | tee -a $O/progress.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) | tee $O/smoke.log

echo "== bench" | tee -a $O/progress.log
timeout 600 python bench.py > $O/bench_256_fp32.json 2> $O/bench_256_fp32.err
$B --size 512 --batch 4 > $O/bench_512_fp32.json 2>/dev/null
$B --dtype bf16 > $O/bench_256_bf16.json 2>/dev/null
$B --dtype fp16 > $O/bench_256_fp16_b32.json 2>/dev/null
$B --dtype fp16 --batch 8 > $O/bench_256_fp16_b8.json 2>/dev/null
$B --deterministic > $O/bench_256_fp32_deterministic.json 2>/dev/null
$B --dtype bf16 --deterministic > $O/bench_256_bf16_deterministic.json 2>/dev/null
for f in $O/bench_*.json; do python - "$f" <<'EOF'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); c = d["config"]
    print(sys.argv[1], d["value"], d["ms_per_step"], c["ms_dis_update"], c["ms_gen_update"], c.get("launch_bound_floor_ms_per_step"), d["roofline"]["frac"],
          d["roofline"]["executed_frac"], d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
EOF
done | tee -a $O/progress.log

echo "== kernel traces" | tee -a $O/progress.log
for cfg in "fp32:" "bf16:--dtype bf16" "fp32_512:--size 512 --batch 4"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    rm -rf /tmp/prof_$tag
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor $args > $O/prof_$tag.log 2>&1
    DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
    python scripts/rocpd_stats.py $DB > $O/kernel_stats_$tag.txt 2>&1
    python scripts/rocpd_bygrid.py $DB 6 "" 70 > $O/by_grid_$tag.txt 2>&1
    head -8 $O/kernel_stats_$tag.txt | cut -c 1-140 | tee -a $O/progress.log
done

echo "== PMC (Winograd pipeline: memory-side bytes, MFMA pipe)" | tee -a $O/progress.log
bash scripts/evidence_r02_pmc.sh | tee -a $O/progress.log
echo "== done" | tee -a $O/progress.log
