# same-box kernel-level comparison: fp32 step with and without the split-bf16 Winograd GEMM slices (side stream off: kernel times add up)
export TMPDIR=/tmp ACLGAN_SIDE_STREAM=0
O=gpurun_out/r03_x3ab; mkdir -p $O
for v in 1 0 1 0; do
  rm -rf /tmp/prof_x3_$v
  ACLGAN_WINO_X3=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_x3_$v -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor > $O/log_$v.txt 2>&1
  DB=$(find /tmp/prof_x3_$v -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB > $O/stats_x3_${v}_$RANDOM.txt 2>&1
  grep -E "ms_per_step" $O/log_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3=$v', d['ms_per_step'])"
done
python - <<'PY'
import glob, re, collections
for v in (1, 0):
    tot = collections.defaultdict(float); n = 0
    for f in glob.glob("gpurun_out/r03_x3ab/stats_x3_%d_*.txt" % v):
        n += 1
        for l in open(f):
            m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
            if m: tot[m.group(1).strip()] += float(m.group(3))
    print("== x3=%d (%d runs): ms per step (6 traced step-equivalents)" % (v, n))
    for k, t in sorted(tot.items(), key=lambda kv: -kv[1])[:24]: print("  %-70s %8.2f" % (k[:70], t / n / 6e3))
    print("  TOTAL %8.2f" % (sum(t for k, t in tot.items() if k != "TOTAL") / n / 6e3))
PY
