mkdir -p gpurun_out/r03_wg
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_dis_update'], d['config']['ms_gen_update'])"; }
(
for side in 1 0; do for nb in 0 2; do
echo "== fp16 b32 side=$side nbuf=$nb"; ACLGAN_SIDE_STREAM=$side ACLGAN_GLDS_NBUF=$nb python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 --warmup 2 2>/dev/null | j
done; done
echo "== fp16 b32 side=1 nbuf=2 wgrad16s off"; ACLGAN_NOWGRAD16S=1 ACLGAN_GLDS_NBUF=2 python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 --warmup 2 2>/dev/null | j
echo "== fp16 b32 side=1 nbuf=0 co16=0"; ACLGAN_CO16=0 python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 --warmup 2 2>/dev/null | j
) > gpurun_out/r03_wg/log4.txt 2>&1
cat gpurun_out/r03_wg/log4.txt
