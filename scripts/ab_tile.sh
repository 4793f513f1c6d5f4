# A/B of the conv_glds16 tile choice inside the step (same box, back to back): ACLGAN_GLDS_TILE=1 = 128-row tiles everywhere (the previous build)
mkdir -p gpurun_out/r03_tile
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_dis_update'], d['config']['ms_gen_update'])"; }
(
timeout 900 python -m pytest tests/test_gpu_ops16s.py -q -x 2>&1 | tail -5
for t in 0 1 2; do
echo "== bf16 b8 tile=$t"; ACLGAN_GLDS_TILE=$t python bench.py --dtype bf16 --no-cpu-baseline --no-launch-floor --steps 6 --warmup 3 2>/dev/null | j
done
for t in 0 1 2; do
echo "== fp16 b32 tile=$t"; ACLGAN_GLDS_TILE=$t python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 --warmup 2 2>/dev/null | j
done
for t in 1 2 3; do echo "== probe tile=$t"; ACLGAN_GLDS_TILE=$t python scripts/probe16s.py 2>&1 | tail -5; done
) > gpurun_out/r03_tile/log.txt 2>&1
cat gpurun_out/r03_tile/log.txt
