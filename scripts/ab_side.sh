mkdir -p gpurun_out/r03_side
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_dis_update'], d['config']['ms_gen_update'])"; }
(
for dt in fp32 bf16; do
  echo "== $dt side=1"; python bench.py --dtype $dt --no-cpu-baseline --no-launch-floor --steps 8 2>/dev/null | j
  echo "== $dt side=0"; ACLGAN_SIDE_STREAM=0 python bench.py --dtype $dt --no-cpu-baseline --no-launch-floor --steps 8 2>/dev/null | j
done
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_determinism.py tests/test_gpu_graph.py tests/test_gpu_ddp.py tests/test_gpu_step16.py -m gpu -q -x 2>&1 | tail -15
) > gpurun_out/r03_side/log.txt 2>&1
cat gpurun_out/r03_side/log.txt | cut -c 1-300
