# A/B of the split-bf16 Winograd GEMM slices (ACLGAN_WINO_X3=0: fp32 MFMA slices) inside the fp32 step, same box, back to back
mkdir -p gpurun_out/r03_x3
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_dis_update'], d['config']['ms_gen_update'], d['config'].get('kernel_launches_per_step'))"; }
(
timeout 1200 python -m pytest ${X3_TESTS:-tests/test_gpu_ops_wino.py tests/test_gpu_ops_misc.py tests/test_gpu_step.py} -q -x 2>&1 | tail -5
for v in ${X3_LIST:-1 0}; do
echo "== fp32 b8 x3=$v"; ACLGAN_WINO_X3=$v python bench.py --no-cpu-baseline --no-launch-floor --steps 6 --warmup 3 2>/dev/null | j
done
for e in ${X3_EXTRA:-}; do
echo "== fp32 b8 x3=1 $e"; env $e python bench.py --no-cpu-baseline --no-launch-floor --steps 6 --warmup 3 2>/dev/null | j
done
) > gpurun_out/r03_x3/log.txt 2>&1
cat gpurun_out/r03_x3/log.txt
