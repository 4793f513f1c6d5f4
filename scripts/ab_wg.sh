mkdir -p gpurun_out/r03_wg
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_dis_update'], d['config']['ms_gen_update'], d['config']['kernel_launches_per_step'], d['roofline']['kernel']['ms'], d['roofline']['kernel']['frac'])"; }
(
echo "== ops tests"; timeout 600 python -m pytest tests/test_gpu_ops16s.py -m gpu -q 2>&1 | tail -3
echo "== fp32 ucache on"; python bench.py --no-cpu-baseline --no-launch-floor --steps 8 2>/dev/null | j
echo "== fp32 ucache off"; ACLGAN_NOUCACHE=1 python bench.py --no-cpu-baseline --no-launch-floor --steps 8 2>/dev/null | j
echo "== bf16"; python bench.py --dtype bf16 --no-cpu-baseline --no-launch-floor --steps 8 2>/dev/null | j
echo "== fp16 b32"; python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 2>/dev/null | j
echo "== fp16 b32 old wgrad"; ACLGAN_NOWGRAD16S=1 python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 2>/dev/null | j
echo "== step tests"; timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize.py::test_forward_and_losses_256_b8 tests/test_gpu_determinism.py tests/test_gpu_step16.py -m gpu -q -x 2>&1 | tail -4
) > gpurun_out/r03_wg/log3.txt 2>&1
cat gpurun_out/r03_wg/log3.txt | cut -c 1-300
