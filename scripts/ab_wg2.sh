# A/B of the pixel-major LDS-DMA weight gradient at every size (default) against the register-transposing kernel, in the step, same box
mkdir -p gpurun_out/r03_wg2
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_dis_update'], d['config']['ms_gen_update'])"; }
(
timeout 900 python -m pytest tests/test_gpu_step16.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -3
for m in 64 100000000 64 100000000; do
echo "== bf16 b8 minpix=$m"; ACLGAN_WGRAD16S_MINPIX=$m python bench.py --dtype bf16 --no-cpu-baseline --no-launch-floor --steps 6 --warmup 3 2>/dev/null | j
done
for m in 64 65536; do
echo "== fp16 b32 minpix=$m"; ACLGAN_WGRAD16S_MINPIX=$m python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 --warmup 2 2>/dev/null | j
done
echo "== bf16 b8 side=0"; ACLGAN_SIDE_STREAM=0 python bench.py --dtype bf16 --no-cpu-baseline --no-launch-floor --steps 6 --warmup 3 2>/dev/null | j
) > gpurun_out/r03_wg2/log.txt 2>&1
cat gpurun_out/r03_wg2/log.txt
