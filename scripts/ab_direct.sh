# A/B of conv_dgrad16s' direct interior writes (ACLGAN_DGRAD16S_DIRECT=0: every pixel through the padded scratch + fold), in the step, same box
mkdir -p gpurun_out/r03_direct
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_dis_update'], d['config']['ms_gen_update'])"; }
(
timeout 900 python -m pytest tests/test_gpu_ops16s.py tests/test_gpu_step16.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -3
for m in 1 0 1 0; do
echo "== bf16 b8 direct=$m"; ACLGAN_DGRAD16S_DIRECT=$m python bench.py --dtype bf16 --no-cpu-baseline --no-launch-floor --steps 6 --warmup 3 2>/dev/null | j
done
for m in 1 0; do
echo "== fp16 b32 direct=$m"; ACLGAN_DGRAD16S_DIRECT=$m python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 --warmup 2 2>/dev/null | j
done
) > gpurun_out/r03_direct/log.txt 2>&1
cat gpurun_out/r03_direct/log.txt
