# A/B of the wave-specialised conv_glds16 kernels (ACLGAN_GLDS_SPEC=0: unified kernels) inside the step, same box, back to back
mkdir -p gpurun_out/r03_spec
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_dis_update'], d['config']['ms_gen_update'])"; }
(
timeout 900 python -m pytest tests/test_gpu_ops16s.py -q -x 2>&1 | tail -5
for sp in 1 0; do for t in 1 4; do
echo "== probe spec=$sp tile=$t"; ACLGAN_GLDS_SPEC=$sp ACLGAN_GLDS_TILE=$t python scripts/probe16s.py 2>&1 | tail -4
done; done
for sp in 1 0; do for t in 1 4; do
echo "== bf16 b8 spec=$sp tile=$t"; ACLGAN_GLDS_SPEC=$sp ACLGAN_GLDS_TILE=$t python bench.py --dtype bf16 --no-cpu-baseline --no-launch-floor --steps 6 --warmup 3 2>/dev/null | j
done; done
for sp in 1 0; do
echo "== fp16 b32 spec=$sp"; ACLGAN_GLDS_SPEC=$sp python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 --warmup 2 2>/dev/null | j
done
echo "== fp16 b32 spec=1 tile=4"; ACLGAN_GLDS_TILE=4 python bench.py --dtype fp16 --no-cpu-baseline --no-launch-floor --steps 4 --warmup 2 2>/dev/null | j
) > gpurun_out/r03_spec/log.txt 2>&1
cat gpurun_out/r03_spec/log.txt
