# quick correctness (golden-based subsets) + bench numbers after a change: bash tools/experiments/quick_check.sh
python -m pytest tests/test_gpu_reconstruct.py tests/test_gpu_train.py tests/test_gpu_refl0.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/pt.log 2>&1; grep -n "passed\|failed" gpurun_out/pt.log | tail -3
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain"
for r in 1 2; do
  $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('train', d['ms_per_step'], r['frac'], 'infer', d['inference_step']['ms_per_step'], d['inference_step']['roofline']['frac'])"
done
