# same-process-class A/B of one library option through the full bench step: bash tools/experiments/ab_option.sh refine_interleave
O=$1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain"
for r in 1 2 3; do
  for v in 0 1; do
    $B --option $O=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$O=$v', 'train', d['ms_per_step'], d['roofline']['frac'], 'infer', d['inference_step']['ms_per_step'], d['inference_step']['roofline']['frac'])"
  done
done
