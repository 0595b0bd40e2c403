# same-box A/B of the committed library (iodine_amd/ab/libhead.so) against the working tree, full bench step (Adam + repack included)
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain"
for r in 1 2 3; do
  for lib in iodine_amd/ab/libhead.so iodine_amd/libiodine_hip.so; do
    IODINE_HIP_LIB=$lib $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', 'train', d['ms_per_step'], 'infer', d['inference_step']['ms_per_step'])"
  done
done
