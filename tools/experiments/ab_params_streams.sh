B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain"
for r in 1 2; do
for o in "params_streams=0" "params_streams=6" "params_streams=12"; do
  echo -n "$o: "; $B --option $o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"
done; done
python -m pytest tests/test_gpu_train.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/pt.log 2>&1; grep -n "passed\|failed" gpurun_out/pt.log | tail -3
