B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain"
for r in 1 2; do
  echo -n "default: "; $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['launches'], r['launches_timed'], r['kernel_time_share'], d['inference_step']['ms_per_step'], d['inference_step']['roofline']['frac'])"
done
python bench.py --steps 8 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -c 600 gpurun_out/bench_full.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_full.json').readlines()[-1]); print(d['ms_per_step'], d['value'], d['exact_fp32']['ms_per_step'], d['exact_fp32']['roofline']['kernel_time_share'], d['roofline']['event_sampling'])"
