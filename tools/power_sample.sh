# sample socket power / clocks while the training bench runs (same box): is the step power-limited?
python bench.py --no-cpu-baseline --steps 200 --warmup 5 > gpurun_out/bb.log 2>&1 &
BP=$!
for i in $(seq 1 40); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.4
done
wait $BP
tail -1 gpurun_out/bb.log | cut -c1-160
