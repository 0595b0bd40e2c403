for v in base sc1 nt; do
  rm -rf gpurun_out/p1; mkdir -p gpurun_out/p1; R=$(pwd)
  (cd /tmp && TMPDIR=/tmp IODINE_HIP_LIB=$R/iodine_amd/ab/libws_$v.so rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p1 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-fp32 > /dev/null 2>&1)
  echo "== $v"; python tools/experiments/gaps.py gpurun_out/p1 | head -5
done
