R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/tl; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tl -- python $R/bench.py --mode infer --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain "$@" > /dev/null 2>&1
cd $R
python tools/step_timeline.py $(ls /tmp/tl/*/*.db | head -1) x_to_nhwc4_kernel > gpurun_out/timeline_infer.txt
head -45 gpurun_out/timeline_infer.txt
