R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/tl; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tl -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain "$@" > /dev/null 2>&1
cd $R
python tools/step_timeline.py $(ls /tmp/tl/*/*.db | head -1) adam_multi_kernel --list > gpurun_out/timeline_train.txt
grep -v "^ " gpurun_out/timeline_train.txt | head -80
