#!/bin/bash
# One steady-state cfg2 (multi-dSprites 64x64, K=6, T=5, B=32) training step and reconstruct step as kernel timelines
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl /tmp/tli
rocprofv3 --kernel-trace -d /tmp/tl -- python $R/bench.py --config dsprites --steps 10 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace -d /tmp/tli -- python $R/bench.py --config dsprites --mode infer --steps 10 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain "$@" > /dev/null 2>&1
cd $R
python tools/step_timeline.py $(ls /tmp/tl/*/*.db | head -1) adam_multi_kernel --list > gpurun_out/timeline_ds_train.txt
python tools/step_timeline.py $(ls /tmp/tli/*/*.db | head -1) final_out_kernel --list > gpurun_out/timeline_ds_infer.txt
grep -v "^ " gpurun_out/timeline_ds_train.txt | head -70
