#!/bin/bash
# Everything the round's measured record is made of, in one GPU-box call (run from the repo root through gpurun):
#   IODINE_COMMIT=$(git rev-parse HEAD) gpurun -- "IODINE_COMMIT=$IODINE_COMMIT bash tools/round_profiles.sh r05"
#                                         -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
# (the GPU box has no .git: the commit recorded in <tag>_pmc.json / roofline.traffic_source comes from IODINE_COMMIT)
TAG=${1:-r05}
export IODINE_COMMIT=${IODINE_COMMIT:-unknown}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
# PMC passes FIRST: bench.py quotes roofline.traffic / roofline_hbm.traffic from profiles/<tag>_pmc.json only while its source digest
# matches the tree, so the record of THIS tree has to be in place before the bench lines are taken
bash tools/pmc_traffic.sh $TAG > $OUT/${TAG}_pmc.log 2>&1
cp $OUT/${TAG}_pmc.json $REPO/profiles/${TAG}_pmc.json
# ... and of the exact-fp32 path (the exact_fp32 sub-line / --conv-precision 0 quote their traffic from it)
bash tools/pmc_traffic.sh $TAG --conv-precision 0 > $OUT/${TAG}_pmc_strict.log 2>&1
cp $OUT/${TAG}_pmc_strict.json $REPO/profiles/${TAG}_pmc_strict.json
# ... and of BASELINE configs[1] (cfg2: multi-dSprites 64 x 64, 32 channels): configs.cfg2.roofline.traffic of the default line and the
# --config dsprites lines quote it; plus one SQ pass that splits the wave cycles of its kernels (parked / issue-stalled / issuing)
bash tools/pmc_traffic.sh ${TAG}ds --config dsprites > $OUT/${TAG}_pmc_ds.log 2>&1
cp $OUT/${TAG}ds_pmc.json $REPO/profiles/${TAG}_pmc_dsprites.json
cp $OUT/${TAG}ds_kernel_traffic.md $OUT/${TAG}_kernel_traffic_dsprites.md
(cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/${TAG}_sq_ds && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/${TAG}_sq_ds -o p -- python $REPO/bench.py --config dsprites --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain > $OUT/${TAG}_sq_ds.log 2>&1)
python tools/sq_split.py $OUT/${TAG}_sq_ds > $OUT/${TAG}_dsprites_wave_cycle_split.md 2>&1
rm -rf $OUT/${TAG}_sq_ds
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
python bench.py --mode infer --no-exact-fp32 --no-sustain > $OUT/${TAG}_bench_infer.json 2> /dev/null
for g in 0 1; do
  python bench.py --config dsprites --graph $g --steps 30 --no-exact-fp32 --no-cpu-baseline --no-sustain > $OUT/${TAG}_bench_dsprites_train_graph$g.json 2> /dev/null
  python bench.py --config dsprites --mode infer --graph $g --steps 30 --no-cpu-baseline --no-sustain > $OUT/${TAG}_bench_dsprites_infer_graph$g.json 2> /dev/null
done
cd /tmp && export TMPDIR=/tmp
for mode in train infer; do
  rm -rf $OUT/${TAG}_prof_$mode
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_$mode -- python $REPO/bench.py --mode $mode --steps 10 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain > $OUT/${TAG}_prof_$mode.log 2>&1
  python $REPO/tools/rocpd_summary.py $(ls $OUT/${TAG}_prof_$mode/*/*.db | head -1) > $OUT/${TAG}_${mode}_kernel_stats.md
done
python $REPO/tools/step_timeline.py $(ls $OUT/${TAG}_prof_train/*/*.db | head -1) > $OUT/${TAG}_step_timeline.md 2>&1
# the strict path (conv_precision 0: exact fp32 MFMA everywhere) as the headline line + its kernel trace
python $REPO/bench.py --conv-precision 0 --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --no-sustain > $OUT/${TAG}_bench_exact_fp32.json 2> /dev/null
rm -rf $OUT/${TAG}_prof_fp32
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_fp32 -- python $REPO/bench.py --conv-precision 0 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs --no-sustain > $OUT/${TAG}_prof_fp32.log 2>&1
python $REPO/tools/rocpd_summary.py $(ls $OUT/${TAG}_prof_fp32/*/*.db | head -1) > $OUT/${TAG}_exact_fp32_train_kernel_stats.md
# the per-GPU shard of BASELINE configs[4] (CLEVR-full shapes: K = 11, T = 7, 8 images per GPU)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_cfg5 -- python $REPO/bench.py --slots 11 --iters 7 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain > $OUT/${TAG}_prof_cfg5.log 2>&1
python $REPO/tools/rocpd_summary.py $(ls $OUT/${TAG}_prof_cfg5/*/*.db | head -1) > $OUT/${TAG}_cfg5shard_train_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_ds -- python $REPO/bench.py --config dsprites --steps 10 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain > $OUT/${TAG}_prof_ds.log 2>&1
python $REPO/tools/rocpd_summary.py $(ls $OUT/${TAG}_prof_ds/*/*.db | head -1) > $OUT/${TAG}_dsprites_train_kernel_stats.md
# the reference's default decoder kernel size on the generic path (CLEVR shapes with DEC.KERNEL_SIZE 5, batch 4: configs.default_dec_kernel5)
rm -rf $OUT/${TAG}_prof_k5
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_k5 -- python $REPO/tools/experiments/k5_prof.py > $OUT/${TAG}_prof_k5.log 2>&1
python $REPO/tools/rocpd_summary.py $(ls $OUT/${TAG}_prof_k5/*/*.db | head -1) > $OUT/${TAG}_dec_kernel5_train_kernel_stats.md
# the reference's configs/test.yaml architecture (KERNEL_SIZE 5 in both stacks, batch 32: configs.test_yaml_arch) - generic decoder + kernels_gens2.hip
rm -rf $OUT/${TAG}_prof_ty
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_ty -- python $REPO/tools/experiments/testyaml_prof.py 32 > $OUT/${TAG}_prof_ty.log 2>&1
python $REPO/tools/rocpd_summary.py $(ls $OUT/${TAG}_prof_ty/*/*.db | head -1) > $OUT/${TAG}_testyaml_train_kernel_stats.md
# matrix-pipe utilisation / clock of every kernel of the strict path
rm -rf $OUT/${TAG}_pmc_strict
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_strict -o p -- python $REPO/bench.py --conv-precision 0 --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain > $OUT/${TAG}_pmc_strict.log 2>&1
python $REPO/tools/mfma_util_all.py $OUT/${TAG}_pmc_strict > $OUT/${TAG}_strict_mfma_util.md
rm -rf $OUT/${TAG}_pmc_strict
cd $REPO
ls $OUT | grep $TAG
