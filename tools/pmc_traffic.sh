#!/bin/bash
# HBM traffic per launch of the dominant kernels (forward / data-gradient / weight-gradient 3x3 conv C->C) of the default
# bench step, from rocprofv3 PMC passes: FETCH_SIZE and WRITE_SIZE in SEPARATE runs (TCC slots, MI355X_MICROARCH.md), kernel
# trace only.  Run on the GPU box from the repo root:   bash tools/pmc_traffic.sh [extra bench args]
# Run as:   bash tools/pmc_traffic.sh <tag> [extra bench args]        (tag = r03 ...)
# Writes gpurun_out/pmc_<tag>/{FETCH_SIZE,WRITE_SIZE,SQ_VALU_MFMA_BUSY_CYCLES}/..., gpurun_out/<tag>_pmc.json (tools/pmc_to_json.py)
# and gpurun_out/<tag>_kernel_traffic.md (tools/pmc_all_kernels.py); copy both to profiles/.
set -u
TAG=${1:-r05}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
# passes over the exact-fp32 path (bash tools/pmc_traffic.sh <tag> --conv-precision 0) are kept beside the default ones:
# gpurun_out/pmc_<tag>_strict/, <tag>_pmc_strict.json, <tag>_kernel_traffic_strict.md
SFX=""
case " $* " in *" --conv-precision 0 "*) SFX="_strict";; esac
OUT=$REPO/gpurun_out/pmc_$TAG$SFX
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/$tag
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- \
      python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain "$@" > $OUT/$tag.log 2>&1 \
      || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; }
done
cd $REPO
python tools/pmc_to_json.py $OUT $TAG "$@"
python tools/pmc_all_kernels.py $OUT > $REPO/gpurun_out/${TAG}_kernel_traffic$SFX.md
