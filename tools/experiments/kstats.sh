# per-kernel averages of a short training bench under rocprofv3 (top 12): bash tools/experiments/kstats.sh [bench args]
mkdir -p gpurun_out/p1; rm -rf gpurun_out/p1/*
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p1 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-fp32 "$@" > /dev/null 2>&1
cd $R
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/p1/**/p_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r["Name"][:78], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
