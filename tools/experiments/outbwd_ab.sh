for v in "1 768" "1 512" "0 1024" "0 768" "1 768"; do set -- $v
IODINE_OUTBWD_PF=$1 IODINE_OUTBWD_CAP=$2 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('pf cap $1 $2', d['ms_per_step'], d['kernels']['dec_out_bwd']['ms_avg'])"
done
