#!/bin/bash
# Time the ablation libraries of tools/build_ablations.sh on the GPU box (rocprofv3 kernel trace of tools/data_dependence.py).
MODE=${1:-2}
cd /tmp && export TMPDIR=/tmp
for a in BASE NOINLOAD NOWLOAD NOLDSREAD NOMFMA NOSTORE; do
  rm -rf /tmp/abl_$a
  IODINE_HIP_LIB=/root/repo/iodine_amd/ab/lib_$a.so rocprofv3 --kernel-trace -d /tmp/abl_$a -o t -- python /root/repo/tools/data_dependence.py $MODE > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
f = glob.glob('/tmp/abl_$a/**/*.db', recursive=True)[0]
c = sqlite3.connect(f)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
s = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_info_kernel_symbol%'")][0]
rows = c.execute(f"select k.kernel_name, d.end-d.start from {t} d join {s} k on d.kernel_id=k.id order by d.start").fetchall()
d = [r[1]/1e3 for r in rows if 'conv3x3_tile' in r[0]]
m = lambda v: sorted(v)[len(v)//2]
print(f'%-10s random {m(d[4:8]):6.0f} us   zeros {m(d[12:16]):6.0f} us   const {m(d[20:24]):6.0f} us' % '$a')
PY
done
