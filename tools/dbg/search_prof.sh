#!/bin/bash
# per-kernel times of batched moves by the one-pass search; debugging aid.  usage: search_prof.sh <tag> [mb K calls]
set -u
export TMPDIR=/tmp
TAG=$1; shift
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_$TAG -- python3 tools/dbg/search_ab.py ${1:-100} ${2:-64} ${3:-10} 1 > gpurun_out/sp_$TAG.log 2>&1 || { echo "failed"; tail -n 5 gpurun_out/sp_$TAG.log; exit 1; }
grep "^search" gpurun_out/sp_$TAG.log
python3 - $TAG <<'PY'
import csv,sys,glob
for p in glob.glob('gpurun_out/sp_'+sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in list(csv.reader(open(p)))[1:]:
        if int(r[1]) >= 10:
            n=r[0]; i=n.find('_kernel'); n=n[max(0,i-34):i+7] if i>=0 else n[:40]
            print('%-44s calls %4s avg %8.1f us min %8.1f'%(n, r[1], float(r[3])/1e3, float(r[5])/1e3))
PY
