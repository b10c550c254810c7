#!/bin/bash
# per-kernel times of batched moves at two index granularities; debugging aid
set -u
export TMPDIR=/tmp
for SPC in "$@"; do
  export TOKSCORE_SPC=$SPC
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/spc_$SPC -- python3 tools/bench_score_toggles.py 100 64 20 > gpurun_out/spc_$SPC.log 2>&1 || { echo "$SPC failed"; tail -n 5 gpurun_out/spc_$SPC.log; exit 1; }
  echo "== SPC=$SPC"; tail -n 1 gpurun_out/spc_$SPC.log | cut -c1-120
  python3 - $SPC <<'PY'
import csv,sys,glob
for p in glob.glob('gpurun_out/spc_'+sys.argv[1]+'/*/*_kernel_stats.csv'):
    rows=list(csv.reader(open(p)))[1:]
    for r in rows:
        n=r[0].split('(')[0].split('::')[-1]
        print('%-40s calls %5s avg %9.1f us total %9.1f us'%(n[:40], r[1], float(r[3])/1e3, float(r[2])/1e3))
PY
done
