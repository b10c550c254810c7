#!/bin/bash
# per-kernel times of experimental builds that leave parts of the count pass out (csrc/Makefile exp_dfa DFA_EXPS="HIST1 ..."); debugging aid
set -u
export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" = base ]; then unset TOKSCORE_LIBRARY; else export TOKSCORE_LIBRARY=$PWD/tools/experiments/libtokdfa_$L.so; fi
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kh_$L -- python3 tools/dbg/dfa_time2.py 2 > gpurun_out/kh_$L.log 2>&1 || { echo "$L failed"; tail -n 5 gpurun_out/kh_$L.log; exit 1; }
  echo "== $L"; grep -h "dfa_hist\|dfa_reduce\|dfa_scan" gpurun_out/kh_$L/*/*_kernel_stats.csv | awk -F'","' '{n=split($1,a,"::"); print substr(a[n],1,24), "avg", $4, "min", $6}'
done
