#!/bin/bash
# which part of dfa_hist_kernel takes the time (debugging; results of the variants are wrong)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in NONE NOARRHIST NOMISSHIST NOFIX; do
  export TOKSCORE_DBG_$v=1
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kh_$v -- python3 tools/dbg/dfa_time2.py 2 > gpurun_out/kh_$v.log 2>&1
  echo "== $v"; grep "dfa_hist" gpurun_out/kh_$v/*/*kernel_stats.csv | sed "s/.*namespace)::\(dfa_[a-z_]*\).*)\",/\1 /"
  unset TOKSCORE_DBG_$v
done
