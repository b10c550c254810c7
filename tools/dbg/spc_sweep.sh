#!/bin/bash
# batched moves against the granularity of the occurrence index (streams per chunk); debugging aid
set -u
for SPC in 16 8 4 2 1; do
  echo "== TOKSCORE_SPC=$SPC"
  TOKSCORE_SPC=$SPC TOKSCORE_TRACE_HOST=0 timeout -k 10 200 python3 tools/bench_score_toggles.py 100 64 20 || exit 1
done
