#!/bin/bash
# timing experiments of the failure-form scan (results WRONG with any bit set): 1 no counters for cold keys, 2 no event log, 4 no global lookups
for d in 0 1 2 3 4 7; do
  echo "dbg=$d"
  TOKSCORE_FSCAN_DBG=$d timeout -k 10 120 python -u tools/dbg/fail_time.py 100 annealed fail:4 || exit 1
done
