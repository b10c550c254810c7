#!/bin/bash
# as fail_dbg.sh, on config 5's corpus (CJK-heavy, V = 32 k) at 400 MB
for d in 0 1 2 4 7; do
  echo "dbg=$d"
  TOKSCORE_FSCAN_DBG=$d timeout -k 10 200 python -u tools/dbg/fail_time.py 400 cjk fail:4 || exit 1
done
TOKSCORE_FSCAN_DBG=0 timeout -k 10 200 python -u tools/dbg/fail_time.py 400 cjk fail:2 fail:1 tile
