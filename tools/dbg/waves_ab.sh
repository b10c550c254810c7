#!/bin/bash
# scan time against waves per workgroup (TOKSCORE_WAVES), library's own events: tools/dbg/waves_ab.sh <out>
OUT=$1; mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for w in 4 6 8 10 12 14 16; do
  echo "waves $w" >> "$OUT"
  TOKSCORE_WAVES=$w timeout -k 10 120 python3 tools/dbg/ab_lib.py 50 2>/dev/null | tail -n 1 >> "$OUT" || exit 1
done
cat "$OUT"
