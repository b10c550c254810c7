#!/bin/bash
# as ab_libs.sh at another shard size: tools/dbg/ab_libs_mb.sh <out> <mb> <lib>...
OUT=$1; MB=$2; shift; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for L in "$@"; do
  if [ "$L" = base ]; then unset TOKSCORE_LIBRARY; else export TOKSCORE_LIBRARY=$PWD/$L; fi
  timeout -k 10 300 python3 tools/dbg/ab_lib.py 30 $MB >> "$OUT" 2>&1 || { echo "$L failed" >> "$OUT"; tail -n 5 "$OUT"; exit 1; }
done
cat "$OUT"
