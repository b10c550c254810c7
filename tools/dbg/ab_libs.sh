#!/bin/bash
# A/B of experimental builds of the library: tools/dbg/ab_libs.sh <out> <lib>... (paths relative to the repo root; "base" = the product library)
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for L in "$@"; do
  if [ "$L" = base ]; then unset TOKSCORE_LIBRARY; else export TOKSCORE_LIBRARY=$PWD/$L; fi
  timeout -k 10 120 python3 tools/dbg/ab_lib.py 100 >> "$OUT" 2>&1 || { echo "$L failed" >> "$OUT"; tail -n 5 "$OUT"; exit 1; }
done
cat "$OUT"
