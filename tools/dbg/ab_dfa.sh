#!/bin/bash
# A/B of experimental builds of the automaton scan's loop: tools/dbg/ab_dfa.sh <out> <chains> <lib>... ("base" = the product library)
OUT=$1; shift
CH=$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for L in "$@"; do
  if [ "$L" = base ]; then unset TOKSCORE_LIBRARY; else export TOKSCORE_LIBRARY=$PWD/$L; fi
  echo "== $L" >> "$OUT"
  timeout -k 10 150 python3 tools/dbg/dfa_time2.py $CH >> "$OUT" 2>&1 || { echo "$L failed" >> "$OUT"; tail -n 5 "$OUT"; exit 1; }
done
cat "$OUT"
