#!/bin/bash
# Automaton scan forced on the vocabularies the policy keeps off it, beside the default path (debugging aid).
set -u
mkdir -p gpurun_out
echo "== default policy" 
timeout -k 10 200 python3 tools/dbg/annealed_time.py 100 1 || exit 1
echo "== forced"
TOKSCORE_DFA_MAX_COLD_PPM=1000000 TOKSCORE_DFA_MAX_VIRTUAL_PPM=1000000 TOKSCORE_DFA_SLOTS=33554432 timeout -k 10 300 python3 tools/dbg/annealed_time.py 100 1 || exit 1
