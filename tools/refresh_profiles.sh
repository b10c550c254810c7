#!/bin/bash
# One GPU call that refreshes everything the judge reads: the bench line, the rocprofv3
# kernel stats of the same command, and the HBM traffic counters (separate --pmc passes).
# Usage (on the GPU box): bash tools/refresh_profiles.sh <tag>     -> gpurun_out/<tag>/
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -n 20 "$OUT/bench.err"; exit 1; }
tail -n 1 "$OUT/bench.json"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -n 5 "$OUT/trace.log"; exit 1; }
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- python3 bench.py --no-cpu --steps 3 --warmup 1 > "$OUT/pmc$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/pmc$i.log"; [ $i -le 3 ] && exit 1; }
done
echo done
