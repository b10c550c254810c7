#!/bin/bash
# Memory-path counters of the scan kernel (texture addresser / L1 / L2 request side), separate
# --pmc passes.  Usage (on the GPU box): bash tools/pmc_mem.sh <tag>   -> gpurun_out/<tag>/mem*/
set -u
TAG=${1:-mem}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for CTRS in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
            "TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/mem$i" -- python3 bench.py --no-cpu --steps 3 --warmup 1 > "$OUT/mem$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/mem$i.log"; }
done
echo done
