#!/bin/bash
# rocprofv3 passes for the scan kernel: kernel trace + stats, then PMC counters in their
# own runs (never combined with a trace domain).  Usage: tools/pmc_passes.sh <outdir> [mb]
set -u
OUT=${1:-gpurun_out/prof}
MB=${2:-100}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/profile_run.py $MB 1 > "$OUT/warm.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/profile_run.py $MB 5 > "$OUT/trace.log" 2>&1 || exit 1
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- python3 tools/profile_run.py $MB 2 > "$OUT/pmc$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/pmc$i.log"; exit 1; }
done
echo done
