#!/bin/bash
# Quick counter passes over the bench command (scan kernel only is summarised by tools/pmc_quick.py).
# Usage (on the GPU box): bash tools/pmc_quick.sh <tag>
set -u
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- python3 bench.py --no-cpu --steps 3 --warmup 2 > "$OUT/pmc$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/pmc$i.log"; }
done
python3 tools/pmc_quick.py "$TAG"
