#!/bin/bash
# PMC passes over tools/kbench/fwd12_bench (GPU box).  Usage: tools/pmc_fwd12.sh <outdir under the repo> [N] [reps]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/$1; N=${2:-4096}; REPS=${3:-6}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- $R/tools/kbench/fwd12_bench $N $REPS 1 only > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" >> "$OUT/status.txt"
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq_b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq_c SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/status.txt"
