#!/bin/bash
# PMC passes of rocprofv3 over an arbitrary command (GPU box).
# Usage: tools/pmc_cmd.sh <outdir under repo> <command ...>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- $CMD > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" >> "$OUT/status.txt"
}
CMD="$*"
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq_b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq_c SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
run tcc_fetch FETCH_SIZE GRBM_GUI_ACTIVE
run tcc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $R
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/status.txt"
