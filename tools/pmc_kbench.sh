#!/bin/bash
# PMC passes over the kbench binary (GPU box).  Usage: tools/pmc_kbench.sh <outdir> <filter> [reps]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/$1; F=$2; REPS=${3:-3}
mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $R/dtcwt_amd/csrc -I $R/include $R/tools/kbench/kbench.hip -o /tmp/kbench 2>/dev/null
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- /tmp/kbench "$F" $REPS > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" >> "$OUT/status.txt"
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq_b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq_c SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
cd $R
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/status.txt"
