#!/bin/bash
# Run on the GPU box (inside gpurun): PMC passes of rocprofv3 over a short bench run.
# Each --pmc set is its own run with --kernel-trace only (MI355X_MICROARCH.md: TCC has 4
# slots, FETCH_SIZE takes 3, WRITE_SIZE 2; SQ has 8).  Usage: tools/pmc.sh <outdir> [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/${1:-gpurun_out/pmc}
shift || true
ARGS=${*:---steps 5 --warmup 2 --no-cpu-baseline}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout 180 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python $R/bench.py $ARGS > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" >> "$OUT/status.txt"
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq_b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq_c SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F32 SQ_THREAD_CYCLES_VALU
run tcc_fetch FETCH_SIZE GRBM_GUI_ACTIVE
run tcc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# (a TCP_* pass aborted inside rocprofv3 on this image and then hung: not collected)
cd $R
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/status.txt"
