#!/bin/bash
# instruction-count PMC pass of the bench with a given library: tools/pmc_valu.sh <outdir> <lib.so>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/$1; LIB=$R/$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DTCWT_HIP_LIBRARY=$LIB
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/sq" -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > "$OUT/sq.log" 2>&1
cd $R
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
