#!/bin/bash
# kernel trace of estimatereg (eager launches so every kernel is a row): tools/prof_reg.sh <outdir>
out=${1:-gpurun_out/prof_reg}
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
DTCWT_HIP_REG_GRAPH=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/trace -o reg --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_registration.py > $GRAFT_REPO_ROOT/$out/log.txt 2>&1
f=$(ls $GRAFT_REPO_ROOT/$out/trace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -25 "$f" > $GRAFT_REPO_ROOT/$out/kernel_stats.txt
