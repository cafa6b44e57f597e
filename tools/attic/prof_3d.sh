#!/bin/bash
# kernel trace of the C4 forward / inverse loops: tools/prof_3d.sh <outdir>
out=${1:-gpurun_out/prof_3d}
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
for w in fwd inv; do
  REPS=20 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/$w -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof3d_$w.py > $GRAFT_REPO_ROOT/$out/$w.log 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/$w/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -14 "$f" > $GRAFT_REPO_ROOT/$out/kernel_stats_$w.csv
done
