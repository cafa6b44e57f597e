#!/bin/bash
# GPU box: the marching launches by batch shape at (nearly) equal pixel counts -- one stream, whole device, hipEvent pairs
# around every launch (bench.py's fwd_kernel_ms / inv_kernel_ms): why do batches run below the C2-in-flight rate?
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05b}
mkdir -p $O
cd $R
echo "# B R C | one_stream_ms | fwd_kernel_ms | inv_kernel_ms | ns/px fwd12m inv21m" > $O/batch_shapes.txt
for s in "4 4096 4096" "16 4096 4096" "64 2048 2048" "32 2048 4096" "32 4096 2048" "16 2048 8192" "16 8192 2048" "64 1024 4096" "64 4096 1024" "256 1024 1024" "4 8192 8192" "1 16384 16384"; do
  set -- $s
  python bench.py --config c5 --batch $1 --rows $2 --cols $3 --streams 1 --sets 2 --steps 10 --warmup 2 --settle-ms 100 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
px=$1*$2*$3
print('$1 $2 $3 |', d['one_stream_ms_per_step'], '|', r['fwd_kernel_ms'], '|', r['inv_kernel_ms'], '| %.3f %.3f' % (r['fwd_kernel_ms'][0]*1e6/px, r['inv_kernel_ms'][1]*1e6/px))
" >> $O/batch_shapes.txt
done
cat $O/batch_shapes.txt
