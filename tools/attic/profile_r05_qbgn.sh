#!/bin/bash
# round 5, "qbgn-style" 3-D (near_sym_b / qshift_b, BASELINE configs[3]): bench lines before (DTCWT_HIP_LONG3D=0: the axis-by-axis
# generic level 1) and after (fused3d_long.hpp), rocprofv3 kernel trace of one volume at a time, FETCH_SIZE / WRITE_SIZE passes.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05qbgn
mkdir -p $O
cd $R
W="--config c4 --biort near_sym_b --qshift qshift_b --no-cpu-baseline --no-other-configs"
DTCWT_HIP_LONG3D=0 timeout 300 python bench.py $W --steps 20 --warmup 5 > $O/bench_before.json 2> $O/bench_before.err
timeout 300 python bench.py $W --steps 40 --warmup 10 > $O/bench_after.json 2> $O/bench_after.err
timeout 300 python bench.py $W --steps 40 --warmup 10 --streams 1 > $O/bench_after_streams1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py $W --steps 20 --warmup 5 --streams 1 --settle-ms 60"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- $B > $O/bench_under_trace.json 2> $O/trace.err
echo "trace rc=$?" > $O/status.txt
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- $B > $O/pmc_fetch.log 2>&1
echo "fetch rc=$?" >> $O/status.txt
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- $B > $O/pmc_write.log 2>&1
echo "write rc=$?" >> $O/status.txt
cp "$(find $O/trace -name "*kernel_stats.csv" | head -1)" $O/kernel_stats.csv 2>/dev/null
(cd $R && python tools/roofline_from_trace.py $O > $O/roofline.json 2> $O/roofline.err)
find $O -name "*kernel_trace.csv" -size +20M -delete
cat $O/status.txt; head -12 $O/kernel_stats.csv | cut -c1-160
