#!/bin/bash
# round-5 final profile collection (GPU box, one gpurun call):
#   c2  tools/profile_round.sh on the driver's command (kernel trace 4 streams + 1 stream, FETCH_SIZE / WRITE_SIZE passes)
#   c3, c5, c4  the same four passes per config (counters and the "alone" medians with --streams 1 --cu-partition off, the
#       in-flight medians under the config's default) -> <out>/<cfg>/roofline.json
#   bench lines: driver's command, 500 steps, c3, c5, c5full, c4, --mgpu, near_sym_b / qshift_b
#   tools/merge_roofline_r05.py then writes profiles/r05/roofline.json + profiles/traffic.json (run here, after the call)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05prof
mkdir -p $O
bash $R/tools/profile_round.sh gpurun_out/r05prof/c2 r05 > $O/profile_round.log 2>&1
cd /tmp && export TMPDIR=/tmp
for c in c3 c5 c4; do
  mkdir -p $O/$c
  B="python $R/bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-probe --settle-ms 60"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$c/trace -o bench --output-format csv -- $B > $O/$c/bench_under_trace.json 2> $O/$c/trace.err
  echo "$c trace rc=$?" >> $O/status.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$c/trace1 -o bench --output-format csv -- $B --streams 1 --cu-partition off > $O/$c/bench_under_trace1.json 2> $O/$c/trace1.err
  echo "$c trace1 rc=$?" >> $O/status.txt
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/$c/pmc_fetch -o p --output-format csv -- $B --streams 1 --cu-partition off > $O/$c/pmc_fetch.log 2>&1
  echo "$c fetch rc=$?" >> $O/status.txt
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/$c/pmc_write -o p --output-format csv -- $B --streams 1 --cu-partition off > $O/$c/pmc_write.log 2>&1
  echo "$c write rc=$?" >> $O/status.txt
  cp "$(find $O/$c/trace -name "*kernel_stats.csv" | head -1)" $O/$c/kernel_stats.csv 2>/dev/null
  cp "$(find $O/$c/trace1 -name "*kernel_stats.csv" | head -1)" $O/$c/kernel_stats_streams1.csv 2>/dev/null
  (cd $R && python tools/roofline_from_trace.py $O/$c > $O/$c/roofline.json 2> $O/$c/roofline.err)
  find $O/$c -name "*kernel_trace.csv" -size +20M -delete
  find $O/$c -name "*counter_collection.csv" -size +20M -delete
done
find $O/c2 -name "*kernel_trace.csv" -size +20M -delete
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --steps 500 --no-cpu-baseline --no-other-configs > $O/bench_500steps.json 2>/dev/null
for c in c3 c5 c4; do python bench.py --config $c --no-cpu-baseline --steps 40 > $O/bench_$c.json 2>/dev/null; done
python bench.py --config c5full --no-cpu-baseline --steps 5 > $O/bench_c5full.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --mgpu > $O/bench_mgpu.json 2>/dev/null
python bench.py --biort near_sym_b --qshift qshift_b --steps 100 --no-cpu-baseline --no-other-configs > $O/bench_near_sym_b_qshift_b.json 2>/dev/null
DTCWT_HIP_MARCH=0 python bench.py --cu-partition off --no-cpu-baseline --no-other-configs --steps 100 > $O/bench_march_off.json 2>/dev/null
cat $O/status.txt; cat $O/c2/status.txt
# SQ counters of the kernels that are new this round (VALU / LDS / wait shares): C4's marching pair and the level-1 marches
bash $R/tools/pmc_cmd.sh gpurun_out/r05prof/pmc_c4 python $R/bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline --streams 1 --cu-partition off --settle-ms 60 > $O/pmc_c4.log 2>&1
bash $R/tools/pmc_cmd.sh gpurun_out/r05prof/pmc_nsb python $R/bench.py --biort near_sym_b --qshift qshift_b --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --streams 1 --settle-ms 60 > $O/pmc_nsb.log 2>&1
find $O/pmc_c4 $O/pmc_nsb -name "*.csv" -size +5M -delete
