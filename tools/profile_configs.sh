#!/bin/bash
# GPU box: the other single-GPU configurations under rocprofv3 (c2 comes from tools/profile_round.sh): kernel trace under the config's
# default streams + one at a time, FETCH_SIZE / WRITE_SIZE passes one at a time (each --pmc pass in a run of its own: gpurun refuses
# counters together with the trace domains) -> <out>/<cfg>/roofline.json.  tools/merge_roofline.py merges them into
# profiles/rNN/roofline.json + profiles/traffic.json afterwards.
#   tools/profile_configs.sh <outdir under gpurun_out> [configs: c3 c5 c4 c4q]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/${1:-gpurun_out/prof}; shift || true
CFGS=${*:-c3 c5 c4 c4q}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  mkdir -p $O/$c
  case $c in
    c4q) A="--config c4 --biort near_sym_b --qshift qshift_b";;
    *)   A="--config $c";;
  esac
  B="python $R/bench.py $A --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-probe --no-clocks --settle-ms 60"
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
cd $R
for c in $CFGS; do
  case $c in
    c4q) A="--config c4 --biort near_sym_b --qshift qshift_b";;
    *)   A="--config $c";;
  esac
  python bench.py $A --no-cpu-baseline --steps 40 > $O/bench_$c.json 2>/dev/null
done
cat $O/status.txt
