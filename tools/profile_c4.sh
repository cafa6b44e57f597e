#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of bench.py --config c4 (one and four volumes in flight) for one wavelet set
#   tools/profile_c4.sh <outdir under gpurun_out> [biort qshift]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/${1:-gpurun_out/prof_c4}
B=${2:-near_sym_b}; Q=${3:-qshift_b}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --config c4 --biort $B --qshift $Q --steps 40 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace1" -o bench --output-format csv -- $BENCH --streams 1 > "$OUT/bench_under_trace_streams1.json" 2> "$OUT/trace1.err"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace4" -o bench --output-format csv -- $BENCH > "$OUT/bench_under_trace.json" 2> "$OUT/trace4.err"
cd $R
cp "$(find $OUT/trace1 -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_streams1.csv" 2>/dev/null
cp "$(find $OUT/trace4 -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/trace1" "$OUT/trace4"
$BENCH > "$OUT/bench.json" 2>/dev/null
$BENCH --streams 1 > "$OUT/bench_streams1.json" 2>/dev/null
for f in kernel_stats_streams1.csv kernel_stats.csv; do echo "== $f"; cut -d, -f1-4 "$OUT/$f" | head -16; done
