#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + HBM-traffic PMC passes of THE DRIVER'S bench command
# (`python bench.py --gpus 1 --steps 20 --warmup 5`: four streams on quarters of the compute units, rotating buffer sets), then the per-kernel
# medians / traffic as JSON (tools/roofline_from_trace.py).
# Usage: tools/profile_round.sh <outdir under gpurun_out>      e.g. gpurun_out/r03prof
# Copy <outdir>/{kernel_stats.csv,roofline.json,pmc_hbm_traffic.txt,bench_*.json} into profiles/rNN/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-probe"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- $BENCH > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
echo "trace rc=$?" > "$OUT/status.txt"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace1" -o bench --output-format csv -- $BENCH --streams 1 > "$OUT/bench_under_trace_streams1.json" 2> "$OUT/trace1.err"
echo "trace1 rc=$?" >> "$OUT/status.txt"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p --output-format csv -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "fetch rc=$?" >> "$OUT/status.txt"
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p --output-format csv -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "write rc=$?" >> "$OUT/status.txt"
cd $R
cp "$(find $OUT/trace -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
cp "$(find $OUT/trace1 -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_streams1.csv" 2>/dev/null
python tools/roofline_from_trace.py "$OUT" --write "$OUT/traffic.json" --source "profiles/${2:-r04}/roofline.json" > "$OUT/roofline.json" 2> "$OUT/roofline.err"
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_hbm_traffic.txt" 2>&1
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2>/dev/null      # the driver's command, unprofiled
python $R/bench.py --streams 1 --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 > "$OUT/bench_streams1.json" 2>/dev/null
for st in 3 4; do python $R/bench.py --streams $st --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 > "$OUT/bench_streams$st.json" 2>/dev/null; done
cat "$OUT/status.txt"
