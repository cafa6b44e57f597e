#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + HBM-traffic PMC passes of the bench command.
# Usage: tools/profile_round.sh <outdir under gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1"     # one stream: kernels of a step do not overlap in the trace
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- $BENCH > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
echo "trace rc=$?" > "$OUT/status.txt"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p --output-format csv -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "fetch rc=$?" >> "$OUT/status.txt"
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p --output-format csv -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "write rc=$?" >> "$OUT/status.txt"
cd $R
python $R/bench.py > "$OUT/bench_plain.json" 2>/dev/null      # the default command (driver contract)
python $R/bench.py --streams 1 --no-cpu-baseline > "$OUT/bench_streams1.json" 2>/dev/null
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1
cat "$OUT/status.txt"
