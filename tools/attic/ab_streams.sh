#!/bin/bash
# stream count of the driver-protocol bench, alternating in one call: tools/ab_streams.sh "2 3 4" rounds [bench args]
S=$1; N=${2:-3}; shift 2 || true
for i in $(seq $N); do
  for st in $S; do
    echo "streams $st: $(python bench.py --no-cpu-baseline --no-other-configs --streams $st "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["one_stream_ms_per_step"])')"
  done
done
