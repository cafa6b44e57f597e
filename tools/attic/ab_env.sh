#!/bin/bash
# A/B of one environment setting under the driver-protocol bench, alternating in one call so that both arms see the same
# box: tools/ab_env.sh "<VAR=value for arm A>" "<VAR=value for arm B>" [rounds] [extra bench args]
A=$1; B=$2; N=${3:-4}; shift 3 || true
for i in $(seq $N); do
  for arm in "$A" "$B"; do
    echo "$arm: $(env $arm python bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["one_stream_ms_per_step"])')"
  done
done
