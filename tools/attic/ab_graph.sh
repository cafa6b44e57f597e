#!/bin/bash
# stream launches vs hipGraph replay, with and without the CU partition, alternating in one call
cd "$(dirname "$0")/.."
for steps in 20 200; do
 for round in 1 2 3; do
  for f in "" "--graph" "--graph --cu-partition off" "--cu-partition off"; do
    echo "steps $steps [$f]: $(timeout 200 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-other-configs $f 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["launch"], d["ms_per_step"], d["one_stream_ms_per_step"])')"
  done
 done
done
