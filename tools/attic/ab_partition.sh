#!/bin/bash
# bench.py with and without the CU-partitioned contexts, alternating in one call: driver's command (c2), c3, c5 share, c4
cd "$(dirname "$0")/.."
pick='import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ("value","ms_per_step","one_stream_ms_per_step","ms_per_step_one_stream","recon_max_abs_err")}, d["config"].get("cu_partition"))'
for round in 1 2; do
  for flag in "--cu-partition on" "--cu-partition off"; do
    echo "== c2 driver command $flag"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $flag | grep '^{' | python -c "$pick"
    echo "== c2 300 steps $flag"; timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-other-configs $flag | grep '^{' | python -c "$pick"
    echo "== c3 $flag"; timeout 300 python bench.py --config c3 --steps 40 --warmup 5 --no-cpu-baseline $flag | grep '^{' | python -c "$pick"
    echo "== c5 share $flag"; timeout 300 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline $flag | grep '^{' | python -c "$pick"
    echo "== c4 $flag"; timeout 300 python bench.py --config c4 --steps 60 --warmup 10 --no-cpu-baseline $flag | grep '^{' | python -c "$pick"
  done
done
