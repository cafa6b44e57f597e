#!/bin/bash
# tools/ab_cu_mask.py, one or two variants per process (a process full of masked queues is slow as a whole)
cd "$(dirname "$0")/.."
echo "4096 x 4096 f32 nlevels=4 forward + inverse, 200 steps per measurement, ms per step; each line a process of its own"
for round in 1 2; do
  for v in 0,1 0,2 0,3 0,4 0,5 0,6 0,7 0,8 0,9; do
    timeout 120 python tools/ab_cu_mask.py 200 2 2 $v 2>&1 | grep -v amdgpu.ids
  done
done
