#!/bin/bash
# band height and stream count of the driver-protocol bench with the one-launch levels 1 + 2
out=gpurun_out/march_band_sweep.txt
: > $out
for br in 40 48 56 64 80 96; do
  for st in 2; do
    echo "band $br streams $st: $(DTCWT_HIP_MARCH_BAND=$br python bench.py --no-cpu-baseline --streams $st 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["one_stream_ms_per_step"], d["roofline"]["fwd_kernel_ms"])')" >> $out
  done
done
for st in 1 3 4; do
  echo "band auto streams $st: $(python bench.py --no-cpu-baseline --streams $st 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["one_stream_ms_per_step"], d["roofline"]["fwd_kernel_ms"])')" >> $out
done
echo "march off streams 3: $(DTCWT_HIP_MARCH=0 python bench.py --no-cpu-baseline --streams 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["one_stream_ms_per_step"])')" >> $out
cat $out
