#!/bin/bash
# band height of the one-launch levels 1 + 2 under the driver-protocol bench (two streams) and on one stream
out=${1:-gpurun_out/march_band_sweep.txt}
: > $out
for br in ${BANDS:-32 36 40 44 48 56 64}; do
  echo "band $br: $(DTCWT_HIP_MARCH_BAND=$br python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["one_stream_ms_per_step"], d["roofline"]["fwd_kernel_ms"][0])')" >> $out
done
echo "band auto: $(python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["one_stream_ms_per_step"], d["roofline"]["fwd_kernel_ms"][0])')" >> $out
cat $out
