#!/bin/bash
# per-kernel times of the 2-D bench under DTCWT_HIP_XCD_ORDER values (applies to every fused 2-D kernel);
# "dflt" = variable unset (per-kernel defaults: groups of 8 for k_fwd1, one run per XCD for the others)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {
  python $R/bench.py --steps 200 --no-cpu-baseline --streams 1 > /tmp/x.json 2>/dev/null
  python -c "
import json;d=json.load(open('/tmp/x.json'));print('order=%-4s %d  %.5f' % ('$1', $2, d['ms_per_step']), d['roofline']['fwd_kernel_ms'], d['roofline']['inv_kernel_ms'])"
}
for i in 1 2; do
  unset DTCWT_HIP_XCD_ORDER; run dflt $i
  for x in 0 1 2 4 8 16; do export DTCWT_HIP_XCD_ORDER=$x; run $x $i; done
done
