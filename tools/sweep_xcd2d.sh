#!/bin/bash
# per-kernel times of the 2-D bench under DTCWT_HIP_XCD_ORDER values (applies to every fused 2-D kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2; do for x in "" 0 1 2 4 8 16; do
  DTCWT_HIP_XCD_ORDER=$x python $R/bench.py --steps 200 --no-cpu-baseline --streams 1 > /tmp/x.json 2>/dev/null
  [ -z "$x" ] && unset DTCWT_HIP_XCD_ORDER
  python -c "
import json;d=json.load(open('/tmp/x.json'));print('order=%-4s %d  %.5f' % ('${x:-dflt}', $i, d['ms_per_step']), d['roofline']['fwd_kernel_ms'], d['roofline']['inv_kernel_ms'])"
done; done
