#!/bin/bash
# C4 forward / inverse under each of the given DTCWT_HIP_XCD3D masks, in one call: tools/ab_3d_env.sh 0 3 7 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2; do for x in "$@"; do
  echo "== XCD3D=$x ($i)"
  DTCWT_HIP_XCD3D=$x REPS=30 python $R/tools/prof3d_fwd.py
  DTCWT_HIP_XCD3D=$x REPS=30 python $R/tools/prof3d_inv.py 2>&1 | tail -2 | head -1
done; done
