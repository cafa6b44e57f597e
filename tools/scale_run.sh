#!/bin/bash
# The scaling runs of BASELINE's metric on one node, for whoever has N > 1 MI355X (the build boxes had one):
# N = 1, 2, 4, 8 for the headline config (c2: one 4096^2 image per GPU per step) and for C5 (64 x 2048^2 per GPU
# per step = the 512-image batch at N = 8), in both multi-GPU forms:
#   ranks  one process per GPU under torch.distributed.run (bench.py spawns them), RCCL tap broadcast + barriers
#   mgpu   one process, dtcwt_hip_mgpu_* with a host thread per device
# Every line of <outdir>/scale.jsonl is one bench.py JSON line (n_gpus, value, ms_per_step, nccl_ranks,
# rank_ms_per_step min / max ...).  Linear weak scaling reads N x the N = 1 value: 8 x ~82,000 Mpix/s for C5.
# Usage: tools/scale_run.sh [outdir] [steps]      (SCALE_RUN_FAST=1: no CPU baseline at N = 1 either)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/scale}
STEPS=${2:-50}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NDEV=$(python -c "import sys; sys.path.insert(0, '$R'); from dtcwt_amd.hip import _lib; print(_lib.device_count())")
: > "$OUT/scale.jsonl"
for cfg in c2 c5; do
  for n in 1 2 4 8; do
    if [ "$n" -gt "$NDEV" ]; then echo "skip $cfg N=$n: $NDEV device(s) visible" | tee -a "$OUT/skipped.txt"; continue; fi
    base="--no-other-configs"; { [ "$n" -gt 1 ] || [ "${SCALE_RUN_FAST:-0}" = 1 ]; } && base="$base --no-cpu-baseline"
    for mode in ranks mgpu; do
      extra=""; [ "$mode" = mgpu ] && extra="--mgpu"
      [ "$cfg" = c5 ] && steps=$((STEPS / 5 + 2)) || steps=$STEPS
      echo "== $cfg N=$n $mode" >&2
      timeout 900 python "$R/bench.py" --gpus $n --config $cfg --steps $steps --warmup 5 $base $extra \
          2> "$OUT/${cfg}_n${n}_${mode}.err" | grep '^{' | tail -1 | tee -a "$OUT/scale.jsonl"
    done
  done
done
python - "$OUT/scale.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith('{')]
one = {}
for r in rows:
    key = (r['metric'], r['launch'].split(' ')[0])
    if r['n_gpus'] == 1:
        one[key] = r['value']
    eff = r['value'] / (r['n_gpus'] * one[key]) if key in one else float('nan')
    print('%-70s %-8s N=%d  %10.0f Mpix/s  %.4f ms/step  efficiency %.3f' % (r['metric'][:70], key[1], r['n_gpus'], r['value'], r['ms_per_step'], eff))
PY
