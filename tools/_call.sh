cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q; mkdir -p $O
# parity of the experiment against the default program
for v in 2 3 4; do DTCWT_HIP_PAIR10=$v python - <<'PY' 2>&1 | tail -3
import os, numpy as np
from dtcwt_amd.hip import Transform2d
rng = np.random.default_rng(3)
for shape in [(512, 768), (4096, 4096), (516, 1020)]:
    x = rng.standard_normal(shape).astype(np.float32)
    t = Transform2d(program='march')
    p = t.forward(x, nlevels=3)

    v = os.environ['DTCWT_HIP_PAIR10']; os.environ['DTCWT_HIP_PAIR10'] = '0'
    q = Transform2d(program='march').forward(x, nlevels=3)
    os.environ['DTCWT_HIP_PAIR10'] = v
    e = max(float(np.abs(a - b).max()) for a, b in zip(p.highpasses + (p.lowpass,), q.highpasses + (q.lowpass,)))
    print(shape, 'max abs diff', e)
PY
done
B="python bench.py --no-cpu-baseline --no-other-configs"
for rep in 1 2; do
for v in 0 2 3 4; do
  export DTCWT_HIP_PAIR10=$v
  for args in "--steps 200 --warmup 50" "--steps 200 --warmup 50 --streams 1" "--config c3 --steps 60 --warmup 20" "--config c5 --steps 20 --warmup 5"; do
    echo "PAIR10=$v $args: $($B $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('fwd_ms_per_step'), d.get('inv_ms_per_step'))")"
  done
done
done | tee $O/pair_headline.txt
