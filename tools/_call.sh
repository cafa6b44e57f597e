cd $GRAFT_REPO_ROOT; O=gpurun_out/r05v; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|FAILED|ERROR" $O/pytest_gpu.txt | tail -8
cp gpurun_out/parity_worst.json $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05v/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('sustained_ms_per_step'), d['roofline']['frac'])
for k,v in d['other_configs'].items(): print(k, {a:v.get(a) for a in ('ms_per_step','fwd_ms_per_step','inv_ms_per_step','step_frac','wall_s','error')})
PY
