cd $GRAFT_REPO_ROOT; O=gpurun_out/r05l; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
cp gpurun_out/parity_worst.json $O/
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; python -c "
import json
d=json.loads(open('$O/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['sustained_ms_per_step'], d['one_stream_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
for k,v in d['other_configs'].items(): print(k, v.get('ms_per_step'), v.get('roofline'))
"
