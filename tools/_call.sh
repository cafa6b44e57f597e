cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
cp gpurun_out/parity_worst.json $O/
python bench.py --config c4 --steps 40 --no-cpu-baseline > $O/bench_c4.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('ms_per_step','ms_per_step_one_stream','fwd_ms_per_step','inv_ms_per_step','step_frac')}, d['roofline']['kernel_ms'], d['roofline']['frac'])"
