cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
cp gpurun_out/parity_worst.json $O/ 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_default.json 2>$O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --mgpu > $O/bench_mgpu.json 2>$O/bench_mgpu.err
python bench.py --gpus 1 --steps 300 --warmup 5 --no-cpu-baseline --mgpu > $O/bench_mgpu_300.json 2>>$O/bench_mgpu.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --mgpu --config c5 > $O/bench_mgpu_c5.json 2>>$O/bench_mgpu.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --config c5 > $O/bench_c5.json 2>>$O/bench_default.err
for f in bench_default bench_mgpu bench_mgpu_300 bench_mgpu_c5 bench_c5; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d.get('sustained_ms_per_step'), d.get('one_stream_ms_per_step'), d['recon_max_abs_err'], d['config'].get('cu_partition'))
"; done
tail -3 $O/bench_mgpu.err
