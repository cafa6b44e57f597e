cd $GRAFT_REPO_ROOT; O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_march.py -x -q -m gpu -k include_scale > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { echo "$1 | $2: $(env $1 python bench.py --no-cpu-baseline --no-other-configs $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('one-stream', d.get('one_stream_ms_per_step'), 'resident', d.get('resident_ms_per_step'), 'fwd12m', r['fwd_kernel_ms'][0], 'inv21m', r['inv_kernel_ms'][1])")"; }
for rep in 1 2; do for br in 40 56 64 72 80 96 112; do run DTCWT_HIP_MARCH_BAND=$br "--steps 40 --streams 1"; done; run DTCWT_X=0 "--steps 40 --streams 1"; done 2>&1 | tee $O/band_sweep_alone.txt
