cd $GRAFT_REPO_ROOT; O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_march.py -x -q -m gpu -k "level1" > $O/pytest_l1.txt 2>&1; tail -5 $O/pytest_l1.txt
for w in "near_sym_b qshift_b" "near_sym_b qshift_d"; do set -- $w
for prog in 0 1; do
  echo "== $1 $2 DTCWT_HIP_MARCH=$prog"
  DTCWT_HIP_MARCH=$prog python bench.py --biort $1 --qshift $2 --steps 60 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], d['sustained_ms_per_step'], d['one_stream_ms_per_step'], r['step_frac'], r['fwd_kernel_ms'], r['inv_kernel_ms'], d['recon_max_abs_err'])"
done; done 2>&1 | tee $O/bench_l1.txt
for br in 20 40 60 80 120 160; do echo "band $br: $(DTCWT_HIP_MARCH_BAND=$br python bench.py --biort near_sym_b --qshift qshift_b --steps 40 --streams 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['one_stream_ms_per_step'], r['fwd_kernel_ms'][0], r['inv_kernel_ms'][0])")"; done 2>&1 | tee $O/band_l1.txt
