cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_transform1d3d.py -q -m gpu -x -k "long_filters" 2>&1 | tail -25
B="python bench.py --no-cpu-baseline --no-other-configs --config c4 --biort near_sym_b --qshift qshift_b"
timeout 200 $B --steps 40 --warmup 10 > $O/c4_b2.json 2>$O/c4_b.err; python -c "
import json; d=json.loads(open('$O/c4_b2.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('ms_per_step','fwd_ms_per_step','inv_ms_per_step','ms_per_step_one_stream','recon_max_abs_err')})"
rm -rf $O/trace; timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- $B --steps 20 --warmup 5 --streams 1 --settle-ms 50 > $O/bench_under_trace.json 2> $O/trace.err
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-200
for ch in 64 86 256; do echo chunk $ch: $(DTCWT_HIP_LONG3D_CHUNK=$ch timeout 200 $B --steps 40 --warmup 10 --streams 1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['fwd_ms_per_step'], d['inv_ms_per_step'])"); done
for ch in 16 32 64 128; do echo ichunk $ch: $(DTCWT_HIP_LONG3D_ICHUNK=$ch timeout 200 $B --steps 40 --warmup 10 --streams 1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['fwd_ms_per_step'], d['inv_ms_per_step'])"); done
