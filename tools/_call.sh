cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_transform1d3d.py -q -m gpu -x -k "long_filters" 2>&1 | tail -25
B="timeout 200 python bench.py --no-cpu-baseline --no-other-configs --config c4"
$B --steps 40 --warmup 10 --biort near_sym_b --qshift qshift_b > $O/c4_b.json 2>$O/c4_b.err; tail -c 900 $O/c4_b.json; tail -3 $O/c4_b.err
