cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05w
timeout 600 python -m pytest tests/test_hip_march.py tests/test_hip_transform2d.py -q -m gpu -x 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-other-configs"
for rep in 1 2 3; do
for v in 0 1; do
  export DTCWT_HIP_INV21_NOGAIN=$v
  echo "NOGAIN=$v c2 300 steps: $(timeout 200 $B --steps 300 --warmup 50 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['one_stream_ms_per_step'])")   c5: $(timeout 200 $B --config c5 --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")  c3: $(timeout 200 $B --config c3 --steps 60 --warmup 10 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")"
done; done | tee gpurun_out/r05w/ab_nogain.txt
