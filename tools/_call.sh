cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05u
timeout 900 python -m pytest tests/test_hip_march.py -q -m gpu 2>&1 | tail -3
timeout 300 python tools/soak_march.py 60 5 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-other-configs --biort legall --qshift qshift_06 --steps 100 --warmup 20"
for rep in 1 2; do
echo "legall/qshift_06 inverse on tiles : $(DTCWT_HIP_MARCH_INV=0 timeout 200 $B | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['one_stream_ms_per_step'])")"
echo "legall/qshift_06 inverse fused    : $(timeout 200 $B | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['one_stream_ms_per_step'])")"
done | tee gpurun_out/r05u/legall_inverse.txt
