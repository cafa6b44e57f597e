cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-other-configs --config c4 --biort near_sym_b --qshift qshift_b --steps 40 --warmup 10"
for rep in 1 2; do
for br in 0 40 64 80 128 256; do
  if [ $br = 0 ]; then unset DTCWT_HIP_MARCH_BAND; else export DTCWT_HIP_MARCH_BAND=$br; fi
  echo "band $br: streams1 $(timeout 200 $B --streams 1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['fwd_ms_per_step'], d['inv_ms_per_step'])")  streams4 $(timeout 200 $B | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['fwd_ms_per_step'], d['inv_ms_per_step'])")"
done; done
