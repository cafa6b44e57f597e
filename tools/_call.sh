cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-other-configs --config c4 --biort near_sym_b --qshift qshift_b --steps 40 --warmup 10"
for rep in 1 2; do
for a in "--streams 2" "--streams 3" "--streams 4" "--streams 6" "--streams 8" "--streams 4 --cu-partition on" "--streams 2 --cu-partition on"; do
  echo "$a: $(timeout 200 $B $a | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['fwd_ms_per_step'], d['inv_ms_per_step'])")"
done; done
