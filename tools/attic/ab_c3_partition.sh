cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3 4; do
for arg in "--cu-partition off" "--cu-partition on"; do
  echo "c3 $arg: $(python bench.py --config c3 --no-cpu-baseline $arg 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("sustained_ms_per_step"), d["one_stream_ms_per_step"])')"
done; done
