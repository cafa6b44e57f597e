cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for arg in "--streams 1" "--streams 2 --cu-partition on" "--streams 2 --cu-partition off" "--streams 4 --cu-partition on" "--streams 4 --cu-partition off" "--streams 3 --cu-partition off"; do
  echo "c5 $arg: $(python bench.py --config c5 --no-cpu-baseline $arg 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("sustained_ms_per_step"))')"
done; done
for arg in "--streams 2 --cu-partition on" "--streams 2 --cu-partition off" "--streams 4 --cu-partition off" "--streams 4 --cu-partition on"; do
  echo "c3 $arg: $(python bench.py --config c3 --no-cpu-baseline $arg 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("sustained_ms_per_step"))')"
done
