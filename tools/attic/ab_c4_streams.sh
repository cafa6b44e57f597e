for s in 1 2 3 4 1 2 4; do
  echo "streams $s"; timeout 200 python bench.py --config c4 --no-cpu-baseline --streams $s --steps 60 --warmup 10 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','fwd_ms_per_step','inv_ms_per_step','ms_per_step_one_stream','step_frac','recon_max_abs_err')})"
done
