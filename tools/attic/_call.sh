cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05x
( python bench.py --no-cpu-baseline --no-other-configs --steps 150000 --warmup 50 > gpurun_out/r05x/long.json 2>/dev/null ) &
BP=$!
for i in $(seq 1 30); do echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' ')"; sleep 1; done | tee gpurun_out/r05x/clocks_c2.txt
wait $BP
python -c "import json; d=json.loads(open('gpurun_out/r05x/long.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])"
