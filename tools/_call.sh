cd $GRAFT_REPO_ROOT; O=gpurun_out/r05m; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|FAILED" $O/pytest_gpu.txt | tail -8
cp gpurun_out/parity_worst.json $O/
