cd $GRAFT_REPO_ROOT; O=gpurun_out/r05p; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|FAILED" $O/pytest_gpu.txt | tail -8
cp gpurun_out/parity_worst.json $O/
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
