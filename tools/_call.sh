cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05u
timeout 400 python tools/soak_long3d.py 150 1 2>&1 | tail -3 | tee gpurun_out/r05u/soak_long3d.txt
