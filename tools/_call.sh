cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_transform1d3d.py -q -m gpu -x -k "long_filters" 2>&1 | tail -15
