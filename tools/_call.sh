cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_transform1d3d.py -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
