cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pointops2.py -x -q -k stratified 2>&1 | tail -15
