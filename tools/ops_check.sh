cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | grep -E "passed|failed|rror" > gpurun_out/oc_tests.log
python tools/ops_roofline.py > gpurun_out/oc_roofline.txt 2>&1
