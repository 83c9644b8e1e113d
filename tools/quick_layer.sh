# fused-layer tests + isolated per-kernel durations
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused_layer.py -x -q 2>&1 | grep -E "passed|failed|rror" > gpurun_out/ql_tests.log
bash tools/layer_kt.sh
