cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_model.py tests/test_gpu_ops.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5 > gpurun_out/bstats_tests.log
bash tools/prof_quick.sh
for f in q_kernels q_categories q_queues; do cp gpurun_out/$f.txt gpurun_out/bs1_$f.txt; done
export PDFOPS_DGRAD_BSTATS=0
bash tools/prof_quick.sh
for f in q_kernels q_categories q_queues; do cp gpurun_out/$f.txt gpurun_out/bs0_$f.txt; done
