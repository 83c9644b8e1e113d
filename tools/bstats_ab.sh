# A/B of the dgrad-epilogue BatchNorm-backward sums (PDFOPS_DGRAD_BSTATS) + the dense / model parity tests
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab_env.log
timeout 1200 python -m pytest tests/test_gpu_dense.py tests/test_gpu_pointwise.py tests/test_gpu_model.py -x -q 2>&1 | tail -5 > gpurun_out/bstats_tests.log
timeout 900 bash tools/ab_env.sh "PDFOPS_DGRAD_BSTATS=0" "PDFOPS_DGRAD_BSTATS=1"
