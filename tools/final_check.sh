# end-of-round check: full GPU suite, smoke, PMC passes + traffic file for the current kernel sources
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" > gpurun_out/fin_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/fin_smoke.log 2>&1
timeout 1200 bash tools/pmc_only.sh r02_j
