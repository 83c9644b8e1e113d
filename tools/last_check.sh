# last call of a round when the kernel sources did not change since the profiles: full GPU suite, smoke, two default bench lines
cd $GRAFT_REPO_ROOT
TAG=${1:-r02_q}
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" > gpurun_out/${TAG}_last_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 > gpurun_out/${TAG}_last_smoke.log
timeout 900 python bench.py > gpurun_out/${TAG}_last_bench.json 2> gpurun_out/${TAG}_last_bench.err
timeout 900 python bench.py --no-cpu-baseline --no-ops-roofline --no-latency-sweep > gpurun_out/${TAG}_last_bench2.json 2>> gpurun_out/${TAG}_last_bench.err
