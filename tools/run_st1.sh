cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pointops2.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head -20 > gpurun_out/st_tests.log
timeout 300 python tools/pointops2_bench.py > gpurun_out/st_p2.txt 2>&1
timeout 600 python bench.py --workload stratified --steps 8 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("stratified", round(d["ms_per_step"],2), d["value"])' > gpurun_out/st_bench.txt
