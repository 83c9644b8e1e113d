cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5 > gpurun_out/rl_tests.log
bash tools/prof_quick.sh
python bench.py --no-cpu-baseline --no-ops-roofline --no-latency-sweep 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), d["host_enqueue_ms_per_step"])' > gpurun_out/rl_bench.log
python bench.py --no-cpu-baseline --no-ops-roofline --no-latency-sweep 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), d["host_enqueue_ms_per_step"])' >> gpurun_out/rl_bench.log
