# pointwise/rowlin/model GPU tests + bench + quick profile
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_fused_layer.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|rror" > gpurun_out/mc_tests.log
python bench.py --no-cpu-baseline --no-ops-roofline --no-latency-sweep 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]; print(round(d["ms_per_step"],2), {n: round(v["avg_ms"],4) for n,v in k.items()})' > gpurun_out/mc_bench.log
bash tools/prof_quick.sh
