cd $GRAFT_REPO_ROOT
for rep in 1 2; do for st in fp32 bf16; do
python bench.py --no-cpu-baseline --no-ops-roofline --no-latency-sweep --storage $st 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]; print(sys.argv[1], round(d["ms_per_step"],2), round(k["bottleneck_backward"]["avg_ms"],4), round(k["bottleneck_forward"]["avg_ms"],4), d["dtype"])' $st >> gpurun_out/ab_storage.log
done; done
