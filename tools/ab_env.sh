# usage: bash tools/ab_env.sh "ENV=a" "ENV=b" ...  (each arg one configuration; two interleaved repetitions; appends to gpurun_out/ab_env.log)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "$@"; do
echo "$cfg :: $(env $cfg python bench.py --no-cpu-baseline --no-ops-roofline --no-latency-sweep 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]; print(round(d["ms_per_step"],2), "bf", round(k["bottleneck_forward"]["avg_ms"],4), "bb", round(k["bottleneck_backward"]["avg_ms"],4))')" >> gpurun_out/ab_env.log
done
done
