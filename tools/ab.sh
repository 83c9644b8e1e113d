# usage: bash tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one configuration, run twice interleaved)
for rep in 1 2; do
for cfg in "$@"; do
echo "$cfg :: $(env $cfg python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]; print(round(d["ms_per_step"],2), "ptf", round(k["pt_layer_forward"]["avg_ms"],3), "ptb", round(k["pt_layer_backward"]["avg_ms"],3))')"
done
done
