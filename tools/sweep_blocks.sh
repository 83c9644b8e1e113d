for cfg in "512 256" "512 512" "1024 512" "1024 256" "512 384" "768 768"; do
set -- $cfg
export PDFOPS_PT_BLOCKS_FWD=$1 PDFOPS_PT_BLOCKS_BWD=$2
echo "fwd=$1 bwd=$2 $(python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]; print(round(d["ms_per_step"],2), "ptf", round(k["pt_layer_forward"]["avg_ms"],3), "ptb", round(k["pt_layer_backward"]["avg_ms"],3))')"
done
