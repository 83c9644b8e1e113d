# A/B of the point walk of the matrix-core layer passes: grid-stride / chunked / chunked + Morton visiting order
R=$GRAFT_REPO_ROOT
for v in "0 0" "1 0" "1 1" "0 1" "0 0"; do
  set -- $v
  PDFOPS_LAYER_CHUNKED=$1 PDFOPS_LAYER_ORDER=$2 python3 $R/bench.py --no-cpu-baseline --no-ops-roofline --no-latency-sweep 2>/dev/null > /tmp/ws.json
  python3 -c "import json; d=json.load(open('/tmp/ws.json')); print('chunked=$1 order=$2', round(d['ms_per_step'],2), round(d['kernels']['bottleneck_backward']['avg_ms'],4), round(d['kernels']['bottleneck_forward']['avg_ms'],4))" >> $R/gpurun_out/walk_sweep.log
done
