# compiler-flag A/B on the GPU box: bash tools/flags_ab.sh "<files,comma,separated>" "<flag set 1>" "<flag set 2>" ...   (baseline first)
cd $GRAFT_REPO_ROOT
F=${1:-"fused_layer.hip,fused_layer_mfma.hip,rowlin2.hip,block.hip,pointwise.hip,transition_down.hip,seg_gather.hip"}; shift
run() { python bench.py --steps 36 --warmup 12 --no-cpu-baseline --no-ops-roofline --no-latency-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels']['bottleneck_backward']['avg_ms'], d['kernels']['bottleneck_forward']['avg_ms'])"; }
echo baseline; run; run
for fl in "$@"; do
  for f in $(echo $F | tr ',' ' '); do touch pointcloudpdf_amd/csrc/$f; done
  PDFOPS_EXTRA_FLAGS="$F:$fl" python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
  echo "$fl"; python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_fused_layer.py -x -q -m gpu 2>&1 | tail -1; run; run
done
