cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 36 --warmup 12 --no-cpu-baseline --no-ops-roofline --no-latency-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels']['bottleneck_backward']['avg_ms'], d['kernels']['bottleneck_forward']['avg_ms'])"; }
F="fused_layer.hip,fused_layer_mfma.hip,rowlin2.hip,block.hip,pointwise.hip,transition_down.hip,seg_gather.hip"
echo baseline; run
for fl in "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy" "-mllvm -amdgpu-schedule-relaxed-occupancy=true" "-mllvm -amdgpu-use-amdgpu-trackers=1"; do
  touch pointcloudpdf_amd/csrc/fused_layer.hip pointcloudpdf_amd/csrc/fused_layer_mfma.hip pointcloudpdf_amd/csrc/rowlin2.hip pointcloudpdf_amd/csrc/block.hip pointcloudpdf_amd/csrc/pointwise.hip pointcloudpdf_amd/csrc/transition_down.hip pointcloudpdf_amd/csrc/seg_gather.hip
  PDFOPS_EXTRA_FLAGS="$F:$fl" python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
  echo "$fl"; run
done
