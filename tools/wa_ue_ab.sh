# A/B of the elements-per-trip knob of the window-attention table-gradient kernels (rebuilds one translation unit per setting on the box)
cd $GRAFT_REPO_ROOT
X="--steps 12 --warmup 4 --no-cpu-baseline --no-ops-roofline --no-latency-sweep --workload stratified"
for ue in 2 4 8; do
  touch pointcloudpdf_amd/csrc/window_attention.hip
  PDFOPS_WA_UE=$ue python -m pointcloudpdf_amd.build > /dev/null 2>gpurun_out/wa_build.err || { echo "build failed ue=$ue"; tail -5 gpurun_out/wa_build.err; continue; }
  python bench.py $X 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('UE=$ue', round(d['ms_per_step'], 2), 'ms/step;', {n: round(k[n]['avg_ms'], 3) for n in k if 'backward' in n and ('dot_prod' in n or 'step2' in n)})"
done 2>&1 | tee gpurun_out/r03_wa_ue_ab.log
touch pointcloudpdf_amd/csrc/window_attention.hip
