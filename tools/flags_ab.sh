# compiler-flag A/B of the hot translation units on the default bench schedule: bash tools/flags_ab.sh <tag> "<flags a>" "<flags b>" ...
# ("" = the tree's own flags).  Every setting rebuilds the listed files on the box (PDFOPS_EXTRA_FLAGS), runs the short bench twice.
TAG=$1; shift
cd $GRAFT_REPO_ROOT
FILES="fused_layer.hip,fused_layer_mfma.hip,fused_layer_slab.hip,rowlin2.hip,transition_down.hip,pointwise.hip,seg_gather.hip,block.hip"
OUT=gpurun_out/${TAG}_flags_ab.txt; : > $OUT
for F in "$@"; do
  for f in ${FILES//,/ }; do touch pointcloudpdf_amd/csrc/$f; done
  if [ -n "$F" ]; then PDFOPS_EXTRA_FLAGS="$FILES:$F" python -m pointcloudpdf_amd.build > /dev/null 2> gpurun_out/${TAG}_flags_build.err || { echo "[$F] BUILD FAILED: $(tail -1 gpurun_out/${TAG}_flags_build.err | cut -c1-200)" >> $OUT; continue; }
  else python -m pointcloudpdf_amd.build > /dev/null 2>&1; fi
  for r in 1 2; do
    line=$(python3 bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep 2>/dev/null | tail -1)
    echo "[$F] run $r: $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms/step loss", d.get("loss"))')" >> $OUT
  done
done
cat $OUT
