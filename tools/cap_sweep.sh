# per-kernel durations of the matrix-core layer passes at levels 3-5 for several persistent-grid caps (PDFOPS_PT_BLOCKS_MFMA)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export REPS=5 LEVELS=${LEVELS:-2,3,4}
for cap in 0 512 1024 2048; do
  export PDFOPS_PT_BLOCKS_MFMA=$cap
  rm -rf /tmp/p/cs$cap
  rocprofv3 --kernel-trace --stats -d /tmp/p/cs$cap -o cs -- python3 $R/tools/pt_layer_bench.py > /tmp/cs$cap.log 2>&1
  echo "=== cap $cap" >> $R/gpurun_out/cap_sweep.txt
  python3 $R/tools/rocpd_stats.py $(find /tmp/p/cs$cap -name "*.db" | head -1) 80 | grep -E "flm::k_|k_colsum|k_bn_finalize|k_seg" | cut -c1-120 >> $R/gpurun_out/cap_sweep.txt
done
