# small persistent grids at level 5 (780 points): per-kernel durations for PDFOPS_PT_BLOCKS_MFMA = 32 .. 256
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export REPS=5 LEVELS=${LEVELS:-3,4}
for cap in 32 64 96 128 256; do
  export PDFOPS_PT_BLOCKS_MFMA=$cap
  rm -rf /tmp/p/cs$cap
  rocprofv3 --kernel-trace --stats -d /tmp/p/cs$cap -o cs -- python3 $R/tools/pt_layer_bench.py > /tmp/cs$cap.log 2>&1
  echo "=== cap $cap" >> $R/gpurun_out/cap_sweep3.txt
  python3 $R/tools/rocpd_stats.py $(find /tmp/p/cs$cap -name "*.db" | head -1) 80 | grep -E "flm::k_|k_colsum|k_bn_finalize" | cut -c1-120 >> $R/gpurun_out/cap_sweep3.txt
done
