# per-pass caps of the matrix-core layer passes (PDFOPS_PT_CAP_P3 / B1 / B3): per-kernel durations at levels 2-5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export REPS=5 LEVELS=${LEVELS:-1,2,3,4}
for cap in 0 256 512 1024 2048; do
  export PDFOPS_PT_CAP_P3=$cap PDFOPS_PT_CAP_B1=$cap PDFOPS_PT_CAP_B3=$cap
  rm -rf /tmp/p/cs$cap
  rocprofv3 --kernel-trace --stats -d /tmp/p/cs$cap -o cs -- python3 $R/tools/pt_layer_bench.py > /tmp/cs$cap.log 2>&1
  echo "=== cap $cap" >> $R/gpurun_out/cap_sweep2.txt
  python3 $R/tools/rocpd_stats.py $(find /tmp/p/cs$cap -name "*.db" | head -1) 80 | grep -E "flm::k_|k_colsum|k_bn_finalize" | cut -c1-120 >> $R/gpurun_out/cap_sweep2.txt
done
