# total-workgroup target of the row-linear forward / dgrad kernels (PDFOPS_RL_BLOCKS): per-kernel times from a kernel trace of the bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 128 256 384 512; do
export PDFOPS_RL_BLOCKS=$t
rm -rf /tmp/p/rl$t
timeout 300 rocprofv3 --kernel-trace -d /tmp/p/rl$t -o kt -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-latency-sweep --no-ops-roofline > /tmp/rl$t.log 2>&1
echo "=== total $t $(grep -o '"ms_per_step": [0-9.]*' /tmp/rl$t.log | head -1)" >> $R/gpurun_out/rl_sweep.txt
python3 $R/tools/rocpd_categories.py $(find /tmp/p/rl$t -name "*.db" | head -1) 10 | grep -E "rowlin" >> $R/gpurun_out/rl_sweep.txt
python3 $R/tools/rocpd_stats.py $(find /tmp/p/rl$t -name "*.db" | head -1) 80 | grep -E "rl2::k_fwd" | head -8 | cut -c1-110 >> $R/gpurun_out/rl_sweep.txt
done
