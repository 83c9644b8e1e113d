# in-flight workgroup target of the weight-gradient kernels (PDFOPS_WG_BLOCKS): k_wg times from a kernel trace of the bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 128 256 512 1024 2048; do
export PDFOPS_WG_BLOCKS=$t
rm -rf /tmp/p/wg$t
timeout 300 rocprofv3 --kernel-trace -d /tmp/p/wg$t -o kt -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-latency-sweep --no-ops-roofline > /tmp/wg$t.log 2>&1
echo "=== target $t" >> $R/gpurun_out/wg_sweep2.txt
python3 $R/tools/rocpd_categories.py $(find /tmp/p/wg$t -name "*.db" | head -1) 10 | grep -E "rowlin|copy/fill" >> $R/gpurun_out/wg_sweep2.txt
python3 $R/tools/rocpd_stats.py $(find /tmp/p/wg$t -name "*.db" | head -1) 80 | grep -E "rl2::k_wg" | cut -c1-100 >> $R/gpurun_out/wg_sweep2.txt
done
