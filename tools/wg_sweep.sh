cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 768 512 256; do
export PDFOPS_WG_BLOCKS=$t
rocprofv3 --kernel-trace -d /tmp/p/wg$t -o rl -- python3 $R/tools/bench_rowlin.py > /dev/null 2>&1
echo "target $t" >> $R/gpurun_out/wg_sweep.txt
python3 $R/tools/rocpd_stats.py $(find /tmp/p/wg$t -name "*.db" | head -1) 80 | grep "k_wg" | cut -c1-110 >> $R/gpurun_out/wg_sweep.txt
done
