# config 4 with the pseudo-label pass (2 x 150k ScanNet-shaped points, ONE captured graph): kernel trace table; bash tools/prof_pl.sh <tag>
TAG=${1:-r05_pl}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p/pl
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p/pl -o kt -- python3 $R/bench.py --workload scannet --pseudo-label 1 --steps 12 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep > $R/gpurun_out/${TAG}_pl_kt_bench.log 2>&1
DB=$(find /tmp/p/pl -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB 400 > $R/gpurun_out/${TAG}_pl_kernel_trace_stats_full.txt
head -3 $R/gpurun_out/${TAG}_pl_kernel_trace_stats_full.txt > $R/gpurun_out/${TAG}_pl_kernel_trace_stats.txt
grep -E "rg::|gp::|k_grid_radius|kg::|randint|distribution|softmax|max_kernel|Reduce|reduce|sort|Sort|CatArray|elementwise" $R/gpurun_out/${TAG}_pl_kernel_trace_stats_full.txt >> $R/gpurun_out/${TAG}_pl_kernel_trace_stats.txt
cat $R/gpurun_out/${TAG}_pl_kernel_trace_stats.txt | cut -c1-220
