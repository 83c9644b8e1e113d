# config 4 with the pseudo-label pass (2 x 150k ScanNet-shaped points; the pass inside the step's one graph): kernel trace table of the
# pass's kernels + busy / idle time of the device timeline; bash tools/prof_pl.sh <tag> [extra bench args]
TAG=${1:-r05_pl}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p/pl
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p/pl -o kt -- python3 $R/bench.py --workload scannet --pseudo-label 1 --steps 12 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep "$@" > $R/gpurun_out/${TAG}_pl_kt_bench.log 2>&1
DB=$(find /tmp/p/pl -name "*.db" | head -1)
python3 $R/tools/rocpd_gaps.py $DB 8 14 > $R/gpurun_out/${TAG}_pl_gaps.txt 2>&1
python3 $R/tools/rocpd_stats.py $DB 400 > $R/gpurun_out/${TAG}_pl_kernel_trace_stats_full.txt
head -1 $R/gpurun_out/${TAG}_pl_kernel_trace_stats_full.txt > $R/gpurun_out/${TAG}_pl_kernel_trace_stats.txt
grep -E "rg::|gp::|k_grid_radius|k_zero_words|random_from_to|softmax|MaxOps" $R/gpurun_out/${TAG}_pl_kernel_trace_stats_full.txt >> $R/gpurun_out/${TAG}_pl_kernel_trace_stats.txt
tail -1 $R/gpurun_out/${TAG}_pl_kernel_trace_stats_full.txt >> $R/gpurun_out/${TAG}_pl_kernel_trace_stats.txt
cut -c1-200 $R/gpurun_out/${TAG}_pl_kernel_trace_stats.txt
