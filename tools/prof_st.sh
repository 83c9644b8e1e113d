# config 5 (StratifiedTransformer, 2 x 80k points): bench line + kernel trace tables; bash tools/prof_st.sh <tag>
TAG=${1:-r03_st}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload stratified > $R/gpurun_out/${TAG}_bench_stratified.json 2> $R/gpurun_out/${TAG}_bench_stratified.err
rm -rf /tmp/p/st
rocprofv3 --kernel-trace --stats -d /tmp/p/st -o kt -- python3 $R/bench.py --workload stratified --steps 6 --warmup 3 > $R/gpurun_out/${TAG}_stratified_kt_bench.log 2>&1
DB=$(find /tmp/p/st -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB 60 > $R/gpurun_out/${TAG}_stratified_kernel_trace_stats.txt
tail -1 $R/gpurun_out/${TAG}_bench_stratified.json | cut -c1-600
head -30 $R/gpurun_out/${TAG}_stratified_kernel_trace_stats.txt
python3 $R/tools/rocpd_queues.py $DB 6 14 > $R/gpurun_out/${TAG}_stratified_kernel_streams.txt 2>&1
