# end-of-round: default bench line, kernel trace tables, PMC passes + traffic file for the current kernel sources; usage: bash tools/final_check.sh <tag>
TAG=${1:-r02_k}
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 1200 bash tools/pmc_only.sh $TAG
DB=$(find /tmp/p/kt -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB 70 > gpurun_out/${TAG}_kernel_trace_stats.txt
python3 tools/rocpd_categories.py $DB 12 > gpurun_out/${TAG}_kernel_categories.txt; python3 tools/rocpd_queues.py $DB 12 16 > gpurun_out/${TAG}_kernel_streams.txt
