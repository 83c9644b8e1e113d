# kernel trace of the default bench schedule (per kernel, per category, per stream): bash tools/prof_kt.sh <tag> [extra bench args]
TAG=${1:-r03_x}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p/kt
rocprofv3 --kernel-trace --stats -d /tmp/p/kt -o kt -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep "$@" > $R/gpurun_out/${TAG}_kt_bench.log 2>&1
DB=$(find /tmp/p/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB 90 > $R/gpurun_out/${TAG}_kernel_trace_stats.txt
python3 $R/tools/rocpd_categories.py $DB 12 > $R/gpurun_out/${TAG}_kernel_categories.txt
python3 $R/tools/rocpd_queues.py $DB 12 16 > $R/gpurun_out/${TAG}_kernel_streams.txt
tail -1 $R/gpurun_out/${TAG}_kt_bench.log | cut -c1-400
