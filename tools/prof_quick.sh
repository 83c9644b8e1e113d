cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/p/q -o kt -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-latency-sweep --no-ops-roofline > $R/gpurun_out/q_bench.log 2>&1
DB=$(find /tmp/p/q -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB 60 > $R/gpurun_out/q_kernels.txt
python3 $R/tools/rocpd_categories.py $DB 18 > $R/gpurun_out/q_categories.txt; python3 $R/tools/rocpd_queues.py $DB 18 > $R/gpurun_out/q_queues.txt
