# round 5, first GPU call: single-XCD barrier probe + kernel trace of the default bench with the per-dispatch timeline of one step
TAG=${1:-r05_a}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 120 $R/tools/probes/bin/xcd_barrier_probe > $R/gpurun_out/${TAG}_xcd_barrier_probe.txt 2>&1
echo "probe rc $?" >> $R/gpurun_out/${TAG}_xcd_barrier_probe.txt
rm -rf /tmp/p/kt
rocprofv3 --kernel-trace --stats -d /tmp/p/kt -o kt -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep > $R/gpurun_out/${TAG}_kt_bench.log 2>&1
DB=$(find /tmp/p/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB 90 > $R/gpurun_out/${TAG}_kernel_trace_stats.txt
python3 $R/tools/rocpd_queues.py $DB 8 16 > $R/gpurun_out/${TAG}_kernel_streams.txt
python3 $R/tools/rocpd_timeline.py $DB 2 > $R/gpurun_out/${TAG}_timeline.txt
python3 $R/tools/stage_times.py > $R/gpurun_out/${TAG}_stage_times.txt 2>&1
tail -1 $R/gpurun_out/${TAG}_kt_bench.log | cut -c1-600
cat $R/gpurun_out/${TAG}_xcd_barrier_probe.txt
