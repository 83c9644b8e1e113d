# Round profile: default bench line, kernel trace summary (per kernel + per category + per stream), HBM traffic counters and the
# traffic file bench.py loads (profiles/<round>_traffic.json).
# usage (on the GPU box, via gpurun): bash tools/prof_round.sh <tag>      e.g. r02_a
TAG=${1:-r02_a}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d /tmp/p/kt -o kt -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep > $R/gpurun_out/${TAG}_kt_bench.log 2>&1
DB=$(find /tmp/p/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB 70 > $R/gpurun_out/${TAG}_kernel_trace_stats.txt
python3 $R/tools/rocpd_categories.py $DB 12 > $R/gpurun_out/${TAG}_kernel_categories.txt; python3 $R/tools/rocpd_queues.py $DB 12 16 > $R/gpurun_out/${TAG}_kernel_streams.txt
for c in FETCH_SIZE WRITE_SIZE; do
# (rocprofv3 --pmc occasionally dies with SIGSEGV inside its dispatch interception on this many-kernel, three-thread process -- seen
#  1 run in 3, never without --pmc -- or hangs after an 'AQL packet is malformed' abort: bounded by `timeout`, retried; the counters come
#  from a complete run)
for attempt in 1 2 3 4; do
rm -rf /tmp/p/$c
timeout -s KILL 240 rocprofv3 --pmc $c --kernel-trace -d /tmp/p/$c -o pm -- python3 $R/bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep --throttle > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1 && break
echo "pmc $c attempt $attempt failed" >> $R/gpurun_out/${TAG}_pmc_retries.log
done
python3 $R/tools/rocpd_pmc.py $(find /tmp/p/$c -name "*.db" | head -1) 400 > $R/gpurun_out/${TAG}_pmc_$c.txt
done
python3 $R/tools/traffic_json.py $DB $(find /tmp/p/FETCH_SIZE -name "*.db" | head -1) $(find /tmp/p/WRITE_SIZE -name "*.db" | head -1) 12 $R/gpurun_out/${TAG}_traffic.json > $R/gpurun_out/${TAG}_traffic.log 2>&1
