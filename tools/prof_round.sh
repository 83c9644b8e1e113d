# Round profile: default bench line, kernel trace summary (per kernel + per category), HBM traffic counters.
# usage (on the GPU box, via gpurun): bash tools/prof_round.sh <tag>
TAG=${1:-r01_c}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d /tmp/p/kt -o kt -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-ops-roofline > $R/gpurun_out/${TAG}_kt_bench.log 2>&1
DB=$(find /tmp/p/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB 70 > $R/gpurun_out/${TAG}_kernel_trace_stats.txt
python3 $R/tools/rocpd_categories.py $DB 18 > $R/gpurun_out/${TAG}_kernel_categories.txt; python3 $R/tools/rocpd_queues.py $DB 18 > $R/gpurun_out/${TAG}_kernel_streams.txt
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace -d /tmp/p/$c -o pm -- python3 $R/bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-ops-roofline > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1
python3 $R/tools/rocpd_pmc.py $(find /tmp/p/$c -name "*.db" | head -1) 400 > $R/gpurun_out/${TAG}_pmc_$c.txt
done
