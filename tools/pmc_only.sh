# the two PMC passes + traffic file of tools/prof_round.sh alone (kernel trace taken separately); usage: bash tools/pmc_only.sh <tag>
TAG=${1:-r02_i}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/p/kt -o kt -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep > $R/gpurun_out/${TAG}_kt_bench.log 2>&1
DB=$(find /tmp/p/kt -name "*.db" | head -1)
for c in FETCH_SIZE WRITE_SIZE; do
for attempt in 1 2 3 4; do
rm -rf /tmp/p/$c
timeout -s KILL 240 rocprofv3 --pmc $c --kernel-trace -d /tmp/p/$c -o pm -- python3 $R/bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep --throttle > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1 && break
echo "pmc $c attempt $attempt failed" >> $R/gpurun_out/${TAG}_pmc_retries.log
done
python3 $R/tools/rocpd_pmc.py $(find /tmp/p/$c -name "*.db" | head -1) 400 > $R/gpurun_out/${TAG}_pmc_$c.txt
done
python3 $R/tools/traffic_json.py $DB $(find /tmp/p/FETCH_SIZE -name "*.db" | head -1) $(find /tmp/p/WRITE_SIZE -name "*.db" | head -1) 12 $R/gpurun_out/${TAG}_traffic.json > $R/gpurun_out/${TAG}_traffic.log 2>&1
