cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for extra in "--optimizer torch" "" "--prefetch 0"; do
i=$((i+1))
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p/pq$i -o pq -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-ops-roofline --no-latency-sweep $extra > $R/gpurun_out/pmc_probe_b$i.log 2>&1
echo "[$extra] rc=$?" >> $R/gpurun_out/pmc_probe.txt
done
