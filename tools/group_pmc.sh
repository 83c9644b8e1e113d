R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/group_probe.py > $R/gpurun_out/group_probe.log 2>&1 || exit 0
for c in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
n=$(echo $c | tr ' ' '_')
rm -rf /tmp/p/g
timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/p/g -o pm -- python3 $R/tools/group_probe.py > /dev/null 2>&1
python3 $R/tools/rocpd_pmc.py $(find /tmp/p/g -name "*.db" | head -1) 400 2>&1 | grep -i "group\|counter\|kernel" > $R/gpurun_out/group_pmc_$n.txt
done
