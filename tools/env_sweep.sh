# usage: bash tools/env_sweep.sh VAR v1 v2 ...   -> per value: bench under a kernel trace, category + selected kernel rows (gpurun_out/env_sweep.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VAR=$1; shift
for t in "$@"; do
export $VAR=$t
rm -rf /tmp/p/es$t
timeout 300 rocprofv3 --kernel-trace -d /tmp/p/es$t -o kt -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-latency-sweep --no-ops-roofline > /tmp/es$t.log 2>&1
echo "=== $VAR=$t" >> $R/gpurun_out/env_sweep.txt
python3 $R/tools/rocpd_categories.py $(find /tmp/p/es$t -name "*.db" | head -1) 10 | grep -E "${PAT_CAT:-pointwise|finalize}" >> $R/gpurun_out/env_sweep.txt
python3 $R/tools/rocpd_stats.py $(find /tmp/p/es$t -name "*.db" | head -1) 90 | grep -E "${PAT_K:-pw::|k_colsum|k_bn_finalize}" | cut -c1-100 >> $R/gpurun_out/env_sweep.txt
done
