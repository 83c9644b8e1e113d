# kernel trace of config 4 with the pseudo-label pass in the step: bash tools/prof_pl.sh <tag> [extra bench args]
TAG=${1:-r04_pl}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p/pl
export PDFOPS_PL_WORKERS=1
rocprofv3 --kernel-trace --stats -d /tmp/p/pl -o pl -- python3 $R/bench.py --workload scannet --pseudo-label 1 --steps 12 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep "$@" > $R/gpurun_out/${TAG}_bench.log 2>&1
DB=$(find /tmp/p/pl -name "*.db" | head -1)
python3 $R/tools/rocpd_gaps.py $DB 8 14 > $R/gpurun_out/${TAG}_gaps.txt
python3 $R/tools/rocpd_stats.py $DB 40 > $R/gpurun_out/${TAG}_kernel_trace_stats.txt
tail -1 $R/gpurun_out/${TAG}_bench.log | cut -c1-300
cat $R/gpurun_out/${TAG}_gaps.txt
