cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mc in ${MCS:-1}; do
export PDFOPS_MATRIX_CORE=$mc
rocprofv3 --kernel-trace -d /tmp/p/kt$mc -o kt -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/bench_mc$mc.log 2>&1
DB=$(find /tmp/p/kt$mc -name "*.db" | head -1); python3 $R/tools/rocpd_stats.py $DB 90 > $R/gpurun_out/kt_mc$mc.txt; python3 $R/tools/rocpd_categories.py $DB 6 > $R/gpurun_out/cat_mc$mc.txt
done
