cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/p/knn -o k -- python3 $R/tools/knn_bench.py > $R/gpurun_out/knn_bench.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/p/knn -name "*.db" | head -1) 30 > $R/gpurun_out/knn_kernels.txt
