cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/p/rl -o rl -- python3 $R/tools/bench_rowlin.py > $R/gpurun_out/bench_rowlin.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/p/rl -name "*.db" | head -1) 80 > $R/gpurun_out/rl_kernels.txt
