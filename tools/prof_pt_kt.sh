cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export REPS=5
rocprofv3 --kernel-trace -d /tmp/p/ptk -o pt -- python3 $R/tools/pt_layer_bench.py > $R/gpurun_out/pt_kt.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/p/ptk -name "*.db" | head -1) 80 | grep -E "fl::|calls" > $R/gpurun_out/pt_kt.txt
