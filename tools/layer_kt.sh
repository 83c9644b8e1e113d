# per-kernel durations of the fused layer passes in isolation (tools/pt_layer_bench.py under a kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export REPS=5 LEVELS=${LEVELS:-0,1,2,3,4}
rm -rf /tmp/p/lk
rocprofv3 --kernel-trace --stats -d /tmp/p/lk -o lk -- python3 $R/tools/pt_layer_bench.py > /tmp/lk.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/p/lk -name "*.db" | head -1) 80 | grep -E "fl::k_|flm::k_|k_colsum|k_bn_finalize|k_seg" | cut -c1-120 > $R/gpurun_out/layer_kt.txt
