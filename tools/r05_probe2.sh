# round 5, second GPU call: graph fork probe + the new stream-safety test
TAG=${1:-r05_b}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 120 $R/tools/probes/bin/graph_fork_probe > $R/gpurun_out/${TAG}_graph_fork_probe.txt 2>&1
echo "probe rc $?" >> $R/gpurun_out/${TAG}_graph_fork_probe.txt
cd $R && timeout 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "loader or private_stream" > $R/gpurun_out/${TAG}_tests.log 2>&1
tail -5 $R/gpurun_out/${TAG}_tests.log
cat $R/gpurun_out/${TAG}_graph_fork_probe.txt
