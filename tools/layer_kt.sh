# per-kernel times of the fused PointTransformerLayer alone (tools/pt_layer_bench.py under a kernel trace), one table per environment setting:
#   bash tools/layer_kt.sh <tag> "ENV=a" "ENV=b" ...     (LEVELS / REPS as for pt_layer_bench.py; FILTER = regex of kernel names, default the layer passes)
TAG=${1:-r06_x}; shift
R=$GRAFT_REPO_ROOT
FILTER=${FILTER:-"k_p[1-4]|k_b[1-4]|k_colsum|k_bn_finalize|k_seg"}
cd /tmp && export TMPDIR=/tmp && mkdir -p /tmp/p
OUT=$R/gpurun_out/${TAG}_layer_kt.txt
: > $OUT
i=0
for SETTING in "$@"; do
  i=$((i+1))
  rm -rf /tmp/p/lkt$i
  ( export $SETTING; rocprofv3 --kernel-trace --stats -d /tmp/p/lkt$i -o kt -- python3 $R/tools/pt_layer_bench.py > /tmp/p/lkt$i.log 2>&1 )
  DB=$(find /tmp/p/lkt$i -name "*.db" | head -1)
  echo "=== $SETTING" >> $OUT
  grep level /tmp/p/lkt$i.log >> $OUT
  python3 $R/tools/rocpd_stats.py $DB 400 | grep -E "$FILTER|total kernel" | cut -c1-150 >> $OUT
done
cat $OUT
