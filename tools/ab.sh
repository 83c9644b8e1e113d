# A/B of environment settings on the default bench schedule, interleaved: bash tools/ab.sh <tag> <rounds> "ENV=a" "ENV=b" ... [-- extra bench args]
TAG=$1; ROUNDS=$2; shift 2
SETS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done; [ "$1" == "--" ] && shift
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_ab.log; : > $OUT
for r in $(seq $ROUNDS); do for e in "${SETS[@]}"; do
  line=$(env $e python3 bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-ops-roofline --no-latency-sweep "$@" 2>>gpurun_out/${TAG}_ab.err | tail -1)
  echo "$e round $r: $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms/step host", round(d.get("host_enqueue_ms_per_step",0),2), "loss", d.get("loss"))')" >> $OUT
done; done
cat $OUT
