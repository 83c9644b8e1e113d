# Everything the bench line and DESIGN.md cite for one state of the kernel sources: default bench line, kernel trace tables (per kernel / per
# category / per stream), PMC traffic (FETCH_SIZE / WRITE_SIZE passes -> <tag>_traffic.json), SQ counter passes (-> <tag>_sq.txt / <tag>_sq.json).
# Usage on the GPU box: bash tools/prof_all.sh <tag>      (then copy gpurun_out/<tag>_* into profiles/)
TAG=${1:-r04_x}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/final_check.sh $TAG
bash tools/prof_sq.sh $TAG
DB=$(find /tmp/p/kt -name "*.db" | head -1)
python3 tools/rocpd_sq.py --json gpurun_out/${TAG}_sq.json $DB 12 gpurun_out/${TAG}_sq1.txt gpurun_out/${TAG}_sq2.txt gpurun_out/${TAG}_sq3.txt gpurun_out/${TAG}_sq4.txt > gpurun_out/${TAG}_sq_json.log 2>&1
