# full GPU suite + smoke, then tools/final_check.sh <tag> (bench line, kernel trace tables, PMC passes + traffic file); usage: bash tools/round_end.sh <tag>
TAG=${1:-r02_p}
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" > gpurun_out/${TAG}_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/${TAG}_smoke.log
bash tools/final_check.sh $TAG
