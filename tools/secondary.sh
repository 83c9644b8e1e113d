# secondary workloads of the bench (one JSON each); usage: bash tools/secondary.sh <tag>
TAG=${1:-r02_k}
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-ops-roofline --no-latency-sweep"
timeout 600 python bench.py $X --workload scannet > gpurun_out/${TAG}_bench_scannet.json 2> gpurun_out/sec_scannet.err
timeout 900 python bench.py $X --workload scannet --pseudo-label 1 > gpurun_out/${TAG}_bench_scannet_pseudo_label.json 2> gpurun_out/sec_pl.err
timeout 600 python bench.py $X --jitter 0.1 > gpurun_out/${TAG}_bench_jitter.json 2> gpurun_out/sec_jitter.err
timeout 600 python bench.py $X --scenes 4 > gpurun_out/${TAG}_bench_bs4.json 2> gpurun_out/sec_bs4.err
timeout 900 python bench.py --workload stratified > gpurun_out/${TAG}_bench_stratified.json 2> gpurun_out/sec_st.err
timeout 600 python bench.py $X --optimizer torch > gpurun_out/${TAG}_bench_torch_sgd.json 2> gpurun_out/sec_tsgd.err
timeout 600 python bench.py $X --jitter 0.1 --size-classes 4 --pool 8 > gpurun_out/${TAG}_bench_jitter_4classes.json 2> gpurun_out/sec_j4.err
rm -f gpurun_out/${TAG}_secondary.txt
for f in scannet scannet_pseudo_label jitter jitter_4classes bs4 stratified torch_sgd; do python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/${TAG}_bench_$f.json')); print('$f', round(d['ms_per_step'],2), round(d['value']/1e6,2), 'M points/s')
except Exception as e: print('$f FAILED', e)
" >> gpurun_out/${TAG}_secondary.txt; done
timeout 600 python bench.py $X --amp f16 > gpurun_out/${TAG}_bench_amp_f16.json 2> gpurun_out/sec_f16.err
timeout 600 python bench.py $X --amp bf16 > gpurun_out/${TAG}_bench_amp_bf16.json 2> gpurun_out/sec_bf16.err
for f in amp_f16 amp_bf16; do python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/${TAG}_bench_$f.json')); print('$f', round(d['ms_per_step'],2), round(d['value']/1e6,2), 'M points/s')
except Exception as e: print('$f FAILED', e)
" >> gpurun_out/${TAG}_secondary.txt; done
