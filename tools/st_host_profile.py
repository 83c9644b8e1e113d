"""cProfile + torch-profiler host view of one StratifiedTransformer training step (BASELINE config 5 shape, coordinate work prefetched):
which Python functions / torch ops the ~38 ms of host enqueue time per step go to."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.stratified import StratifiedPrefetcher
dev = torch.device("cuda")
step = engine.OpenSegStep(backbone="ST-v1m1", loss_weight=0.008).to(dev); step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
b = synthetic.make_batch([80000, 80000], device=dev)
pf = StratifiedPrefetcher(step.model.backbone)
geom = pf.get(pf.submit(b))
torch.cuda.synchronize()
def one():
    opt.zero_grad()
    out = step(dict(b, st_geometry=geom)); out["loss"].backward(); opt.step()
for _ in range(3): one()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): one()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 5:.2f} ms per step, drained after {1e3 * (t2 - t0) / 5:.2f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(3): one()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(60)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=50, max_name_column_width=70))
