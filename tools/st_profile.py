"""torch profiler of one StratifiedTransformer training step (BASELINE config 5 shape): where the time goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
dev = torch.device("cuda")
step = engine.OpenSegStep(backbone="ST-v1m1", loss_weight=0.008).to(dev); step.train()
opt = torch.optim.SGD(step.parameters(), lr=1e-3, momentum=0.9)
b = synthetic.make_batch([80000, 80000], device=dev)
from pointcloudpdf_amd.stratified import StratifiedPrefetcher
pf = StratifiedPrefetcher(step.model.backbone)
def one():
    t = pf.submit(b)                       # (as bench.py --workload stratified: the coordinate-only work runs ahead on a worker thread)
    geom = pf.get(t)
    torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    out = step(dict(b, st_geometry=geom)); out["loss"].backward(); opt.step()
for _ in range(2): one()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=80))
