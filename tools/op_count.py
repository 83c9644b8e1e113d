"""torch profiler: op counts of one training step (which aten ops the step issues, how often)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import Geometry
dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = torch.optim.SGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4, fused=True)
b = synthetic.make_batch([100000, 100000], device=dev)
geom = Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
def one():
    opt.zero_grad(set_to_none=True)
    out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=geom))
    out["loss"].backward(); opt.step()
for _ in range(3): one()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=45, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=40, max_name_column_width=50, max_src_column_width=110))
