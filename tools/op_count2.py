"""torch profiler on the bench loop itself (grouped pre-pass, split geometry): where the D2D copies come from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher
dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = torch.optim.SGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4, fused=True)
pool = [synthetic.make_batch([100000, 100000], first_scene_id=10 * i, device=dev) for i in range(3)]
pf = GeometryPrefetcher(depth=2)
def one(b, t):
    opt.zero_grad(set_to_none=True)
    out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=pf.get(t)))
    out["loss"].backward(); opt.step()
tk = pf.submit_group([pool[i % 3] for i in range(4)])
for i in range(4): one(pool[i % 3], tk[i])
torch.cuda.synchronize()
tk = pf.submit_group([pool[i % 3] for i in range(4)])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tk2 = pf.submit_group([pool[i % 3] for i in range(4)])
    for i in range(4): one(pool[i % 3], tk[i])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=30, max_name_column_width=70))
rows = [r for r in prof.key_averages(group_by_stack_n=8) if ("copy" in r.key.lower() or "Memcpy" in r.key or "clone" in r.key or "contiguous" in r.key) and r.count >= 4]
for r in sorted(rows, key=lambda r: -r.count)[:12]:
    print(r.count, r.key, "\n      ", "\n       ".join(r.stack[:8]))
