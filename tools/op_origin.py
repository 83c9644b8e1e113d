"""torch profiler on the bench loop: which python lines the remaining aten kernels on the main stream (sum / add / mul / copy ...) come from."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(4): one(pool[i % 3], tk[i])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
for r in sorted([r for r in prof.key_averages(group_by_input_shape=True) if r.key.startswith("aten::") and r.self_device_time_total > 0], key=lambda r: -r.self_device_time_total)[:40]:
    print(f"{r.self_device_time_total / 4:9.1f} us/step  x{r.count / 4:5.1f}  {r.key:24s} {str(r.input_shapes)[:110]}")
rows = []
for r in sorted(rows, key=lambda r: -r.self_device_time_total)[:40]:
    st = [s for s in r.stack if "pointcloudpdf_amd" in s or "bench" in s or "tools/" in s][:3]
    print(f"{r.self_device_time_total / 4:9.1f} us/step  x{r.count / 4:5.1f}  {r.key:28s}", " <- ".join(s.split('/')[-1] for s in st))
