"""torch profiler on the bench loop: aten ops per step by python origin (count, host time, device time) -- where the non-pdfops launches come from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher
dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4) if hasattr(engine, "FusedSGD") else torch.optim.SGD(step.parameters(), lr=1e-3)
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
    for i in range(4): one(pool[i % 3], tk[i])
    torch.cuda.synchronize()
rows = [r for r in prof.key_averages(group_by_stack_n=12) if r.key.startswith("aten::") and r.device_time_total > 0 and r.self_device_time_total > 0]
agg = {}
for r in rows:
    st = [s for s in r.stack if ("pointcloudpdf_amd" in s or "bench" in s or "tools/" in s)]
    origin = " <- ".join(s.split("/")[-1].strip() for s in st[:2]) or "(autograd / torch internals)"
    k = (r.key, origin)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += r.count; a[1] += r.self_cpu_time_total; a[2] += r.self_device_time_total
tot = [0, 0.0, 0.0]
for (key, origin), a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"x{a[0] / 4:6.1f}/step  host {a[1] / 4:7.1f} us  dev {a[2] / 4:7.1f} us  {key:26s} {origin[:150]}")
for a in agg.values():
    for i in range(3): tot[i] += a[i]
print(f"total aten ops with device work: {tot[0] / 4:.1f}/step, host self {tot[1] / 4:.1f} us, device {tot[2] / 4:.1f} us")
