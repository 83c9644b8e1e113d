import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher

dev = torch.device("cuda", 0)
n, steps = 100000, 6
batches = [synthetic.make_batch([n, n], first_scene_id=10 * i, device=dev) for i in range(3)]
pf = GeometryPrefetcher(depth=2)
variant = sys.argv[1]
step = engine.OpenSegStep().to(dev)
synthetic.fill_parameters_deterministic(step, seed=1)
step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
if "eagerfirst" in variant:
    tk = pf.submit_group([batches[i % 3] for i in range(3)])
    for i in range(3):
        b = batches[i % 3]; geom = pf.get(tk[i]); opt.zero_grad(set_to_none=True)
        out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=geom))
        out["loss"].backward(); opt.step()
    del out
cap = engine.CapturedStep(step, batches[0])
tickets = pf.submit_group([batches[i % 3] for i in range(steps)])
params = [p for p in step.parameters() if p.requires_grad]
L = []
for i in range(steps):
    b = batches[i % 3]
    geom = pf.get(tickets[i])
    out = cap(b, geom)
    if "sync" in variant:
        torch.cuda.synchronize()
    if "alloc" in variant:
        junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(64)]
        del junk
    if "gsum" in variant:
        gs = sum(p.grad.double().abs().sum() for p in params)
    opt.step()
    L.append(float(out["loss"]))
print(variant, " ".join(f"{v:.5f}" for v in L), flush=True)
