"""Host enqueue time vs GPU time of the forward / backward / optimizer phases of one step (geometry precomputed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import Geometry
dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = torch.optim.SGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4, fused=True)
b = synthetic.make_batch([100000, 100000], device=dev)
geom = Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
torch.cuda.synchronize()
def one(rec=None):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t = [time.perf_counter()]
    ev[0].record()
    opt.zero_grad(set_to_none=True)
    out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=geom))
    t.append(time.perf_counter()); ev[1].record()
    out["loss"].backward()
    t.append(time.perf_counter()); ev[2].record()
    opt.step()
    t.append(time.perf_counter()); ev[3].record()
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    if rec is not None:
        rec.append(([1e3 * (t[i + 1] - t[i]) for i in range(4)], [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]))
for _ in range(3): one()
rec = []
for _ in range(5): one(rec)
import numpy as np
h = np.mean([r[0] for r in rec], 0); g = np.mean([r[1] for r in rec], 0)
print(f"host enqueue ms: fwd {h[0]:.1f}  bwd {h[1]:.1f}  opt {h[2]:.1f}  final-sync wait {h[3]:.1f}   total {sum(h):.1f}")
print(f"gpu elapsed  ms: fwd {g[0]:.1f}  bwd {g[1]:.1f}  opt {g[2]:.1f}   total {sum(g):.1f}")
