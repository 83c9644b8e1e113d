"""Throughput probe: one geometry pre-pass per GROUP of G batches (2G scenes in one FPS launch) on a side stream,
while the model trains on a static geometry (results of the grouped pre-pass are only waited on, not used)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import Geometry
dev = torch.device("cuda")
G = int(os.environ.get("G", "4")); STEPS = int(os.environ.get("STEPS", "24"))
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = torch.optim.SGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
b = synthetic.make_batch([100000, 100000], device=dev)
big = synthetic.make_batch([100000] * (2 * G), first_scene_id=50, device=dev)
geom = Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
side = [torch.cuda.Stream(), torch.cuda.Stream()]
def submit(k):
    s = side[k % 2]
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = Geometry(big["coord"], big["offset"], big["offset_host"]).precompute()
        ev = torch.cuda.Event(); ev.record(s)
    return g, ev
def one():
    opt.zero_grad(set_to_none=True)
    out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=geom))
    out["loss"].backward(); opt.step()
pending = [submit(0), submit(1)]
for _ in range(2 * G): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
k = 2
for i in range(STEPS):
    if i % G == 0:
        g, ev = pending.pop(0)
        torch.cuda.current_stream().wait_event(ev)   # this group's tables are needed from here on
        pending.append(submit(k)); k += 1
    one()
torch.cuda.synchronize()
print(f"G={G}: {(time.perf_counter() - t0) / STEPS * 1e3:.2f} ms/step")
