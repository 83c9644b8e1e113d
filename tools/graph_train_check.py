"""Eager vs captured training loops on identical models / batches: loss per step (debug aid for engine.CapturedStep)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher

dev = torch.device("cuda", 0)
n = int(os.environ.get("POINTS", "20000"))
steps = int(os.environ.get("STEPS", "10"))
batches = [synthetic.make_batch([n, n], first_scene_id=10 * i, device=dev) for i in range(3)]


def make():
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)
    step.train()
    return step, engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)


pf = GeometryPrefetcher(depth=2)
losses = {}
for mode in ("eager", "graph", "mixed"):
    step, opt = make()
    cap = engine.CapturedStep(step, batches[0]) if mode != "eager" else None
    tickets = pf.submit_group([batches[i % 3] for i in range(steps)])
    out_l = []
    for i in range(steps):
        b = batches[i % 3]
        geom = pf.get(tickets[i])
        if cap is not None and not (mode == "mixed" and i % 4 == 0):
            out = cap(b, geom)
        else:
            opt.zero_grad(set_to_none=True)
            out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=geom))
            out["loss"].backward()
        opt.step()
        out_l.append(float(out["loss"]))
    losses[mode] = out_l
    print(mode, " ".join(f"{v:.5f}" for v in out_l), flush=True)
