"""Debug aid: eager / graph / mixed training loops with per-step checksums of gradients, parameters and momentum buffers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher

dev = torch.device("cuda", 0)
n = int(os.environ.get("POINTS", "100000"))
steps = int(os.environ.get("STEPS", "12"))
eager_at = [int(v) for v in os.environ.get("EAGER_AT", "8").split(",") if v]
batches = [synthetic.make_batch([n, n], first_scene_id=10 * i, device=dev) for i in range(3)]
pf = GeometryPrefetcher(depth=2)


def run(mode):
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)
    step.train()
    opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    cap = engine.CapturedStep(step, batches[0]) if mode != "eager" else None
    tickets = pf.submit_group([batches[i % 3] for i in range(steps)])
    params = [p for p in step.parameters() if p.requires_grad]
    rows = []
    for i in range(steps):
        b = batches[i % 3]
        geom = pf.get(tickets[i])
        if cap is not None and not (mode == "mixed" and i in eager_at):
            out = cap(b, geom)
        else:
            opt.zero_grad(set_to_none=True)
            out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=geom))
            out["loss"].backward()
        gs = float(sum(p.grad.double().abs().sum() for p in params))
        nonc = sum(1 for p in params if not p.grad.is_contiguous())
        opt.step()
        ps = float(sum(p.detach().double().abs().sum() for p in params))
        ms = float(sum(opt.state[p]["momentum_buffer"].double().abs().sum() for p in params))
        rows.append((float(out["loss"]), gs, ps, ms, nonc))
    return rows


res = {m: run(m) for m in ("eager", "graph", "mixed")}
for i in range(steps):
    print(i, " | ".join(f"{m}: L {res[m][i][0]:.5f} g {res[m][i][1]:.6e} p {res[m][i][2]:.8e} m {res[m][i][3]:.6e} nc {res[m][i][4]}" for m in res), flush=True)
