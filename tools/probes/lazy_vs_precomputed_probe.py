"""One eager training step of OpenSegStep: coordinate tables computed inside the forward (no ``pdf_geometry``) against the tables
attached up front (Geometry.precompute).  Prints per-parameter gradient differences (largest first)."""
import sys
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import Geometry

dev = torch.device("cuda", 0)
sizes = [int(v) for v in sys.argv[1:]] or [2400, 2000]
b0 = synthetic.make_batch(sizes, first_scene_id=80, device=dev)


def run(pre):
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=9)
    step.train()
    b = {k: b0[k] for k in ("coord", "feat", "offset", "offset_host", "segment")}
    if pre:
        b["pdf_geometry"] = Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
    out = step(b)
    out["loss"].backward()
    g = {n: p.grad.detach().clone() for n, p in step.named_parameters() if p.grad is not None}
    engine.release_autograd_state(step)
    return float(out["loss"]), float(out["model_loss"]), float(out["recognizer_loss"]), g


a, b, c = run(False), run(True), run(False)
print("losses lazy", a[:3], "pre", b[:3])
print("lazy vs lazy max diff", max(float((a[3][n] - c[3][n]).abs().max()) for n in a[3]))
rows = sorted(((float((a[3][n] - b[3][n]).abs().max()), float(a[3][n].abs().max()), n) for n in a[3]), reverse=True)
for d, m, n in rows[:25]:
    print(f"{d:.3e}  |g|max {m:.3e}  {n}")
print("params with any diff:", sum(1 for d, _, _ in rows if d > 0), "of", len(rows))
