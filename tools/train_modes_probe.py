"""One fixed batch, 30 SGD steps in every execution / precision mode of the step: loss at steps 0, 10, 20, 29 (sanity of the autocast
variants and of graph replay as TRAINING paths, not only as single steps).  usage: python tools/train_modes_probe.py [points] [lr]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import Geometry

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
batch = synthetic.make_batch([n, int(0.8 * n)], first_scene_id=40, device=dev)
geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()


def run(mode):
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16}.get(mode.split("-")[0])
    scale = 4096.0 if dtype == torch.float16 else 1.0
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=2)
    step.train()
    opt = engine.FusedSGD(step.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    cap = engine.CapturedStep(step, batch, geom=geom, autocast=dtype, loss_scale=scale) if mode.endswith("graph") else None
    losses = []
    for i in range(30):
        if cap is not None:
            out = cap(batch, geom)
        else:
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=dtype or torch.float16, enabled=dtype is not None):
                out = step(dict(batch, pdf_geometry=geom))
            (out["loss"] * scale).backward()
            if scale != 1.0:
                with torch.no_grad():
                    torch._foreach_mul_([p.grad for p in step.parameters() if p.grad is not None], 1.0 / scale)
        opt.step()
        losses.append(float(out["loss"]))
    engine.release_autograd_state(step)
    return losses


for mode in ("f32-eager", "f32-graph", "f16-eager", "f16-graph", "bf16-eager", "bf16-graph"):
    l = run(mode)
    print(f"{mode:11s}", " ".join(f"{l[i]:.4f}" for i in (0, 1, 5, 10, 20, 29)), flush=True)
