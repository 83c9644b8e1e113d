"""How far the reduced-precision products (torch.autocast -> fp16 / bfloat16 operands of the streaming Linear kernels) move logits and
gradients of one training step, by scene size, next to the path's own sensitivity (the same step with the features perturbed by 1e-6).
usage: python tools/amp_error.py [points_per_scene ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic

dev = torch.device("cuda", 0)
sizes = [int(v) for v in sys.argv[1:]] or [4500, 30000, 100000]


def run(batch, dtype=None, scale=1.0):
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=5)
    step.train()
    with torch.autocast("cuda", dtype=dtype or torch.float16, enabled=dtype is not None):
        out = step(dict(batch))
    (out["loss"] * scale).backward()
    logits = step.hooks["backbone"]["forward_output"].detach().clone()
    grads = {n: (p.grad.detach() / scale).double() for n, p in step.named_parameters() if p.grad is not None}
    engine.release_autograd_state(step)
    return float(out["loss"]), logits, grads


def l2(a, b, keys=None):
    keys = keys or list(a)
    num = sum(float((a[n] - b[n]).pow(2).sum()) for n in keys)
    den = sum(float(a[n].pow(2).sum()) for n in keys)
    return (num / max(den, 1e-300)) ** 0.5


for n in sizes:
    batch = synthetic.make_batch([n, int(n * 0.8)], first_scene_id=30, device=dev)
    loss0, lg0, g0 = run(batch)
    pert = dict(batch, feat=batch["feat"] * (1 + 1e-6 * torch.randn_like(batch["feat"])))
    rows = {}
    for name, (b, dt, sc) in {"f32 features * (1 + 1e-6 noise)": (pert, None, 1.0), "f16 operands, loss scale 4096": (batch, torch.float16, 4096.0),
                              "f16 operands, no loss scale": (batch, torch.float16, 1.0), "bf16 operands": (batch, torch.bfloat16, 1.0)}.items():
        loss, lg, g = run(b, dt, sc)
        groups = {}
        for k in g0:
            groups.setdefault(k.split(".")[2] if k.startswith("model.backbone.") else k.split(".")[0] + "." + k.split(".")[1], []).append(k)
        rows[name] = dict(loss=loss, logits_rel_max=float((lg - lg0).abs().max() / lg0.abs().max()), grad_rel_l2=l2(g0, g),
                          by_stage={s: round(l2(g0, g, ks), 4) for s, ks in sorted(groups.items())})
    print(json.dumps({"points": [n, int(n * 0.8)], "loss_f32": loss0, "runs": rows}))
