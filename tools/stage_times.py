"""Where the step's GPU time goes by model stage: CUDA events on forward / backward hooks of the top-level stages (one warm step is
timed; hooks synchronise nothing).  Forward time of stage s = event(post s) - event(pre s); backward likewise from the backward hooks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher
dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
pool = [synthetic.make_batch([100000, 100000], first_scene_id=10 * i, device=dev) for i in range(3)]
pf = GeometryPrefetcher(depth=2)
ev = {}
def mk(name, kind):
    def f(*a):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault((name, kind), []).append(e)
    return f
stages = {}
bb = step.model.backbone
for n in ["enc1", "enc2", "enc3", "enc4", "enc5", "dec5", "dec4", "dec3", "dec2", "dec1", "cls"]:
    mod = getattr(bb, n)
    if n.startswith("enc") or n.startswith("dec"):
        for j, sub in enumerate(mod):
            stages[f"seg.{n}.{j}:{type(sub).__name__}"] = sub
    else:
        stages["seg." + n] = mod
rec = step.recognizer
for n, m in rec.named_modules():
    if n and n.count(".") <= 1 and not isinstance(m, (torch.nn.ModuleList, torch.nn.Sequential)) or n.count(".") == 1:
        stages["rec." + n] = m
for name, m in stages.items():
    m.register_forward_pre_hook(mk(name, "f0")); m.register_forward_hook(mk(name, "f1"))
    m.register_full_backward_pre_hook(mk(name, "b0")); m.register_full_backward_hook(mk(name, "b1"))
def one(b, t, timed=False):
    opt.zero_grad()
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=pf.get(t)))
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    out["loss"].backward()
    e2 = torch.cuda.Event(enable_timing=True); e2.record()
    opt.step()
    e3 = torch.cuda.Event(enable_timing=True); e3.record()
    return e0, e1, e2, e3
tk = pf.submit_group([pool[i % 3] for i in range(6)])
for i in range(5): one(pool[i % 3], tk[i])
torch.cuda.synchronize(); ev.clear()
e = one(pool[5 % 3], tk[5]); torch.cuda.synchronize()
print(f"forward {e[0].elapsed_time(e[1]):.2f} ms  backward {e[1].elapsed_time(e[2]):.2f} ms  optimizer {e[2].elapsed_time(e[3]):.3f} ms")
tf = tb = 0.0
for name in stages:
    f = ev[(name, "f0")][0].elapsed_time(ev[(name, "f1")][0]) if (name, "f0") in ev and (name, "f1") in ev else float("nan")
    b = ev[(name, "b0")][0].elapsed_time(ev[(name, "b1")][0]) if (name, "b0") in ev and (name, "b1") in ev else float("nan")
    tf += 0 if f != f else f; tb += 0 if b != b else b
    print(f"{name:40s} fwd {f:7.3f} ms   bwd {b:7.3f} ms")
print(f"sum of stages: fwd {tf:.2f} bwd {tb:.2f}")
