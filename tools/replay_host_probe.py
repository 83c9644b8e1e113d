"""Host time of one replayed training step by part (staging call, graph launch, gradient re-binding, optimizer): what the host must
sustain per step so that the device stays the bound.  usage: python tools/replay_host_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import Geometry

dev = torch.device("cuda", 0)
batch = synthetic.make_batch([100000, 100000], first_scene_id=1, device=dev)
geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
step = engine.OpenSegStep().to(dev); step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
cap = engine.CapturedStep(step, batch, geom=geom)
torch.cuda.synchronize()
parts = {"stage": 0.0, "replay": 0.0, "rebind": 0.0, "opt": 0.0}
N = 20
for it in range(N + 3):
    if it == 3:
        torch.cuda.synchronize(); parts = {k: 0.0 for k in parts}; t_all = time.perf_counter()
    if os.environ.get("SYNC_EACH"):
        torch.cuda.synchronize()   # pure host cost per part: nothing in the step can wait for the device
    t0 = time.perf_counter()
    cap.geometry.stage(geom, extra=[(batch[k], cap.static[k]) for k in cap.KEYS])
    t1 = time.perf_counter()
    cap.graph.replay()
    t2 = time.perf_counter()
    for p, g in zip(cap.params, cap.grads):
        p.grad = g
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    parts["stage"] += t1 - t0; parts["replay"] += t2 - t1; parts["rebind"] += t3 - t2; parts["opt"] += t4 - t3
host = time.perf_counter() - t_all
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
print({k: round(v / N * 1e3, 3) for k, v in parts.items()}, "host ms/step", round(host / N * 1e3, 3), "wall ms/step", round(wall / N * 1e3, 3))
