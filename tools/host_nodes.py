"""Host time spent inside the forward / backward staticmethods of our autograd Functions, per training step (wall clock of the Python
call incl. the C launches it makes) -- the backward ones run on the autograd thread and are invisible to cProfile of the main thread."""
import os, sys, time, collections, inspect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pointcloudpdf_amd
from pointcloudpdf_amd import engine, synthetic, dense, segmentor, point_transformer, recognizer
from pointcloudpdf_amd.pointops import _ops
from pointcloudpdf_amd.geometry import Geometry
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(cls, name):
    fn = getattr(cls, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); d = time.perf_counter() - t
        e = acc[f"{cls.__name__}.{name}"]; e[0] += 1; e[1] += d
        return r
    setattr(cls, name, staticmethod(w))
for mod in (dense, segmentor, point_transformer, recognizer, _ops):
    for _, cls in inspect.getmembers(mod, inspect.isclass):
        if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls.__module__ == mod.__name__:
            wrap(cls, "forward"); wrap(cls, "backward")
dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
b = synthetic.make_batch([100000, 100000], device=dev)
geom = Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
def one():
    opt.zero_grad()
    t0 = time.perf_counter()
    out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=geom))
    t1 = time.perf_counter()
    out["loss"].backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(4): one()
torch.cuda.synchronize(); acc.clear()
N = 6
tot = [0.0, 0.0, 0.0]
for _ in range(N):
    for i, v in enumerate(one()): tot[i] += v
torch.cuda.synchronize()
print(f"host per step: forward {tot[0] / N * 1e3:.2f} ms, backward {tot[1] / N * 1e3:.2f} ms, optimizer {tot[2] / N * 1e3:.2f} ms")
fs = sum(v[1] for k, v in acc.items() if k.endswith(".forward")) / N * 1e3
bs = sum(v[1] for k, v in acc.items() if k.endswith(".backward")) / N * 1e3
print(f"inside our Function.forward: {fs:.2f} ms, inside our Function.backward: {bs:.2f} ms per step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44s} x{v[0] / N:5.1f}  {v[1] / N * 1e3:7.3f} ms/step  {v[1] / max(v[0], 1) * 1e6:7.1f} us/call")
