"""Config 5 (StratifiedTransformer, 2 x 80k points) through engine.GroupedGeometryLoader + engine.TrainStep exactly as bench.py runs it,
with the host time of every step split into: waiting for the loader (pre-pass worker not done), the trainer call (enqueue), and -- from a
second run with the pre-pass of ALL batches done before the loop -- the same without a worker thread beside the step.
    python tools/probes/st_loop_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.stratified import StratifiedPrefetcher

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda")
step = engine.OpenSegStep(backbone="ST-v1m1", loss_weight=0.008).to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
trainer = engine.TrainStep(step, opt, graph=False)
pool = [synthetic.make_batch([80000, 80000], first_scene_id=10 * i, device=dev) for i in range(4)]

def stream():
    i = 0
    while True:
        b = pool[i % len(pool)]
        yield dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"])
        i += 1

pf = StratifiedPrefetcher(step.model.backbone)

def run(loader_iter, n, label):
    wait = call = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); b = next(loader_iter); c = time.perf_counter(); trainer(b); d = time.perf_counter()
        wait += c - a; call += d - c
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{label}: {1e3 * (t1 - t0) / n:.2f} ms per step; host: loader wait {1e3 * wait / n:.2f}, trainer call {1e3 * call / n:.2f}", flush=True)

it = iter(engine.GroupedGeometryLoader(stream(), group=3, prefetcher=pf, key="st_geometry", submit_delay=0))
run(it, 12, "warm-up (look-ahead groups of 3)")
run(it, steps, "look-ahead groups of 3, worker thread beside the step")
# every batch's pre-pass done ahead of the loop: the step alone
src = stream()
ready = []
for _ in range(len(pool)):
    b = next(src)
    b["st_geometry"] = pf.get(pf.submit(b))
    ready.append(b)
torch.cuda.synchronize()
def cycle():
    i = 0
    while True:
        yield dict(ready[i % len(ready)]); i += 1
it2 = cycle()
run(it2, 8, "warm-up (pre-pass done ahead)")
run(it2, steps, "pre-pass done ahead of the loop (no worker thread, no side stream)")
pf.close()
