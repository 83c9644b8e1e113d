"""Where the host is while the replayed training loop runs (2 x 100k points, look-ahead groups as bench.py): host time per step spent in
graph.replay(), in the staging launch, in the optimizer, in the loader -- next to the wall time per step.  Usage: python tools/probes/step_host_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from pointcloudpdf_amd import engine, synthetic  # noqa: E402
from pointcloudpdf_amd.geometry import StaticGeometry  # noqa: E402

dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev)
synthetic.fill_parameters_deterministic(step, seed=1)
step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
train = engine.TrainStep(step, opt, graph=True)
pool = [synthetic.make_batch([100000, 100000], first_scene_id=10 * i, device=dev) for i in range(4)]
keys = ("coord", "feat", "offset", "offset_host", "segment")
group = int(os.environ.get("PROBE_GROUP", "24"))
steps, warm = 48, 24


def stream():
    i = 0
    while True:
        yield {k: pool[i % 4][k] for k in keys}
        i += 1


acc = {"replay": 0.0, "stage": 0.0, "optimizer": 0.0, "loader": 0.0, "n": 0}
orig_replay = torch.cuda.CUDAGraph.replay
orig_stage = StaticGeometry.stage
orig_step = engine.FusedSGD.step


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[name] += time.perf_counter() - t
        return r
    return w


torch.cuda.CUDAGraph.replay = timed("replay", orig_replay)
StaticGeometry.stage = timed("stage", orig_stage)
engine.FusedSGD.step = timed("optimizer", orig_step)
it = iter(engine.GroupedGeometryLoader(stream(), group=group))
for _ in range(warm):
    train(next(it))
torch.cuda.synchronize()
for k in acc:
    acc[k] = 0.0
t0 = time.perf_counter()
for _ in range(steps):
    t = time.perf_counter()
    b = next(it)
    acc["loader"] += time.perf_counter() - t
    train(b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
if os.environ.get("PROBE_EVENTS"):
    # device-side view of the step boundaries, without a profiler: events after the replay, after the optimizer launch and after the next
    # staging launch -> the time between the end of the graph and the end of k_sgd (34 us of kernel) and between k_sgd and the end of
    # the two staging kernels (69 us)
    ev = {"replay": [], "optimizer": [], "stage": []}

    def marked(name, fn):
        def w(*a, **k):
            r = fn(*a, **k)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev[name].append(e)
            return r
        return w

    torch.cuda.CUDAGraph.replay = marked("replay", orig_replay)
    StaticGeometry.stage = marked("stage", orig_stage)
    engine.FusedSGD.step = marked("optimizer", orig_step)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for _ in range(steps):
        train(next(it))
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    n = len(ev["replay"])
    g_end_to_sgd = sorted(ev["replay"][i].elapsed_time(ev["optimizer"][i]) for i in range(n))
    sgd_to_stage = sorted(ev["optimizer"][i].elapsed_time(ev["stage"][i + 1]) for i in range(n - 1))
    stage_to_graph_end = sorted(ev["stage"][i].elapsed_time(ev["replay"][i]) for i in range(n))
    print(f"with events: wall {1e3 * (t4 - t3) / steps:.2f} ms per step; median graph end -> k_sgd end {1e3 * g_end_to_sgd[n // 2]:.0f} us (kernel 34), "
          f"k_sgd end -> staging end {1e3 * sgd_to_stage[n // 2]:.0f} us (kernels 69), staging end -> graph end {stage_to_graph_end[n // 2]:.3f} ms", flush=True)
print(f"wall {1e3 * (t2 - t0) / steps:.2f} ms per step; host enqueue {1e3 * (t1 - t0) / steps:.2f}; of it: " +
      ", ".join(f"{k} {1e3 * acc[k] / steps:.2f}" for k in ("loader", "stage", "replay", "optimizer")) +
      f", rest {1e3 * ((t1 - t0) - sum(acc[k] for k in ('loader', 'stage', 'replay', 'optimizer'))) / steps:.2f} ms", flush=True)
