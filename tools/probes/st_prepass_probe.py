"""Host side of config 5's coordinate-only pre-pass (StratifiedGeometry.precompute with the window tables and neighbour searches) for one
batch of 2 x 80k points on an otherwise idle device: wall time per batch, cProfile by function, torch-profiler op counts."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloudpdf_amd import engine, synthetic
dev = torch.device("cuda")
step = engine.OpenSegStep(backbone="ST-v1m1", loss_weight=0.008).to(dev)
bb = step.model.backbone
pool = [synthetic.make_batch([80000, 80000], first_scene_id=10 * i, device=dev) for i in range(3)]
def one(b):
    return bb.make_geometry(b["coord"], b["offset"], b["offset_host"]).precompute(bb.layers_by_level())
for b in pool: one(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in pool: one(b)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"pre-pass of one batch: host returns after {1e3 * (t1 - t0) / 3:.2f} ms, device done after {1e3 * (t2 - t0) / 3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for b in pool: one(b)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30); st.sort_stats("cumulative").print_stats(45)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one(pool[0]); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
