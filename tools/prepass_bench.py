"""Time of one grouped geometry pre-pass (12 scenes) on the GPU: total and the fused-TransitionDown tables alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import Geometry
b = synthetic.make_batch([100000] * 12, first_scene_id=3, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    keys = [k for k in g._memo if k[0] == "td"]
    for k in keys: del g._memo[k]
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for k in keys: g.td(k[1], k[2], k[3])
    torch.cuda.synchronize(); t3 = time.perf_counter()
    parts = g.split([2] * 6)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"precompute {1e3 * (t1 - t0):8.1f} ms   td tables alone {1e3 * (t3 - t2):8.1f} ms   split {1e3 * (t4 - t3):8.1f} ms")
