"""cProfile of one grouped pre-pass submission (host side) while a previous one may still be running."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher
pool = [synthetic.make_batch([100000, 100000], first_scene_id=10 * i, device="cuda") for i in range(3)]
pf = GeometryPrefetcher(depth=2)
group = [pool[j % 3] for j in range(6)]
for _ in range(2):
    t = pf.submit_group(group); torch.cuda.synchronize()
t0 = time.perf_counter(); t = pf.submit_group(group); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"submit host {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms")
pr = cProfile.Profile(); pr.enable(); t = pf.submit_group(group); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
