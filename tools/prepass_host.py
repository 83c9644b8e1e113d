"""cProfile of one grouped geometry pre-pass submission (12 batches of 2 x 100k points): where its host time goes."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher
dev = torch.device("cuda")
pool = [synthetic.make_batch([100000, 100000], first_scene_id=10 * i, device=dev) for i in range(12)]
pf = GeometryPrefetcher(depth=2)
for _ in range(2):
    t = pf.submit_group(pool); [pf.get(x) for x in t]
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    t = pf.submit_group(pool)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(30); st.sort_stats("tottime").print_stats(25)
