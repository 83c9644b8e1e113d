"""cProfile of the host side of one training step (geometry precomputed): where the ~30 ms of Python / launch time go."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import Geometry
dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
b = synthetic.make_batch([100000, 100000], device=dev)
geom = Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
def one():
    opt.zero_grad()
    out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"], pdf_geometry=geom))
    out["loss"].backward(); opt.step()
for _ in range(3): one()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): one()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(40)
