#!/usr/bin/env python3
"""GPU probe (library built with PDFOPS_FPS_PROFILE=1): cycles per phase of k_fps_multi at level 1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import _native, synthetic
be = _native.hip_backend(); be.collect_fps_stats = True
batch = synthetic.make_batch([100000] * 2, device="cuda")
noff = torch.tensor([25000, 50000], dtype=torch.int32, device="cuda")
for k, nw, mw in (("8", "8", "0"), ("8", "8", "1")):
    os.environ["PDFOPS_FPS_K"] = k; os.environ["PDFOPS_FPS_NW"] = nw; os.environ["PDFOPS_FPS_MW"] = mw
    for _ in range(2):
        be.farthest_point_sampling(batch["coord"], batch["offset"], noff, 100000, 50000)
    torch.cuda.synchronize()
    print(f"K={k} NW={nw} MW={mw} kilo-cycles [candidates, lists, updates, refresh]:", be.last_fps_stats.tolist())
