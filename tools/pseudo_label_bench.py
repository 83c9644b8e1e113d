#!/usr/bin/env python3
"""GPU probe: the pseudo-label pass on 2 ScanNet-shaped scenes of 100k points (radius query, region growing, host pruning)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudpdf_amd import pseudo_label, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
b = synthetic.make_batch([n, n], first_scene_id=70, kind="scannet", device="cuda")
coord, off = b["coord"], b["offset"]
g = torch.Generator(device="cuda").manual_seed(0)
centre = coord[torch.randint(0, n, (1,), device="cuda", generator=g)]
conf = 6.0 * torch.sigmoid((torch.norm(coord - centre, dim=-1) - 0.8) * 4.0) + 0.3 * torch.randn(2 * n, device="cuda", generator=g)
logits = 0.2 * torch.randn(2 * n, 20, device="cuda", generator=g)
logits[torch.arange(2 * n, device="cuda"), (coord[:, 0] * 3).long() % 20] += conf
for _ in range(2):
    nn = pseudo_label.radius_neighbors(coord, off, 0.1, 64)
torch.cuda.synchronize(); t0 = time.perf_counter()
nn = pseudo_label.radius_neighbors(coord, off, 0.1, 64)
torch.cuda.synchronize(); t_nn = time.perf_counter() - t0
np.random.seed(0)
t0 = time.perf_counter()
mask = pseudo_label.get_pseudo_mask(coord, logits, off, neighbors=nn, generator=torch.Generator().manual_seed(0))
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"2 x {n} points: radius neighbours (64 within 0.1 m) {t_nn * 1e3:.2f} ms; region growing + pruning {t_all * 1e3:.1f} ms; masked {int(mask.sum())} points")
