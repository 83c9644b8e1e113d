"""Phase times of pdf_graph_forest's rounds (build with PDFOPS_EXTRA_FLAGS="graph_prune.hip:-DGP_PROFILE"): a region graph like config 4's
(2,400 nodes of a 150k-point scene, 64 neighbours within 0.1 m).  usage: PYTHONPATH=. python tools/probes/forest_phase_probe.py"""
import ctypes
import numpy as np
import torch
from pointcloudpdf_amd import _native, pseudo_label, synthetic

be = _native.hip_backend()
sc = synthetic.make_scene(150000, scene_id=3, kind="scannet")
coord = torch.from_numpy(sc["coord"]).cuda()
n = coord.shape[0]
nn = pseudo_label.radius_neighbors(coord, torch.tensor([n], dtype=torch.int32, device="cuda"), 0.1, 64)
g = torch.Generator().manual_seed(0)
msp = torch.rand(n, generator=g).cuda()
seeds = torch.randint(0, n, (100,), generator=g).cuda()
region = torch.unique(torch.cat([seeds, nn[seeds].reshape(-1)]))
region = region[region != -1][:2400]
node_nn = nn[region]
sim = pseudo_label._pair_similarity(region, node_nn, coord, msp)
member = torch.zeros(n + 1, dtype=torch.bool, device="cuda")
member[region] = True
keep = member[node_nn] & (node_nn != -1) & (node_nn != region[:, None])
eu, ev, ew = region[:, None].expand_as(node_nn)[keep], node_nn[keep], sim[keep]
print("nodes", region.numel(), "entries", eu.numel())
for rep in range(3):
    torch.cuda.synchronize()
    chosen, comp = be.graph_forest(n, eu, ev, region, weight=ew)
    torch.cuda.synchronize()
buf = (ctypes.c_longlong * (64 * 8))()
be.lib.pdf_graph_forest_profile.restype = ctypes.c_int
assert be.lib.pdf_graph_forest_profile(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(64, 8)
print("tree entries", int(chosen.sum()), "components", torch.unique(comp[region]).numel())
for r in range(48):
    if t[r, 2] == 0 or (r and t[r, 0] < t[r - 1, 0]):
        break
    us = lambda a, b: (t[r, b] - t[r, a]) / 100.0
    print(f"round {r}: reset {us(0, 1):7.1f} us, edge walk {us(1, 2):7.1f} us, hook + jump {us(2, 3) if t[r, 3] > t[r, 2] else 0:7.1f} us")
