"""Pseudo-label pass on the GPU against the reference fixtures: mask sizes and the number of differing points per scene (exact = 0)."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from pointcloudpdf_amd import pseudo_label
import test_pseudo_label as T
gp = np.load(os.path.join("tests", "golden", [f for f in os.listdir("tests/golden") if "pseudo" in f][0]))
tags = sorted(T.PSEUDO_CASES)
scenes = [T.pseudo_label_scene(*T.PSEUDO_CASES[t]) for t in tags]
coord = torch.cat([s[0] for s in scenes]).cuda(); logits = torch.cat([s[1] for s in scenes]).cuda()
sizes = [s[0].shape[0] for s in scenes]
off = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device="cuda")
nn = pseudo_label.radius_neighbors(coord, off, 0.1, 64)
start = 0
for t, n in zip(tags, sizes):
    seed = T.PSEUDO_CASES[t][0]
    np.random.seed(seed)
    local = nn[start:start + n].clone(); local[local != -1] -= start
    mask = pseudo_label.pseudo_labeling(coord[start:start + n], logits[start:start + n], local, generator=torch.Generator().manual_seed(seed), **T.PSEUDO_KW).numpy()
    ref = gp[f"{t}_mask"]
    print(t, "mask", int(mask.sum()), "ref", int(ref.sum()), "xor", int((mask ^ ref).sum()), "iou", (mask & ref).sum() / max((mask | ref).sum(), 1))
    start += n
