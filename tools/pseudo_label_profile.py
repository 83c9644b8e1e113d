import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pointcloudpdf_amd import pseudo_label, synthetic
n = 100000
b = synthetic.make_batch([n], first_scene_id=70, kind="scannet", device="cuda")
coord, off = b["coord"], b["offset"]
g = torch.Generator(device="cuda").manual_seed(0)
centre = coord[torch.randint(0, n, (1,), device="cuda", generator=g)]
conf = 6.0 * torch.sigmoid((torch.norm(coord - centre, dim=-1) - 0.8) * 4.0) + 0.3 * torch.randn(n, device="cuda", generator=g)
logits = 0.2 * torch.randn(n, 20, device="cuda", generator=g)
logits[torch.arange(n, device="cuda"), (coord[:, 0] * 3).long() % 20] += conf
nn = pseudo_label.radius_neighbors(coord, off, 0.1, 64)
np.random.seed(0)
pseudo_label.get_pseudo_mask(coord, logits, off, neighbors=nn, generator=torch.Generator().manual_seed(0))
np.random.seed(0)
pr = cProfile.Profile(); pr.enable()
m = pseudo_label.get_pseudo_mask(coord, logits, off, neighbors=nn, generator=torch.Generator().manual_seed(0))
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
