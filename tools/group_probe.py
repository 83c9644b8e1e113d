"""grouping2 vs grouping(with_xyz) forward at the level-1 shape, a few launches each (for rocprofv3 --pmc passes: tools/group_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic, _native
from pointcloudpdf_amd.geometry import Geometry
dev = torch.device("cuda")
be = _native.hip_backend()
b = synthetic.make_batch([100000, 100000], first_scene_id=0, device=dev)
geom = Geometry(b["coord"], b["offset"], b["offset_host"])
idx, _ = geom.knn(8, 0, 0)
p = geom.levels[0].p
n = p.shape[0]
feat = torch.randn(n, 32, device=dev)
for _ in range(5):
    be.grouping_forward(feat, idx)
    be.group_forward(feat, p, p, idx, True)
torch.cuda.synchronize()
print("ok", n)
