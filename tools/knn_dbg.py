import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pointcloudpdf_amd import _native
be = _native.hip_backend()
def cloud(n, seed, snap=0):
    rng = np.random.default_rng(seed)
    xyz = rng.random((n, 3)).astype(np.float32) * np.array([8, 6, 3], dtype=np.float32)
    if snap:
        xyz = (np.floor(xyz * snap) / snap).astype(np.float32)
    return torch.from_numpy(xyz)
for sizes in ([1500, 37, 2900, 600], [3000, 2037], [1500, 37], [37, 1500]):
    for snap in (0, 5):
        xyz = cloud(sum(sizes), 11 + 3, snap).cuda()
        off = torch.tensor(np.cumsum(sizes), dtype=torch.int32).cuda()
        for k in (3, 16):
            be.knn_mode = "scan"; i0, d0 = be.knn_query(k, xyz, xyz, off, off)
            be.knn_mode = "grid"; i1, d1 = be.knn_query(k, xyz, xyz, off, off)
            torch.cuda.synchronize()
            bad = (i0 != i1).any(1).nonzero().flatten()
            print(sizes, "snap", snap, "k", k, "bad rows", bad.numel(), "first", bad[:5].tolist(), "last", bad[-3:].tolist())
            for r in bad[:2].tolist():
                print("  ", r, i0[r].tolist(), i1[r].tolist()); print("  ", d0[r].tolist()); print("  ", d1[r].tolist())
